#!/bin/bash
# Bisection of the v_cvt_pk_f16_f32 hazard (pp_edge_f16.hip cvt2): tagged libraries with gfx950's packed conversion at single call
# sites (-DPP_X_CVT_PK_SITES=mask: 1 high part of a split, 2 low part, 4 the geometry operands' point features, 8 / 16 the high / low part of their {distance, 0} pair), each through a short soak.
#   build here:  for m in 1 2 4 3; do PACKPPI_VARIANT_SOURCES=pp_edge_f16.hip python -m packppi_amd.build --tag cvs$m -DPP_LAB -DPP_X_CVT_PK_SITES=$m; done
#   GPU box:     bash tools/debug/cvt_pk_bisect.sh
export PACKPPI_ALLOW_LAB_LIBRARY=1
for m in ${MASKS:-1 2 4 3}; do
  echo "== v_cvt_pk_f16_f32 at sites mask $m"
  PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.cvs$m.so timeout -k 10 200 python tools/debug/soak.py 3 300 800 2>&1 | grep -E "deviate|SOAK|worst"
done
