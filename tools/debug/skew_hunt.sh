#!/bin/bash
# Race hunting: a library built with -DPP_LAB -DPP_X_SKEW holds one wave of every edge-level workgroup back (~30 k cycles) at one
# point of the chain; a complete barrier protocol gives the same bits under any skew.
#   build here:  PACKPPI_VARIANT_SOURCES=pp_edge_f16.hip python -m packppi_amd.build --tag skew -DPP_LAB -DPP_X_SKEW
#   GPU box:     bash tools/debug/skew_hunt.sh
export PACKPPI_ALLOW_LAB_LIBRARY=1 PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.${SKEW_TAG:-skew}.so
for pt in ${POINTS:-0 1 2 3 4 5 6 7 8 9}; do
  for w in ${WAVES:-0 1 3}; do
    r=$(PP_SKEW=$((16 * pt + w)) timeout -k 10 120 python tools/debug/soak.py 2 300 800 2>&1 | grep -E "deviate|worst" | sed 's/ runs deviate from the majority output//; s/ each)//' | tr '\n' ' ')
    echo "point $pt wave $w: $r"
  done
done
