#!/bin/bash
# GPU box: clocks after every stage of the edge update, 16 stages per build (libpackppi_hip.tsf<k0>.so built with
# PACKPPI_VARIANT_SOURCES="pp_edge_f16.hip pp_api.hip" python -m packppi_amd.build --tag tsf<k0> -DPP_LAB -DPP_X_TS -DPP_X_TS_FINE=<k0>, k0 = 0 16 32 48)
export PACKPPI_ALLOW_LAB_LIBRARY=1      # tagged variant libraries (lib.load() refuses them otherwise)
for k in 0 16 32 48; do
  PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.tsf$k.so PACKPPI_SKIP_BUILD_CHECK=1 PP_EDGE_R=2 timeout -k 10 200 python tools/debug/stage_times.py $k 2>&1 | grep -v amdgpu
done
