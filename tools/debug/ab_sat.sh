#!/bin/bash
# GPU box: the default library against the variant built without the sticky-flag bookkeeping (-DPP_X_NOSAT), interleaved, three
# workloads.  Build the variant first: PACKPPI_VARIANT_SOURCES="pp_edge_f16.hip pp_api.hip" python -m packppi_amd.build --tag nosat -DPP_LAB -DPP_X_NOSAT (load with PACKPPI_ALLOW_LAB_LIBRARY=1) ;  bash tools/debug/ab_sat.sh
export PACKPPI_ALLOW_LAB_LIBRARY=1      # tagged variant libraries (lib.load() refuses them otherwise)
for rep in 1 2; do
  for so in libpackppi_hip.so libpackppi_hip.nosat.so; do
    for wl in t1124 s1500; do
      PACKPPI_LIB=$PWD/packppi_amd/csrc/$so timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d.get('secondary') or {}
print('%-26s %-6s %8.0f res/s %7.3f ms  EU %.2f us  secondary %s' % ('$so', '$wl', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, ('%.0f' % s['value']) if s else '-'))"
    done
  done
done
