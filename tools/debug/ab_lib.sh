#!/bin/bash
# GPU box: the default library against one variant library, interleaved.   bash tools/debug/ab_lib.sh <variant tag> [workloads...]
export PACKPPI_ALLOW_LAB_LIBRARY=1      # tagged variant libraries (lib.load() refuses them otherwise)
TAG=$1; shift; WLS=${@:-t1124 s1500}
for rep in 1 2; do
  for so in libpackppi_hip.so libpackppi_hip.$TAG.so; do
    for wl in $WLS; do
      PACKPPI_LIB=$PWD/packppi_amd/csrc/$so timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d.get('secondary') or {}
print('%-28s %-6s %8.0f res/s %7.3f ms  EU %.2f us  NM %.2f us  secondary %s  dchi %s' % ('$so', '$wl', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, r['node_message_kernel_ms']*1e3, ('%.0f' % s['value']) if s else '-', d['parity']['max_abs_dchi_vs_reference_rad']))"
    done
  done
done
