#!/bin/bash
# GPU box: HIP_FORCE_DEV_KERNARG=0 against 1 (the package default), interleaved.   bash tools/debug/ab_devkernarg.sh
grep -m1 "model name" /proc/cpuinfo
for rep in 1 2 3; do
  for v in 0 1; do
    for wl in t1124 s1500; do
      HIP_FORCE_DEV_KERNARG=$v timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-steps 0 --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('HIP_FORCE_DEV_KERNARG=$v %-6s %8.0f res/s %7.3f ms  EU %.2f  NM %.2f  NU %.2f us' % ('$wl', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, r['node_message_kernel_ms']*1e3, r['node_update']['kernel_ms']*1e3))"
    done
  done
done
