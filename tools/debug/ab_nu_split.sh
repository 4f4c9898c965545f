#!/bin/bash
# GPU box: split launches of the middle-layer node update (PP_NU_SPLIT = 1 never, 2, 4 = default), same library, interleaved.
for rep in 1 2; do
  for d in 1 2 4; do
    for wl in t1124 s1500; do
      PP_NU_SPLIT=$d timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d.get('secondary') or {}
print('split %-2s %-6s %8.0f res/s %7.3f ms  EU %.2f us  secondary %s  parity %s' % ('$d', '$wl', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, ('%.0f' % s['value']) if s else '-', json.dumps(d.get('parity'))[:120]))"
    done
  done
done
