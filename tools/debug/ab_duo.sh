#!/bin/bash
# GPU box: the ping-pong launch of the edge update (PP_EDGE_DUO=1) against the default launches, three workloads.
for rep in 1 2; do
  for duo in 0 1; do
    for wl in t1124 s1500; do
      PP_EDGE_DUO=$duo timeout -k 10 200 python bench.py --workload $wl --steps 10 --warmup 3 --cpu-steps 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d.get('secondary') or {}
print('duo=$duo %-6s %8.0f res/s %7.3f ms  EU %.2f us  secondary %s  dchi %s' % ('$wl', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, ('%.0f' % s['value']) if s else '-', d['parity']['max_abs_dchi_vs_reference_rad']))"
    done
  done
done
