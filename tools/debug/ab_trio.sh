#!/bin/bash
# GPU box: the trio launch (one 512-thread workgroup per CU, two teams in lockstep, one weight fill per CU) against the mixed launch;
# libpackppi_hip.dbg.so reads PP_EDGE_TRIO.   bash tools/debug/ab_trio.sh [reps]
for rep in $(seq ${1:-3}); do
  for f in 1 0; do
    PP_EDGE_TRIO=$f PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.dbg.so timeout -k 10 120 python bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('PP_EDGE_TRIO=$f %8.0f res/s %7.3f ms  EU %.2f us  NM %.2f us  NU %.2f us  dchi %s' % (d['value'], d['ms_per_step'], r['kernel_ms']*1e3, r['node_message_kernel_ms']*1e3, r['node_update']['kernel_ms']*1e3, d['parity']['max_abs_dchi_vs_reference_rad']))" || { echo "run failed (PP_EDGE_TRIO=$f)"; exit 1; }
  done
done
