#!/bin/bash
# GPU box: layer 0 as one fused launch (k_nm_nu0) against the two launches, same library (libpackppi_hip.dbg.so reads PP_FUSE0), alternating.
#   bash tools/debug/ab_fuse0.sh [workload] [reps]
WL=${1:-t1124}; REPS=${2:-3}
for rep in $(seq $REPS); do
  for f in 1 0; do
    PP_FUSE0=$f PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.dbg.so timeout -k 10 200 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-steps 0 --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('PP_FUSE0=$f %-6s %8.0f res/s %7.3f ms  EU %.2f us  NM %.2f us  NU %.2f us (mean of 3)  fused %s  dchi %s' % ('$WL', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, (r['node_message_kernel_ms'] or 0)*1e3, r['node_update']['kernel_ms']*1e3, r.get('layer0_fused_launch'), d['parity']['max_abs_dchi_vs_reference_rad']))"
  done
done
