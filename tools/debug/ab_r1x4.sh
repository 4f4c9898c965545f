#!/bin/bash
# default library (R = 2 at two per CU) against the r1x4 lab library with PP_EDGE_R=1 (one-residue workgroups at FOUR per CU)
WL=${1:-c5share}; REPS=${2:-2}
export PACKPPI_ALLOW_LAB_LIBRARY=1
for rep in $(seq $REPS); do for m in base r1x4; do
if [ "$m" = base ]; then unset PP_EDGE_R; so=libpackppi_hip.so; else export PP_EDGE_R=1; so=libpackppi_hip.r1x4.so; fi
PACKPPI_LIB=$PWD/packppi_amd/csrc/$so timeout -k 10 300 python bench.py --workload $WL --steps 5 --warmup 2 --cpu-steps 0 --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-6s %-8s %8.0f res/s %8.3f ms  EU %.2f us  NM %.2f us  NU %.2f us dchi %s' % ('$m', '$WL', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, r['node_message_kernel_ms']*1e3, r['node_update']['kernel_ms']*1e3, d['parity']['max_abs_dchi_vs_reference_rad']))"
done; done
