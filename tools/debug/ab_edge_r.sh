#!/bin/bash
# GPU box: residues per workgroup of the edge update (and of the node message) in the throughput regimes: R = 1 at three workgroups per CU
# against the default R = 2 at two (libpackppi_hip.dbg.so, PP_EDGE_R).   bash tools/debug/ab_edge_r.sh [workload] [reps]
WL=${1:-c5share}; REPS=${2:-2}
for rep in $(seq $REPS); do for m in default 1; do
if [ "$m" = default ]; then unset PP_EDGE_R; else export PP_EDGE_R=$m; fi
PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.dbg.so timeout -k 10 300 python bench.py --workload $WL --steps 5 --warmup 2 --cpu-steps 0 --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('PP_EDGE_R=%-8s %-8s %8.0f res/s %8.3f ms  EU %.2f us  NM %.2f us  NU %.2f us' % ('$m', '$WL', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, r['node_message_kernel_ms']*1e3, r['node_update']['kernel_ms']*1e3))"
done; done
