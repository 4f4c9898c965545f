#!/bin/bash
# GPU box: the node update of a packed batch as two-tile 1024-thread workgroups (teams in lockstep, one weight fill per CU) against
# plain 512-thread workgroups (libpackppi_hip.dbg.so, PP_NU_TEAMS = 0 never / 1 always / unset: when tiles > 2 x CUs).
#   bash tools/debug/ab_nu_teams.sh [workload] [reps]
WL=${1:-c5share}; REPS=${2:-3}
for rep in $(seq $REPS); do for m in 0 auto; do
if [ "$m" = auto ]; then unset PP_NU_TEAMS; else export PP_NU_TEAMS=$m; fi
PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.dbg.so timeout -k 10 300 python bench.py --workload $WL --steps 5 --warmup 2 --cpu-steps 0 --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('PP_NU_TEAMS=%-4s %-8s %8.0f res/s %8.3f ms  EU %.2f us  NM %.2f us  NU %.2f us' % ('$m', '$WL', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, r['node_message_kernel_ms']*1e3, r['node_update']['kernel_ms']*1e3))"
done; done
