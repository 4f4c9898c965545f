#!/bin/bash
# GPU box: the default library against several variant libraries, interleaved, one workload.
#   bash tools/debug/ab_libs.sh "<tag> <tag> ..." [workload] [reps]      ("base" = libpackppi_hip.so)
export PACKPPI_ALLOW_LAB_LIBRARY=1      # tagged variant libraries (lib.load() refuses them otherwise)
TAGS=$1; WL=${2:-t1124}; REPS=${3:-2}
for rep in $(seq $REPS); do
  for t in $TAGS; do
    if [ "$t" = base ]; then so=libpackppi_hip.so; else so=libpackppi_hip.$t.so; fi
    PACKPPI_LIB=$PWD/packppi_amd/csrc/$so timeout -k 10 200 python bench.py --workload $WL --steps 10 --warmup 3 --cpu-steps 0 --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-28s %-6s %8.0f res/s %7.3f ms  EU %.2f us  NM %.2f us  NU %.2f us  dchi %s' % ('$so', '$WL', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, r['node_message_kernel_ms']*1e3, r['node_update']['kernel_ms']*1e3, d['parity']['max_abs_dchi_vs_reference_rad']))"
  done
done
