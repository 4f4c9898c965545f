#!/bin/bash
# GPU box: the 256-complex job on one GPU (bench.py --workload c5) for several packed-batch sizes of parallel.sample_sharded.
#   bash tools/debug/ab_c5_rows.sh "200000 40000 20000 10000" [reps]
for rep in $(seq ${2:-2}); do
  for m in $1; do
    timeout -k 10 300 python bench.py --workload c5 --steps 3 --warmup 1 --cpu-steps 0 --no-secondary --no-roofline --c5-max-rows $m 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('max_rows %7d  %8.0f res/s %8.2f ms  rows gathered %d' % ($m, d['value'], d['ms_per_step'], d['metrics_rows_gathered']))"
  done
done
