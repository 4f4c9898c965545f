#!/bin/bash
# Runs on the GPU box: instruction-cache counter passes over a short bench run (the edge kernels are ~28 KB of straight-line
# code per instance; every workgroup executes its instance once).
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/icache
mkdir -p $OUT
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-steps 0 --no-secondary --no-roofline $ICACHE_BENCH_ARGS"
i=0
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQC_TC_INST_REQ SQC_ICACHE_BUSY_CYCLES SQC_TC_STALL SQC_ICACHE_INPUT_VALID_READYB"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o run -- $BENCH > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed: $(tail -2 $OUT/p$i.err)"
  echo "pass $i done"
done
cd $ROOT; find $OUT -type f ! -name "*counter_collection.csv" ! -name "*.err" -delete; du -sh $OUT | cut -f1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/icache/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        if "edge_update" in k or "node_message" in k or "node_update" in k:
            print(f.split("/")[2], k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
