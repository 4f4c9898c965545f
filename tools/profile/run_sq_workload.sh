#!/bin/bash
# GPU box: MFMA-busy / wave-cycle counters of the edge kernels for another workload.  Usage: bash tools/profile/run_sq_workload.sh c5
set -e
W=${1:-c5}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sqw_$W; mkdir -p $OUT
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}      # before rocprofv3 starts: its preloaded library initialises the HIP runtime ahead of python (packppi_amd/__init__.py would set it too late)
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-steps 0 --no-secondary --build-workers 1 --workload $W > $OUT/bench.json 2> $OUT/err.txt
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:40]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    if "edge_update" in k or "node_message" in k or "node_update" in k:
        print(k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
find $OUT -type f ! -name "*counter_collection.csv" -delete
