#!/bin/bash
# GPU box: SQ counter passes (wave cycles, waits, instruction mix, MFMA busy, LDS) of the edge kernels for one workload.
#   bash tools/profile/run_sq_edge.sh <workload> [output name]      -> gpurun_out/<name>.txt
set -e
W=${1:-c5}; NAME=${2:-sq_edge_$W}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/sqe_$W; mkdir -p $OUT
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}      # before rocprofv3 starts (its preloaded library initialises the HIP runtime ahead of python)
cd /tmp
BENCH="python3 $ROOT/bench.py --workload $W --steps 1 --warmup 1 --cpu-steps 0 --no-secondary --build-workers 1 --no-roofline"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o run -- $BENCH > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed: $(tail -2 $OUT/p$i.err)"
done
cd $ROOT; find $OUT -type f ! -name "*counter_collection.csv" ! -name "*.err" -delete
python3 - "$OUT" > gpurun_out/$NAME.txt <<'PY'
import csv, glob, collections, sys
print("# rocprofv3 --pmc <set> -- python3 bench.py --workload W --steps 1 --warmup 1 --cpu-steps 0 --no-secondary --build-workers 1 --no-roofline; mean per launch")
for f in sorted(glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        if "edge_update" in k or "node_message" in k or "node_update" in k:
            print(k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
cat gpurun_out/$NAME.txt
