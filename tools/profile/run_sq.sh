#!/bin/bash
# Runs on the GPU box: SQ counter passes over a short bench run (which pipes the edge kernels' wave cycles go to).
set -e
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sq
mkdir -p $OUT
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}      # before rocprofv3 starts: its preloaded library initialises the HIP runtime ahead of python (packppi_amd/__init__.py would set it too late)
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-steps 0 --no-secondary"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_WAVE32"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d $OUT/p$i -o run -- $BENCH > $OUT/p$i.json 2> $OUT/p$i.err || echo "pass $i failed: $(tail -2 $OUT/p$i.err)"
  echo "pass $i done"
done
cd $ROOT; find $OUT -type f ! -name "*counter_collection.csv" ! -name "*.err" -delete; du -sh $OUT | cut -f1
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/sq/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k in agg:
        if "edge_update" in k or "node_message" in k or "node_update" in k:
            print(f.split("/")[2], k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
