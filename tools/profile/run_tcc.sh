#!/bin/bash
# Runs on the GPU box: L2 hit / miss counters of the hot kernels for one workload (is the per-workgroup weight stream served
# from the XCD's L2, or does the h_E stream evict it?).   bash tools/profile/run_tcc.sh <workload> [extra bench flags]
set -e
ROOT=$(pwd)
WL=${1:-t1124}; shift || true
OUT=$ROOT/gpurun_out/tcc_$WL
mkdir -p $OUT
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}      # before rocprofv3 starts: its preloaded library initialises the HIP runtime ahead of python (packppi_amd/__init__.py would set it too late)
cd /tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/p1 -o run -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 1 --cpu-steps 0 --no-secondary --build-workers 1 "$@" > $OUT/p1.json 2> $OUT/p1.err || echo "pass failed: $(tail -2 $OUT/p1.err)"
cd $ROOT; find $OUT -type f ! -name "*counter_collection.csv" ! -name "*.err" -delete
python3 tools/profile/tcc_summary.py $OUT $WL
