#!/bin/bash
# Runs on the GPU box (via gpurun): one kernel-trace/stats pass and two separate PMC passes over the bench command.
# Usage: bash tools/profile/run_profiles.sh <tag>      -> gpurun_out/prof_<tag>/{stats,fetch,write}
set -e
TAG=${1:-final}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}      # before rocprofv3 starts: its preloaded library initialises the HIP runtime ahead of python (packppi_amd/__init__.py would set it too late)
cd /tmp
BENCH="python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o run -- $BENCH > $OUT/stats.json 2> $OUT/stats.err
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o run -- $BENCH > $OUT/fetch.json 2> $OUT/fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o run -- $BENCH > $OUT/write.json 2> $OUT/write.err
echo "write pass done"
cd $ROOT; find $OUT -type f ! -name "*.csv" ! -name "*.json" ! -name "*.err" -delete; du -sh $OUT | cut -f1
tail -2 $OUT/stats.err; python3 tools/profile/summarize.py $OUT $TAG
