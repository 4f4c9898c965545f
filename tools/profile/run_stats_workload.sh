#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --stats summary of bench.py for another workload -> profiles/<ROUND>_<tag>_kernel_stats_<workload>.csv
# Usage: bash tools/profile/run_stats_workload.sh <tag> <workload> [extra bench flags, e.g. --proximal] [suffix of the output name]
set -e
TAG=${1:-v1}; WL=${2:-s1500}; EXTRA=${3:-}; SUF=${4:-$WL}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/stats_$WL; mkdir -p $OUT
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}      # before rocprofv3 starts: its preloaded library initialises the HIP runtime ahead of python (packppi_amd/__init__.py would set it too late)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --workload $WL $EXTRA --steps 2 --warmup 1 --cpu-steps 0 --no-secondary --build-workers 1 > $OUT/bench.json 2> $OUT/err.txt
cd $ROOT
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/${GRAFT_ROUND:-r05}_${TAG}_kernel_stats_$SUF.csv
find $OUT -type f ! -name "bench.json" -delete
head -8 gpurun_out/${GRAFT_ROUND:-r05}_${TAG}_kernel_stats_$SUF.csv | cut -c1-150
