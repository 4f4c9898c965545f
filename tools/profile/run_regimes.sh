#!/bin/bash
# GPU box: MFMA-busy and L2-request counters of the edge kernels in the three regimes the bench line reports (T1124: one round of
# the mixed launch; S1500: 1.5 rounds; c5share: one GPU's share of BASELINE configs[4], 38 residues per CU), each its own rocprofv3
# --pmc pass of `bench.py --workload W --steps 1 --warmup 0 --cpu-steps 0 --no-secondary --build-workers 1 --no-roofline --diffusion-steps 6` (six
# evaluations = twelve edge launches: per-launch counters do not need the 100 steps of the metric, and a TCC pass over 100 steps of
# the 32-complex batch ran past the pool's 7-minute silence limit).
#   GRAFT_ROUND=r05 PROFILE_TAG=r05_v1 bash tools/profile/run_regimes.sh      -> profiles/<tag>_regimes.json (+ gpurun_out/ copy)
set -e
export PROFILE_TAG=${PROFILE_TAG:-r05_v1}
ROOT=$(pwd); OUT=$ROOT/gpurun_out/regimes; mkdir -p $OUT
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}      # before rocprofv3 starts (its preloaded library initialises the HIP runtime ahead of python)
cd /tmp
for W in t1124 s1500 c5share; do
  BENCH="python3 $ROOT/bench.py --workload $W --steps 1 --warmup 0 --cpu-steps 0 --no-secondary --build-workers 1 --no-roofline --diffusion-steps 6"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $OUT/sq_$W -o run -- $BENCH > $OUT/sq_$W.json 2> $OUT/sq_$W.err || echo "sq $W failed: $(tail -2 $OUT/sq_$W.err)"
  echo "sq $W done"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc_$W -o run -- $BENCH > $OUT/tcc_$W.json 2> $OUT/tcc_$W.err || echo "tcc $W failed: $(tail -2 $OUT/tcc_$W.err)"
  echo "tcc $W done"
done
cd $ROOT; find $OUT -type f ! -name "*counter_collection.csv" ! -name "*.err" -delete
python3 tools/profile/regimes.py $OUT
cp profiles/${PROFILE_TAG}_regimes.json gpurun_out/
