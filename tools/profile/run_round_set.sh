#!/bin/bash
# GPU box: the whole committed measurement set of a build (rocprofv3 stats + PMC passes, L2 counters, SQ counters, per-workload
# stats, the bench lines); summaries land in profiles/ on the box and are copied to gpurun_out/ for the trip home.
#   GRAFT_ROUND=r05 PROFILE_TAG=r05_v2 bash tools/profile/run_round_set.sh      (two calls: STOP_BEFORE_REGIMES=1, then RESUME_AT_REGIMES=1)
set -e
export GRAFT_ROUND=${GRAFT_ROUND:-r05}
export PROFILE_TAG=${PROFILE_TAG:-${GRAFT_ROUND}_v1}
V=${PROFILE_TAG#${GRAFT_ROUND}_}
if [ -z "$RESUME_AT_REGIMES" ]; then
bash tools/profile/run_profiles.sh $V > gpurun_out/set_profiles.log 2>&1
echo profiles done
bash tools/profile/run_tcc.sh t1124 > gpurun_out/set_tcc.log 2>&1
echo tcc done
bash tools/profile/run_sq.sh > gpurun_out/${PROFILE_TAG}_sq_counters.txt 2> gpurun_out/set_sq.err
bash tools/profile/run_sq_workload.sh c5share > gpurun_out/${PROFILE_TAG}_sq_c5.txt 2>> gpurun_out/set_sq.err
bash tools/profile/run_sq_workload.sh s1500 > gpurun_out/${PROFILE_TAG}_sq_s1500.txt 2>> gpurun_out/set_sq.err
echo sq done
fi
if [ -n "$STOP_BEFORE_REGIMES" ]; then cp profiles/${PROFILE_TAG}_* gpurun_out/ 2>/dev/null || true; exit 0; fi      # first half of a set that does not fit one call
bash tools/profile/run_regimes.sh > gpurun_out/set_regimes.log 2>&1       # MFMA-busy + L2 requests per regime -> profiles/<tag>_regimes.json (bench.py reads it)
echo regimes done
bash tools/profile/run_stats_workload.sh $V s1500 > gpurun_out/set_stats_s1500.log 2>&1
bash tools/profile/run_stats_workload.sh $V c5share "" c5 > gpurun_out/set_stats_c5.log 2>&1
bash tools/profile/run_stats_workload.sh $V t1124 --proximal t1124_prox > gpurun_out/set_stats_prox.log 2>&1
echo stats done
cp profiles/${PROFILE_TAG}_* gpurun_out/ 2>/dev/null || true      # (before the bench lines: a committed older line in profiles/ must not overwrite a fresh one)
python bench.py --steps 20 --warmup 5 > gpurun_out/${PROFILE_TAG}_bench_t1124.json 2> gpurun_out/set_bench.err
python bench.py --proximal > gpurun_out/${PROFILE_TAG}_bench_t1124_prox.json 2>> gpurun_out/set_bench.err
python bench.py --workload s1500 > gpurun_out/${PROFILE_TAG}_bench_s1500.json 2>> gpurun_out/set_bench.err
python bench.py --workload s1500 --proximal > gpurun_out/${PROFILE_TAG}_bench_s1500_prox.json 2>> gpurun_out/set_bench.err
python bench.py --workload c5 > gpurun_out/${PROFILE_TAG}_bench_c5.json 2>> gpurun_out/set_bench.err
echo bench done
