#!/bin/bash
# GPU box: the whole committed measurement set of a build (rocprofv3 stats + PMC passes, L2 counters, per-workload stats, the
# bench lines); summaries land in profiles/ on the box and are copied to gpurun_out/ for the trip home.
set -e
export PROFILE_TAG=${PROFILE_TAG:-r04_v1}
bash tools/profile/run_profiles.sh v3 > gpurun_out/v2_profiles.log 2>&1
echo profiles done
bash tools/profile/run_tcc.sh t1124 > gpurun_out/v2_tcc.log 2>&1
echo tcc done
bash tools/profile/run_stats_workload.sh v3 s1500 > gpurun_out/v2_stats_s1500.log 2>&1
bash tools/profile/run_stats_workload.sh v3 c5 > gpurun_out/v2_stats_c5.log 2>&1
echo stats done
python bench.py > gpurun_out/${PROFILE_TAG}_bench_t1124.json 2> gpurun_out/v2_bench.err
python bench.py --proximal > gpurun_out/${PROFILE_TAG}_bench_t1124_prox.json 2>> gpurun_out/v2_bench.err
python bench.py --workload s1500 > gpurun_out/${PROFILE_TAG}_bench_s1500.json 2>> gpurun_out/v2_bench.err
python bench.py --workload s1500 --proximal > gpurun_out/${PROFILE_TAG}_bench_s1500_prox.json 2>> gpurun_out/v2_bench.err
python bench.py --workload c5 > gpurun_out/${PROFILE_TAG}_bench_c5.json 2>> gpurun_out/v2_bench.err
echo bench done
python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "split_launch" > gpurun_out/v2_split_test.log 2>&1
tail -2 gpurun_out/v2_split_test.log
cp profiles/${PROFILE_TAG}_* gpurun_out/ 2>/dev/null || true
