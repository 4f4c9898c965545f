export PROFILE_TAG=r04_v2
python bench.py --steps 20 --warmup 5 > gpurun_out/${PROFILE_TAG}_bench_t1124.json 2> gpurun_out/set_bench.err
python bench.py --proximal > gpurun_out/${PROFILE_TAG}_bench_t1124_prox.json 2>> gpurun_out/set_bench.err
python bench.py --workload s1500 > gpurun_out/${PROFILE_TAG}_bench_s1500.json 2>> gpurun_out/set_bench.err
python bench.py --workload s1500 --proximal > gpurun_out/${PROFILE_TAG}_bench_s1500_prox.json 2>> gpurun_out/set_bench.err
python bench.py --workload c5 > gpurun_out/${PROFILE_TAG}_bench_c5.json 2>> gpurun_out/set_bench.err
echo bench done
