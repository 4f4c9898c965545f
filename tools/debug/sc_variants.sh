set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q 2>&1 | tail -3
PP_DEBUG=1 python bench.py --cpu-steps 0 2>gpurun_out/dr.err | cut -c1-200; grep resident gpurun_out/dr.err | head -1
python tools/debug/soak.py 12 2>&1 | grep -v amdgpu | tail -9
python tools/debug/net_repro.py 2300 40 2>&1 | grep -v amdgpu
