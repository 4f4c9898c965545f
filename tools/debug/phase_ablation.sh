export PACKPPI_ALLOW_LAB_LIBRARY=1      # tagged variant libraries (lib.load() refuses them otherwise)
for t in ts ts_nowl ts_nobt ts_nowlbt ts_nomf; do
  echo "=== $t"
  PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.$t.so PACKPPI_SKIP_BUILD_CHECK=1 PP_EDGE_R=2 timeout -k 10 120 python tools/debug/phase_alone.py 2>&1 | grep -v amdgpu | grep -E "L =|FFN block 1|second layer|prologue|LN2  |tail first"
done
