#!/bin/bash
# GPU box: kernel-trace stats of the bench command for each prebuilt node-kernel variant library (timing experiments).
# Usage: bash tools/profile/node_variants.sh "<tag> <tag> ..."   ("base" = the default library)
export PACKPPI_ALLOW_LAB_LIBRARY=1      # tagged variant libraries (lib.load() refuses them otherwise)
ROOT=$(pwd)
export TMPDIR=/tmp
export HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-1}      # before rocprofv3 starts: its preloaded library initialises the HIP runtime ahead of python (packppi_amd/__init__.py would set it too late)
for v in $1; do
  OUT=$ROOT/gpurun_out/nv_$v; mkdir -p $OUT
  if [ "$v" = base ]; then export PACKPPI_LIB=$ROOT/packppi_amd/csrc/libpackppi_hip.so; else export PACKPPI_LIB=$ROOT/packppi_amd/csrc/libpackppi_hip.$v.so; fi
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 $ROOT/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-secondary --workload ${WORKLOAD:-t1124} > $OUT/bench.json 2> $OUT/err.txt)
  f=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0]
    if "k_node_update" in n or "k_edge_update" in n or "k_node_message" in n:
        print(f'  {n[:60]:60s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:8.2f} us  {r["Percentage"]}%')
PY
  find $OUT -type f ! -name "*stats.csv" ! -name "bench.json" -delete
done
