#!/bin/bash
# Build several flag variants of the library side by side (csrc/variants/<name>.so) for A/B runs in one gpurun call:
#   tools/debug/build_variants.sh name1 "flags1" name2 "flags2" ...   then   PACKPPI_LIB=packppi_amd/csrc/variants/name1.so python bench.py
set -e
cd "$(dirname "$0")/../.."
mkdir -p packppi_amd/csrc/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  PACKPPI_CFLAGS="$flags" python -m packppi_amd.build --force > /dev/null
  cp packppi_amd/csrc/libpackppi_hip.so packppi_amd/csrc/variants/$name.so
  echo "built $name: $flags"
done
python -m packppi_amd.build --force > /dev/null      # leave the default build in place
