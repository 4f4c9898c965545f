#!/bin/bash
# On the GPU box: bench every library variant under packppi_amd/csrc/variants/ (tools/debug/build_variants.sh).
export PACKPPI_ALLOW_LAB_LIBRARY=1      # tagged variant libraries (lib.load() refuses them otherwise)
for so in packppi_amd/csrc/variants/*.so; do
  PACKPPI_LIB=$PWD/$so timeout -k 10 200 python bench.py --steps 5 --warmup 2 --cpu-steps 0 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-10s %8.0f res/s %7.2f ms  EU %.1f us  NM0 %.1f us  NU %.1f us  dchi %s' % ('$(basename $so .so)', d['value'], d['ms_per_step'], r['kernel_ms']*1e3, r['node_message_kernel_ms']*1e3, r['node_update_kernel_ms']*1e3, d['parity']['max_abs_dchi_vs_reference_rad']))"
done
