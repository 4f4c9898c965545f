#!/bin/bash
# Register / LDS / spill figures of the kernels of one source file (compiler remarks; device code only, nothing is written).
# Usage: bash tools/debug/kernel_regs.sh packppi_amd/csrc/pp_edge_f16.hip "edge_update|node_message" [-DFLAG ...]
src=$1; pat=${2:-.}; shift 2
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -DPP_EDGE_F16 "$@" --cuda-device-only -c $src -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re
cur = {}
for line in sys.stdin:
    m = re.search(r'remark: .*?(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|VGPR Spill|Occupancy \[waves/SIMD\]): (\S+)', line)
    if not m: continue
    k, v = m.groups()
    if k == 'Function Name':
        cur = {'name': v}
    cur[k] = v
    if k.startswith('LDS') and re.search(r'$pat', cur['name']):
        import subprocess
        n = subprocess.run(['c++filt', cur['name']], capture_output=True, text=True).stdout.strip().split('(')[0]
        print(f\"{n[:64]:64s} vgpr {cur.get('VGPRs','?'):>4s} agpr {cur.get('AGPRs','?'):>3s} spill {cur.get('VGPR Spill','?'):>3s} scratch {cur.get('ScratchSize [bytes/lane]','?'):>4s} occ {cur.get('Occupancy [waves/SIMD]','?')}\")
"
