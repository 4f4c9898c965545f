# GPU box: residues per workgroup of the layer-0 node message (libpackppi_hip.dbg.so, PP_NM_R), T1124
for rep in 1 2; do for m in 1 2; do
PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.dbg.so PP_NM_R=$m timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('PP_NM_R=$m %8.0f res/s %7.3f ms  NM %.2f us  EU %.2f us dchi %s' % (d['value'], d['ms_per_step'], r['node_message_kernel_ms']*1e3, r['kernel_ms']*1e3, d['parity']['max_abs_dchi_vs_reference_rad']))"
done; done
