for rep in 1 2 3; do
  for f in 1 0; do
    PP_NM_MIX=$f PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.dbg.so timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-steps 0 --no-secondary 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('PP_NM_MIX=$f %8.0f res/s %7.3f ms  EU %.2f us  NM %.2f us  NU %.2f us  dchi %s' % (d['value'], d['ms_per_step'], r['kernel_ms']*1e3, r['node_message_kernel_ms']*1e3, r['node_update']['kernel_ms']*1e3, d['parity']['max_abs_dchi_vs_reference_rad']))"
  done
done
