"""GPU box, libpackppi_hip.dbg.so: the proximal loop with the static candidate lists against the per-step scan (PP_CLASH_SCAN=1), same process
order independent: run as  PP_CLASH_SCAN=0|1 PACKPPI_LIB=.../libpackppi_hip.dbg.so python tools/debug/prox_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
import torch
from bench import load_s1500, load_t1124
from packppi_amd.functional import _ctx_for
for name, load in (("T1124", load_t1124), ("S1500", load_s1500)):
    b, init, ref = load()
    gb = b.to("cuda:0")
    chi = ref.to("cuda:0").float()
    ctx = _ctx_for(gb)
    for _ in range(3):
        ctx.proximal(chi, 12.0, 0.5, 1.0, 50, want_traj=False)
    res = {}
    for which, kn in ((3, "k_clash"),):
        ctx.profile_kernel(which)
        ctx.proximal(chi, 12.0, 0.5, 1.0, 50, want_traj=False)
        res[kn] = ctx.profile_read()[0] * 1e3
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        ctx.proximal(chi, 12.0, 0.5, 1.0, 50, want_traj=False)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print(f"PP_CLASH_SCAN={os.environ.get('PP_CLASH_SCAN', '0')} {name}: 50 Adam steps {dt * 1e3:.3f} ms   k_clash (the one launch per step) {res['k_clash']:.1f} us", flush=True)
