"""GPU box: the proximal loop of the library named by PACKPPI_LIB (default: the product library) on T1124 and S1500 -- time of 50 Adam
steps, in-situ time of the per-step launch(es), and a digest of the trajectory and the loss curve (two builds that agree print the same
digest).   PACKPPI_LIB=... [PACKPPI_SKIP_BUILD_CHECK=1] python tools/debug/prox_fused_ab.py"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
import numpy as np
import torch
from bench import load_s1500, load_t1124
from packppi_amd.functional import _ctx_for, proximal_optimizer
tag = os.path.basename(os.environ.get("PACKPPI_LIB", "libpackppi_hip.so"))
for name, load in (("T1124", load_t1124), ("S1500", load_s1500)):
    b, init, ref = load()
    gb = b.to("cuda:0")
    chi = ref.to("cuda:0").float()
    chis, losses = proximal_optimizer(gb, chi, 12.0, 0.5, 1.0, 50)
    h = hashlib.sha256(b"".join(c.cpu().numpy().tobytes() for c in chis) + np.asarray(losses, dtype=np.float64).tobytes()).hexdigest()[:16]
    ctx = _ctx_for(gb)
    for _ in range(3):
        ctx.proximal(chi, 12.0, 0.5, 1.0, 50, want_traj=False)
    per = {}
    for which in (3, 4):
        try:
            ctx.profile_kernel(which)
            ctx.proximal(chi, 12.0, 0.5, 1.0, 50, want_traj=False)
            per[which] = ctx.profile_read()[0] * 1e3
        except RuntimeError:
            per[which] = None
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10):
        ctx.proximal(chi, 12.0, 0.5, 1.0, 50, want_traj=False)
    torch.cuda.synchronize(); dt = (time.time() - t0) / 10
    k4 = "-" if per[4] is None else f"{per[4]:.1f}"
    print(f"{tag:32s} {name}: 50 Adam steps {dt * 1e3:.3f} ms  launch(3) {per[3]:.1f} us  launch(4) {k4} us  loss {losses[0]:.6f} -> {losses[-1]:.6f}  digest {h}", flush=True)
