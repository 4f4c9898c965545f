"""GPU box: in-situ time of the edge-update launches (T1124, S1500, the configs[4] share), with the duo launch (k_edge_update_duo)
forced off and on when the loaded library is the diagnostic one (PACKPPI_LIB=.../libpackppi_hip.dbg.so).
    python tools/debug/time_edge.py [t1124] [s1500] [c5] [steps]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
import torch  # noqa: E402
from bench import c5_share, c5_inits, load_s1500, load_t1124  # noqa: E402
from packppi_amd import lib  # noqa: E402
from packppi_amd.batch import pack  # noqa: E402
from packppi_amd.lib import Context  # noqa: E402
from packppi_amd.module import TDiffusionModule  # noqa: E402
from packppi_amd.weights import make_random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
l = lib.load()
force = getattr(l, "pp_debug_set_edge_duo", None)      # only libraries built from the duo experiment commits export it
if force is not None:
    force.argtypes = [C.c_int]
    force.restype = None
m = TDiffusionModule(make_random_state_dict(20251003), device=dev)
steps = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 20
sched = torch.linspace(1, 0, steps + 1)
wls = [a for a in sys.argv[1:] if not a.isdigit()] or ["s1500", "c5"]
for wl in wls:
    if wl == "s1500":
        b, init, ref = load_s1500()
        gb, x0 = b.to(dev), init.to(dev)
    elif wl == "t1124":
        b, init, ref = load_t1124()
        gb, x0 = b.to(dev), init.to(dev)
    else:
        _, share = c5_share(0, 8, dev)
        ini = c5_inits(share, 1000)
        x0 = torch.cat([ini[i][:, : c.true_residues()] for i, c in share.items()], 1).to(dev)
        gb = pack(list(share.values()))
    for mode in ((0, 1) if force is not None else (None,)):
        if mode is not None:
            force(mode)
        ctx = Context(m._plan, gb)
        ctx.sample(x0, sched)
        ctx.profile_kernel(1)
        out = ctx.sample(x0, sched)
        ms, n = ctx.profile_read()
        torch.cuda.synchronize()
        t0 = time.time()
        ctx.sample(x0, sched)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"{wl:6s} duo={mode}  edge launch {ms * 1e3:8.1f} us x {n}   pass {dt / steps * 1e3:7.3f} ms/eval   finite {bool(torch.isfinite(out).all())}", flush=True)
if force is not None:
    force(-1)
