"""Per-phase time of the throughput-regime edge kernel (pp_edge_w.inc).  Build a stamped library first:
    python -m packppi_amd.build --tag tsw -DPP_LAB -DPP_X_TS
    PACKPPI_ALLOW_LAB_LIBRARY=1 PACKPPI_LIB=$PWD/packppi_amd/csrc/libpackppi_hip.tsw.so python tools/debug/phase_times_w.py [s1500|c5]
Mean / max over the waves of every wave's s_memtime stamps (core-clock cycles)."""
import ctypes as C
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from bench import c5_inits, c5_share, load_s1500  # noqa: E402
from packppi_amd import lib  # noqa: E402
from packppi_amd.batch import pack  # noqa: E402
from packppi_amd.lib import Context  # noqa: E402
from packppi_amd.module import TDiffusionModule  # noqa: E402
from packppi_amd.weights import make_random_state_dict  # noqa: E402

dev = torch.device("cuda", 0)
m = TDiffusionModule(make_random_state_dict(20251003), device=dev)
wl = sys.argv[1] if len(sys.argv) > 1 else "s1500"
if wl == "s1500":
    b, init, _ = load_s1500()
    gb, x0 = b.to(dev), init.to(dev)
else:
    _, share = c5_share(0, 8, dev)
    ini = c5_inits(share, 1000)
    x0 = torch.cat([ini[i][:, : c.true_residues()] for i, c in share.items()], 1).to(dev)
    gb = pack(list(share.values()))
l = lib.load()
l.pp_debug_set_dbg.argtypes = [C.c_void_p]; l.pp_debug_set_dbg.restype = None
l.pp_debug_edge.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
l.pp_debug_set_edge_w.argtypes = [C.c_int]; l.pp_debug_set_edge_w.restype = None
l.pp_debug_set_edge_w(1)
ctx = Context(m._plan, gb)
ctx.sample(x0, torch.linspace(1, 0, 3))
N = x0.shape[1]
nw = (N + 7) // 8 * 4
dbg = torch.zeros(nw, 24, device=dev)
l.pp_debug_set_dbg(C.c_void_p(dbg.data_ptr()))
names = ["prologue (loads, h_E split -> X, ring start, barrier)", "first layer (W_B, geometry, W_G) + publish", "W_mid + publish", "W_out",
         "LN2 + x1 -> X", "FFN tiles 0-3", "FFN tiles 4-7", "FFN tiles 8-11", "FFN tiles 12-15", "LN3 + store h_E (+ X <- h_E)",
         "next message: first layer", "next message: W_mid + reduce + store"]
for layer in (0, 1):
    for rep in range(3):
        dbg.zero_()
        assert l.pp_debug_edge(ctx.handle, layer, None) == 0
        torch.cuda.synchronize()
    t = dbg.cpu()
    t = t[t[:, 11] > 0]
    d = torch.diff(torch.cat([torch.zeros(t.shape[0], 1), t[:, :12]], 1), dim=1)
    start = t[:, 12]
    print(f"{wl} layer {layer}: {t.shape[0]} waves, total mean {t[:, 11].mean():.0f} max {t[:, 11].max():.0f} cycles; start spread {((start - start.min()) % 2**20).max() * 64:.0f} cycles")
    for i, nm in enumerate(names):
        print("   %-56s mean %7.0f  max %7.0f  (%4.1f %%)" % (nm, d[:, i].mean(), d[:, i].max(), 100 * d[:, i].mean() / t[:, 11].mean()))
l.pp_debug_set_dbg(None)
