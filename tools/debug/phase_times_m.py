"""Per-phase time of the one-residue-per-wave edge update (build with -DPP_X_TS, run with PP_EDGE_IMPL=m): mean over waves of the
cycle-counter stamps."""
import os, sys, ctypes as C
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from bench import load_t1124
from packppi_amd import lib
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
b, init, ref = load_t1124()
b = b.to("cuda:0"); L = b.X.shape[1]
ctx = m._context(b)
l = lib.load()
l.pp_debug_set_dbg.argtypes = [C.c_void_p]; l.pp_debug_set_dbg.restype = None
l.pp_debug_edge.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
m.network(b, b.SC_D, torch.full((L,), 0.5, device="cuda:0"))
dbg = torch.zeros(L, 24, device="cuda:0")
l.pp_debug_set_dbg(C.c_void_p(dbg.data_ptr()))
names = ["prologue", "layer 1 (gathers, W_B, geometry)", "layer 2", "layer 3", "residual + LN2 + split", "FFN tiles 0-3", "FFN 4-7", "FFN 8-11",
         "FFN 12-15", "LN3 + store", "fused nm: split, geometry", "fused nm: first layer", "fused nm: last layer + reduce"]
for layer in (1,):
    for rep in range(3):
        dbg.zero_()
        assert l.pp_debug_edge(ctx.handle, layer, None) == 0
        torch.cuda.synchronize()
    t = dbg.cpu()[:, :13]
    t = t[t[:, 12] > 0]
    d = torch.diff(torch.cat([torch.zeros(t.shape[0], 1), t], 1), dim=1)
    print("layer %d: %d waves, mean total %.0f cycles" % (layer, t.shape[0], t[:, 12].mean()))
    for i, nm in enumerate(names):
        print("   %-40s %7.0f cycles  (%4.1f %%)" % (nm, d[:, i].mean(), 100 * d[:, i].mean() / t[:, 12].mean()))
