"""Per-phase time of the fused edge update (build with --tag ts -DPP_LAB -DPP_X_TS, load with PACKPPI_ALLOW_LAB_LIBRARY=1): mean over workgroups of wave 0's
s_memtime stamps (core-clock cycles, ~2.1 GHz under load)."""
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
names = ["prologue (loads, geometry, x split)", "first layer + publish", "fetch + second layer", "publish 2", "fetch + third layer",
         "residual + LN2 exchange", "LN2 (stats, affine, split)", "FFN block 0", "FFN block 1", "FFN block 2", "FFN block 3",
         "residual + LN3 exchange", "LN3 + store h_E", "tail: publish, geometry, fetch", "tail first layer", "tail publish + fetch", "tail second layer", "reduce + store S"]
for layer in (0, 1):
    for rep in range(3):
        dbg.zero_()
        assert l.pp_debug_edge(ctx.handle, layer, None) == 0
        torch.cuda.synchronize()
    t_all = dbg.cpu()[:, :18]
    npairs = (L + 2) // 3
    rows = torch.arange(L)
    groups = {"two-residue workgroups / team 0": (rows < 2 * npairs) & (t_all[:, 17] > 0),
              "one-residue workgroups / team 1": (rows >= 2 * npairs) & (t_all[:, 17] > 0)}
    for gname, sel in groups.items():
        t = t_all[sel]
        if not len(t):
            continue
        d = torch.diff(torch.cat([torch.zeros(t.shape[0], 1), t], 1), dim=1)
        print("layer %d, %s: %d, mean total %.0f cycles" % (layer, gname, t.shape[0], t[:, 17].mean()))
        for i, nm in enumerate(names):
            print("   %-40s %7.0f cycles  (%4.1f %%)" % (nm, d[:, i].mean(), 100 * d[:, i].mean() / t[:, 17].mean()))
