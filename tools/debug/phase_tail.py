"""Distribution of the per-workgroup duration of the fused edge update (T1124, mixed launch): which workgroups end the launch.
Build: python -m packppi_amd.build --tag ts -DPP_LAB -DPP_X_TS ; run with PACKPPI_LIB=...ts.so PACKPPI_ALLOW_LAB_LIBRARY=1."""
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
npairs = (L + 2) // 3
for layer in (0, 1):
    for rep in range(3):
        dbg.zero_()
        assert l.pp_debug_edge(ctx.handle, layer, None) == 0
        torch.cuda.synchronize()
    t = dbg.cpu()
    tot, pro = t[:, 17], t[:, 0]
    rows = torch.arange(L)
    for name, sel in (("pairs", (rows < 2 * npairs) & (tot > 0)), ("singles", (rows >= 2 * npairs) & (tot > 0))):
        x = tot[sel]
        q = torch.quantile(x, torch.tensor([0.0, 0.1, 0.5, 0.9, 0.99, 1.0]))
        p = pro[sel]
        print(f"layer {layer} {name:8s} n {int(sel.sum()):4d} total cycles min/p10/p50/p90/p99/max " + " ".join(f"{v:7.0f}" for v in q.tolist())
              + f"   prologue p50 {p.median():6.0f} max {p.max():6.0f}")
    # the slowest ten workgroups: residue row, kind, per-phase durations
    idx = torch.argsort(tot, descending=True)[:8]
    for i in idx.tolist():
        d = torch.diff(torch.cat([torch.zeros(1), t[i, :18]]))
        print(f"   row {i:4d} {'pair' if i < 2 * npairs else 'single':6s} total {tot[i]:7.0f}: " + " ".join(f"{v:5.0f}" for v in d.tolist()))
