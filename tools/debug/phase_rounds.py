"""First-round against later-round workgroups of a multi-round edge-update launch (two-residue workgroups throughout): the first
round finds the instruction caches cold (every launch starts with them invalidated; the kernel is ~28 KB of straight-line code
per instance), later rounds find the code resident.  Build: python -m packppi_amd.build --tag ts -DPP_LAB -DPP_X_TS ; run with
PACKPPI_LIB=...ts.so PACKPPI_ALLOW_LAB_LIBRARY=1."""
import os, sys, ctypes as C
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from packppi_amd import lib, synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
l = lib.load()
l.pp_debug_set_dbg.argtypes = [C.c_void_p]; l.pp_debug_set_dbg.restype = None
l.pp_debug_edge.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
names = ["prologue", "first layer + publish", "second layer", "publish 2", "third layer", "residual + LN2 exchange", "LN2",
         "FFN block 0", "FFN block 1", "FFN block 2", "FFN block 3", "residual + LN3 exchange", "LN3 + store h_E",
         "tail: publish, geometry, fetch", "tail first layer", "tail publish + fetch", "tail second layer", "reduce + store S"]
for L in (int(a) for a in (sys.argv[1:] or ["2048", "3072"])):
    b = protein_to_batch(synth.make_complex(L, 77)).to("cuda:0")
    ctx = m._context(b)
    m.network(b, b.SC_D, torch.full((L,), 0.5, device="cuda:0"))
    dbg = torch.zeros(L, 24, device="cuda:0")
    l.pp_debug_set_dbg(C.c_void_p(dbg.data_ptr()))
    for rep in range(3):
        dbg.zero_()
        assert l.pp_debug_edge(ctx.handle, 1, None) == 0
        torch.cuda.synchronize()
    l.pp_debug_set_dbg(None)
    t = dbg.cpu()
    live = t[:, 17] > 0
    rows = torch.arange(L)
    d = torch.diff(torch.cat([torch.zeros(L, 1), t[:, :18]], 1), dim=1)
    # workgroup g owns residues 2g, 2g + 1 and is dispatched in index order: the first 512 fill the chip (two per CU), the others
    # take slots as they free up (the per-CU clocks are not synchronised across XCDs, so the index is the classifier)
    first = live & (rows < 1024)
    later = live & (rows >= 1024)
    print(f"L = {L}: {int(live.sum())} workgroups stamped; first round: {int(first.sum())} (mean total {t[first, 17].mean():.0f}, "
          f"p90 {torch.quantile(t[first, 17], 0.9):.0f}), later: {int(later.sum())} (mean total {t[later, 17].mean():.0f}, p90 {torch.quantile(t[later, 17], 0.9):.0f})")
    for i, nm in enumerate(names):
        print("   %-34s %7.0f %7.0f" % (nm, d[first, i].mean(), d[later, i].mean()))
