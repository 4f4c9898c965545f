"""Per-stage clock of wave 0 through 16 consecutive stages of the edge update (build with --tag: -DPP_LAB -DPP_X_TS
-DPP_X_TS_FINE=k0; run with PP_EDGE_R=2): one two-residue workgroup per CU (512 residues) and two (1 024)."""
import os, sys, ctypes as C
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from packppi_amd import lib, synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
k0 = int(sys.argv[1]) if len(sys.argv) > 1 else 19
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
l = lib.load()
l.pp_debug_set_dbg.argtypes = [C.c_void_p]; l.pp_debug_set_dbg.restype = None
l.pp_debug_edge.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for L in (512, 1024):
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
    t = dbg.cpu()[:, :16]
    t = t[t[:, 15] > 0]
    d = torch.diff(t, dim=1)
    print("L = %d, %d workgroups: cycles from the end of stage k-1 to the end of stage k (layer 1: FFN block c = stages 15 + 8c .. 22 + 8c, W1 x4 | W2 x4)" % (L, t.shape[0]))
    print("   " + "  ".join("%d:%5.0f" % (k0 + 1 + i, d[:, i].mean()) for i in range(15)))
