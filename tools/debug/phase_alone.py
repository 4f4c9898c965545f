"""Phase stamps of the two-residue edge-update workgroup with the CU to itself and with a neighbour (build with
python -m packppi_amd.build --tag ts -DPP_LAB -DPP_X_TS; PACKPPI_LIB=...ts.so PACKPPI_ALLOW_LAB_LIBRARY=1 PP_EDGE_R=2): a synthetic complex of 512 residues gives 256 workgroups (one per CU), 1 024 residues 512
(two per CU).  Mean over workgroups of wave 0's stamps, core-clock cycles."""
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
    t = dbg.cpu()[:, :18]
    t = t[t[:, 17] > 0]
    d = torch.diff(torch.cat([torch.zeros(t.shape[0], 1), t], 1), dim=1)
    print("L = %d: %d workgroups stamped, mean total %.0f cycles" % (L, t.shape[0], t[:, 17].mean()))
    for i, nm in enumerate(names):
        print("   %-34s %7.0f" % (nm, d[:, i].mean()))
