"""Run the layer-0 fused edge update repeatedly on identical inputs and compare its outputs bit for bit."""
import os, sys, ctypes as C
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from packppi_amd import synth, lib
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
b = protein_to_batch(synth.make_complex(L, 5)).to("cuda:0")
ctx = m._context(b)
t = torch.full((L,), 0.5, device="cuda:0")
m.network(b, b.SC_D, t)            # leaves PAe / PCe / ptsE etc. of layer... (state of the last evaluation)
l = lib.load()
l.pp_debug_edge.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
l.pp_debug_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
K = min(32, L)
LAYER = int(sys.argv[2]) if len(sys.argv) > 2 else 0
l.pp_debug_set_hE.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
hE_in = torch.empty(L * K * 128, device="cuda:0")
assert l.pp_debug_buffer(ctx.handle, 0, C.c_void_p(hE_in.data_ptr()), hE_in.numel()) == 0
def run():
    assert l.pp_debug_set_hE(ctx.handle, C.c_void_p(hE_in.data_ptr()), hE_in.numel()) == 0
    assert l.pp_debug_edge(ctx.handle, LAYER, None) == 0
    hE = torch.empty(L * K * 128, device="cuda:0"); S = torch.empty(L * 128, device="cuda:0")
    assert l.pp_debug_buffer(ctx.handle, 0, C.c_void_p(hE.data_ptr()), hE.numel()) == 0
    assert l.pp_debug_buffer(ctx.handle, 1, C.c_void_p(S.data_ptr()), S.numel()) == 0
    D = torch.empty(L * 4 * 64 * 4, device="cuda:0")
    if os.environ.get("PP_DEBUG_TAIL"):
        assert l.pp_debug_buffer(ctx.handle, 5, C.c_void_p(D.data_ptr()), D.numel()) == 0
    return hE.cpu().reshape(L, K, 128), S.cpu().reshape(L, 128), D.cpu().reshape(L, 4, 64, 4)
ref = run()
for rep in range(8):
    hE, S, D = run()
    if os.environ.get("PP_DEBUG_TAIL"):
        dd = (D != ref[2])
        for name, i in (("nbr", 0), ("acc_init", 1), ("geom", 2), ("x", 3)):
            bad = torch.nonzero(dd[..., i].reshape(L, -1).any(1)).flatten().tolist()
            if bad:
                r = bad[0]; w = torch.nonzero(dd[r, :, :, i].any(1)).flatten().tolist()
                print("    DBG %s differs in residues %s; residue %d waves %s lanes %d" % (name, bad[:8], r, w, int(dd[r, :, :, i].sum())))
    dE = (hE != ref[0]); dS = (S != ref[1])
    resE = torch.nonzero(dE.reshape(L, -1).any(1)).flatten().tolist()
    resS = torch.nonzero(dS.any(1)).flatten().tolist()
    print("rep", rep, "residues with differing hE:", len(resE), resE[:6], "| differing S:", len(resS), resS[:6])
    for r in resE[:3]:
        d = dE[r]
        print("    residue", r, ": edges differing", torch.nonzero(d.any(1)).flatten().tolist()[:40], "features differing (count per 32-tile)", [int(d[:, 32*t:32*t+32].sum()) for t in range(4)],
              "max abs diff %.3e" % (hE[r] - ref[0][r]).abs().max())
    for r in [q for q in resS if q not in resE][:3]:
        print("    S residue", r, "features differing per tile", [int(dS[r, 32*t:32*t+32].sum()) for t in range(4)], "max %.3e" % (S[r] - ref[1][r]).abs().max())
