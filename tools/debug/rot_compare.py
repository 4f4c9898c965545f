"""Rotation edge kernel (k_edge_update_rot, three residues per workgroup) against k_edge_update<1>: same inputs, the new
h_E, the fused node message S and msum must agree bit for bit, for sizes that are and are not multiples of three."""
import os, sys, ctypes as C
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from packppi_amd import synth, lib
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
l = lib.load()
l.pp_debug_set_edge_R.argtypes = [C.c_int]; l.pp_debug_set_edge_R.restype = None
l.pp_debug_set_edge_rot.argtypes = [C.c_int]; l.pp_debug_set_edge_rot.restype = None
l.pp_debug_edge.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
l.pp_debug_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
l.pp_debug_set_hE.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
bad_total = 0
MODE = int(os.environ.get("ROT_MODE", "1"))      # PP_EDGE_ROT value under test
for L in [int(a) for a in sys.argv[1:]] or [301, 100, 739, 64]:
    b = protein_to_batch(synth.make_complex(L, 5)).to("cuda:0")
    ctx = m._context(b)
    l.pp_debug_set_edge_R(1); l.pp_debug_set_edge_rot(0)
    m.network(b, b.SC_D, torch.full((L,), 0.5, device="cuda:0"))
    K = min(32, L)
    def buf(which, n):
        t = torch.empty(n, device="cuda:0")
        assert l.pp_debug_buffer(ctx.handle, which, C.c_void_p(t.data_ptr()), n) == 0
        return t
    hE_saved = buf(0, L * K * 128).clone()
    def run(layer, rot):
        l.pp_debug_set_edge_R(0 if rot else 1); l.pp_debug_set_edge_rot(MODE if rot else 0)
        assert l.pp_debug_set_hE(ctx.handle, C.c_void_p(hE_saved.data_ptr()), hE_saved.numel()) == 0
        assert l.pp_debug_edge(ctx.handle, layer, None) == 0
        torch.cuda.synchronize()
        return buf(0, L * K * 128).cpu().reshape(L, K * 128), buf(1, L * 128).cpu().reshape(L, 128), buf(2, L).cpu().reshape(L, 1)
    for layer in (0, 1):
        ref = run(layer, False)
        for rep in range(2):
            got = run(layer, True)
            for name, r, g in zip(("h_E", "S", "msum"), ref, got):
                bad = torch.nonzero((g != r).any(1)).flatten().tolist()
                bad_total += len(bad)
                print(f"L {L} layer {layer} rep {rep} {name:5s}", "ok" if not bad else
                      f"DIFF residues {len(bad)} first {bad[:9]} max {(g - r).abs().max():.3e} nan {int(torch.isnan(g).sum())}", flush=True)
print("TOTAL differing rows", bad_total)
