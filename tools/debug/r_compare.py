"""Split-f16 edge kernels: a residue's result must not depend on how many residues share its workgroup.
Runs the node message (layers 0-2) and the edge update (layers 0-1) with R = 1, 2, 3 residues per workgroup on identical
inputs and compares bit for bit against R = 1."""
import os, sys, ctypes as C
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from packppi_amd import synth, lib
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
L = int(sys.argv[1]) if len(sys.argv) > 1 else 301
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
b = protein_to_batch(synth.make_complex(L, 5)).to("cuda:0")
ctx = m._context(b)
l = lib.load()
l.pp_debug_set_edge_R.argtypes = [C.c_int]
l.pp_debug_set_edge_R.restype = None
l.pp_debug_set_edge_R(1)
m.network(b, b.SC_D, torch.full((L,), 0.5, device="cuda:0"))
l.pp_debug_nm.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
l.pp_debug_edge.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
l.pp_debug_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
l.pp_debug_set_hE.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
K = min(32, L)
def buf(which, n):
    t = torch.empty(n, device="cuda:0")
    assert l.pp_debug_buffer(ctx.handle, which, C.c_void_p(t.data_ptr()), n) == 0
    return t
hE_saved = buf(0, L * K * 128).clone()
def run_nm(layer, R):
    l.pp_debug_set_edge_R(R)
    assert l.pp_debug_set_hE(ctx.handle, C.c_void_p(hE_saved.data_ptr()), hE_saved.numel()) == 0
    assert l.pp_debug_nm(ctx.handle, layer, None) == 0
    return buf(1, L * 128).cpu().reshape(L, 128), buf(2, L).cpu()
def run_eu(layer, R):
    l.pp_debug_set_edge_R(R)
    assert l.pp_debug_set_hE(ctx.handle, C.c_void_p(hE_saved.data_ptr()), hE_saved.numel()) == 0
    assert l.pp_debug_edge(ctx.handle, layer, None) == 0
    return buf(0, L * K * 128).cpu().reshape(L, K * 128), buf(1, L * 128).cpu().reshape(L, 128)
def report(name, ref, o):
    bad = torch.nonzero((o != ref).reshape(L, -1).any(1)).flatten().tolist()
    msg = "ok" if not bad else "DIFF residues %d (first %s) max %.3e" % (len(bad), bad[:8], (o - ref).abs().max())
    print("   ", name, msg)
    if bad and o.shape[1] == 128:
        r = bad[0]; d = o[r] != ref[r]
        print("        residue", r, "features differing per tile", [int(d[32 * t:32 * t + 32].sum()) for t in range(4)])
for layer in (0, 1, 2):
    refS, refm = run_nm(layer, 1)
    for R in (1, 2, 3):
        for rep in range(2):
            S, ms = run_nm(layer, R)
            print("node message layer", layer, "R", R)
            report("S", refS, S); report("msum", refm.reshape(L, 1), ms.reshape(L, 1))
for layer in (0, 1):
    refE, refS = run_eu(layer, 1)
    for R in (1, 2, 3):
        for rep in range(2):
            E, S = run_eu(layer, R)
            print("edge update layer", layer, "R", R)
            report("hE", refE, E); report("S(fused)", refS, S)
