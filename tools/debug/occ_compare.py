"""Same binary, same inputs: one workgroup per CU (LDS request padded) against several per CU (no padding).
Reports which residues / edges / features differ from the one-per-CU result."""
import os, sys, ctypes as C
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from packppi_amd import synth, lib
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
PAD = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
b = protein_to_batch(synth.make_complex(L, 5)).to("cuda:0")
ctx = m._context(b)
l = lib.load()
l.pp_debug_set_lds_pad.argtypes = [C.c_int]; l.pp_debug_set_lds_pad.restype = None
l.pp_debug_set_edge_R.argtypes = [C.c_int]; l.pp_debug_set_edge_R.restype = None
l.pp_debug_set_edge_R(1)
l.pp_debug_set_lds_pad(84 * 1024)
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
def run_nm(layer, pad):
    l.pp_debug_set_lds_pad(pad)
    assert l.pp_debug_set_hE(ctx.handle, C.c_void_p(hE_saved.data_ptr()), hE_saved.numel()) == 0
    assert l.pp_debug_nm(ctx.handle, layer, None) == 0
    return buf(1, L * 128).cpu().reshape(L, 128)
def run_eu(layer, pad):
    l.pp_debug_set_lds_pad(pad)
    assert l.pp_debug_set_hE(ctx.handle, C.c_void_p(hE_saved.data_ptr()), hE_saved.numel()) == 0
    assert l.pp_debug_edge(ctx.handle, layer, None) == 0
    return buf(0, L * K * 128).cpu().reshape(L, K, 128)
def ranges(xs):
    out = []; 
    for x in xs:
        if out and x == out[-1][1] + 1: out[-1][1] = x
        else: out.append([x, x])
    return ",".join("%d-%d" % (a, c) if a != c else "%d" % a for a, c in out[:12]) + (" ..." if len(out) > 12 else "")
for layer in (0, 1):
    ref = run_nm(layer, 84 * 1024)
    again = run_nm(layer, 84 * 1024)
    print("node message layer %d: one/CU repeat identical: %s" % (layer, bool((ref == again).all())))
    for rep in range(3):
        o = run_nm(layer, PAD)
        bad = torch.nonzero((o != ref).any(1)).flatten().tolist()
        print("   pad %d rep %d: %d residues differ: %s" % (PAD, rep, len(bad), ranges(bad)))
        if bad:
            r = bad[0]; d = o[r] != ref[r]
            print("      residue %d: features differing per tile %s  (features %s)" % (r, [int(d[32 * t:32 * t + 32].sum()) for t in range(4)], ranges(torch.nonzero(d).flatten().tolist())))
for layer in (0, 1):
    ref = run_eu(layer, 84 * 1024)
    for rep in range(3):
        o = run_eu(layer, PAD)
        dd = (o != ref)
        bad = torch.nonzero(dd.reshape(L, -1).any(1)).flatten().tolist()
        print("edge update layer %d pad %d rep %d: %d residues differ: %s" % (layer, PAD, rep, len(bad), ranges(bad)))
        if bad:
            r = bad[0]
            print("      residue %d: edges %s ; features %s ; max %.3e" % (r, ranges(torch.nonzero(dd[r].any(1)).flatten().tolist()),
                  ranges(torch.nonzero(dd[r].any(0)).flatten().tolist()), (o[r] - ref[r]).abs().max()))
