"""Run one node-message launch repeatedly on identical inputs and compare S bit for bit."""
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
m.network(b, b.SC_D, torch.full((L,), 0.5, device="cuda:0"))
l = lib.load()
l.pp_debug_nm.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
l.pp_debug_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
def run(layer):
    assert l.pp_debug_nm(ctx.handle, layer, None) == 0
    S = torch.empty(L * 128, device="cuda:0")
    assert l.pp_debug_buffer(ctx.handle, 1, C.c_void_p(S.data_ptr()), S.numel()) == 0
    return S.cpu().reshape(L, 128)
for layer in (0, 1, 2):
    outs = [run(layer) for _ in range(10)]
    ref = torch.stack(outs).median(0).values
    bad = [torch.nonzero((o != ref).any(1)).flatten().tolist() for o in outs]
    print("layer", layer, "residues differing from the median per run:", [len(x) for x in bad], [x[:4] for x in bad if x][:3])
    for o, x in zip(outs, bad):
        if x:
            r = x[0]; d = (o[r] != ref[r])
            print("    residue", r, "features differing per tile", [int(d[32*t:32*t+32].sum()) for t in range(4)], "max %.2e" % (o[r]-ref[r]).abs().max())
            break
