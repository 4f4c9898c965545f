"""Layer-0 node message with per-lane probes: which intermediate differs when several workgroups share a CU?"""
import os, sys, ctypes as C
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from packppi_amd import synth, lib
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
LAYER = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
b = protein_to_batch(synth.make_complex(L, 5)).to("cuda:0")
ctx = m._context(b)
l = lib.load()
l.pp_debug_set_lds_pad.argtypes = [C.c_int]; l.pp_debug_set_lds_pad.restype = None
l.pp_debug_set_edge_R.argtypes = [C.c_int]; l.pp_debug_set_edge_R.restype = None
l.pp_debug_set_dbg.argtypes = [C.c_void_p]; l.pp_debug_set_dbg.restype = None
l.pp_debug_set_edge_R(1)
l.pp_debug_set_lds_pad(84 * 1024)
m.network(b, b.SC_D, torch.full((L,), 0.5, device="cuda:0"))
l.pp_debug_nm.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
dbg = torch.zeros(L, 4, 64, 8, device="cuda:0")
l.pp_debug_set_dbg(C.c_void_p(dbg.data_ptr()))
names = ["nbr", "acc_init", "geom", "layer1", "x_all", "acc_final", "s_half", "s_full"]
def run(pad):
    l.pp_debug_set_lds_pad(pad)
    dbg.zero_()
    assert l.pp_debug_nm(ctx.handle, LAYER, None) == 0
    torch.cuda.synchronize()
    return dbg.cpu().clone()
ref = run(84 * 1024)
assert (run(84 * 1024) == ref).all()
for rep in range(6):
    o = run(0)
    dd = o != ref
    bad = torch.nonzero(dd.reshape(L, -1).any(1)).flatten().tolist()
    print("rep", rep, "residues differing:", bad[:12])
    for r in bad[:3]:
        for q, nm in enumerate(names):
            w = dd[r, :, :, q]
            if w.any():
                waves = torch.nonzero(w.any(1)).flatten().tolist()
                lanes = torch.nonzero(w.any(0)).flatten().tolist()
                print("     residue %d probe %-9s waves %s lanes %s" % (r, nm, waves, lanes if len(lanes) < 40 else str(lanes[:40]) + "..."))
