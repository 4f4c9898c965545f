"""Per-slot time of the rotation edge kernel (library built with -DPP_X_TS): mean over workgroups of thread 0's core-clock
stamps at the end of every slot (after its barrier), relative to the end of the prologue."""
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
l.pp_debug_set_edge_rot.argtypes = [C.c_int]; l.pp_debug_set_edge_rot.restype = None
l.pp_debug_edge.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
MODE = int(os.environ.get('ROT_MODE', '1')); R = 3 if MODE == 1 else 2
l.pp_debug_set_edge_rot(MODE)
m.network(b, b.SC_D, torch.full((L,), 0.5, device="cuda:0"))
nwg = (L + R - 1) // R
dbg = torch.zeros(nwg, 64, device="cuda:0")
l.pp_debug_set_dbg(C.c_void_p(dbg.data_ptr()))
grp = {0: ["geo", "L2", "L3"] + [f"ffn{'io'[i & 1]}{i >> 1}" for i in range(8)] + ["nmE", "nmG", "nm2"],
       1: ["hE", "geo", "L2", "L3"] + [f"ffn{'io'[i & 1]}{i >> 1}" for i in range(8)] + ["nmE", "nmG", "nm2"]}
for layer in (0, 1):
    for rep in range(3):
        dbg.zero_()
        assert l.pp_debug_edge(ctx.handle, layer, None) == 0
        torch.cuda.synchronize()
    t = dbg.cpu()
    ns = int((t[0] > 0).sum())
    t = t[:, :ns]
    d = torch.diff(torch.cat([torch.zeros(t.shape[0], 1), t], 1), dim=1).mean(0)
    print(f"layer {layer}: {nwg} workgroups, {ns} slots, mean total after prologue {t[:, -1].mean():.0f} cycles")
    names = grp[layer]
    for g in range((ns + R - 1) // R):
        row = [f"{d[R * g + r]:6.0f}" for r in range(R) if R * g + r < ns]
        print(f"   group {g:2d} {names[g] if g < len(names) else 'drain':6s}", " ".join(row))
