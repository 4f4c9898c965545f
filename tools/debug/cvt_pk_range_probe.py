import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
from packppi_amd import lib, synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
L = lib.load()
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
b = protein_to_batch(synth.make_complex(300, 11)).to("cuda:0")
e, n = C.c_ulonglong(0), C.c_ulonglong(0)
L.pp_range_check_parts(C.byref(e), C.byref(n), 1)
g = torch.Generator().manual_seed(300)
init = ((torch.rand(1, 300, 4, generator=g) * 2 - 1) * 3.0 * b.SC_D_mask.cpu()).to("cuda:0")
ctx = m._context(b)
outs = []
for i in range(3):
    out = ctx.sample(init, torch.linspace(1, 0, 21))
    L.pp_range_check_parts(C.byref(e), C.byref(n), 1)
    outs.append(out.cpu())
    print("run", i, "range events edge", e.value, "node", n.value, "finite", bool(torch.isfinite(out).all()), "equal to run 0", bool(torch.equal(outs[0], outs[-1])))
