import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from packppi_amd import synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
Ls = [int(a) for a in sys.argv[1:]] or [128, 256, 384, 512, 640, 768, 1024, 1536]
for L in Ls:
    b = protein_to_batch(synth.make_complex(L, 77)).to("cuda:0")
    ctx = m._context(b)
    chi = ctx.sample(b.SC_D, torch.linspace(1, 0, 3))
    sched = torch.linspace(1, 0, 21)
    ctx.sample(b.SC_D, sched)
    r = []
    for which in (1, 0, 2):
        ctx.profile_kernel(which); ctx.sample(b.SC_D, sched); r.append(ctx.profile_read()[0] * 1e3)
    print("L=%d  edge %.1f us  node_msg %.1f us  node_upd %.1f us (in situ)" % (L, r[0], r[1], r[2]), flush=True)
