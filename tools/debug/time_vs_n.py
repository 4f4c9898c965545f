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
    ctx.time_kernel(1, 10); ctx.time_kernel(0, 10)
    print("L=%d  edge %.1f us  node_msg %.1f us" % (L, ctx.time_kernel(1, 30) * 1e3, ctx.time_kernel(0, 30) * 1e3), flush=True)
