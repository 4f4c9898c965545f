import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from bench import load_t1124
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
b, init, ref = load_t1124()
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
ctx = m._context(b.to("cuda:0"))
chi = ctx.sample(init.to("cuda:0"), torch.linspace(1, 0, 4))
for it in range(3):
    print("edge %.1f us  node_msg %.1f us" % (ctx.time_kernel(1, 50) * 1e3, ctx.time_kernel(0, 50) * 1e3))
