"""How long does it take to set up a complex (pp_complex_prepare: allocations + kNN + edge embedding + static term)?"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from bench import load_t1124
from packppi_amd.lib import Context
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
b, init, ref = load_t1124()
t0 = time.perf_counter()
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
torch.cuda.synchronize(); print("module + plan: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
bd = b.to("cuda:0")
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ctx = Context(m._plan, bd)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    del ctx
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("ctx create %.2f ms, destroy %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
