"""Wall time of the 50-step proximal stage on T1124 (one synchronised call) against the sum of its kernels' durations."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from bench import load_t1124
from packppi_amd.functional import _ctx_for, proximal_optimizer
b, init, ref = load_t1124()
gb = b.to("cuda:0")
chi = torch.from_numpy(ref).to("cuda:0") if not torch.is_tensor(ref) else ref.to("cuda:0")
ctx = _ctx_for(gb)
for _ in range(3):
    ctx.proximal(chi, 12.0, 0.5, 1.0, 50)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    ctx.proximal(chi, 12.0, 0.5, 1.0, 50)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("pp_proximal (50 steps), enqueue + run: %.3f ms per call" % ((t1 - t0) / 20 * 1e3))
t0 = time.perf_counter()
for _ in range(20):
    proximal_optimizer(gb, chi, 12.0, 0.5, 1.0, 50)
torch.cuda.synchronize()
t1 = time.perf_counter()
print("proximal_optimizer() incl. loss list: %.3f ms per call" % ((t1 - t0) / 20 * 1e3))
