import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))
import torch
from bench import load_t1124
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
from packppi_amd import lib
print("variant", lib.load().pp_edge_variant(), os.environ.get("PACKPPI_LIB"), "PP_NODE_F16", os.environ.get("PP_NODE_F16"))
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
b, init, ref = load_t1124(); b = b.to("cuda:0"); init = init.to("cuda:0")
m.schedule = torch.linspace(1, 0, 101)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.time(); out = m.sample_from(b, init); torch.cuda.synchronize(); dt = time.time() - t0
    print("T1124 100 steps: %.2f ms/step" % (dt * 10))
if ref is not None:
    d = (out.cpu() - ref + torch.pi) % (2 * torch.pi) - torch.pi
    print("max |d| vs reference %.3e" % float(d.abs().max()))
