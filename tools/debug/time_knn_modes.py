"""Device time of pp_complex_prepare (events around 20 context creations) under the three kNN tie modes, for T1124, S1500
and the 32-complex packed share of config 4: what the reference-CPU tie emulation costs."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from bench import c5_complexes, load_s1500, load_t1124
from packppi_amd.batch import pack
from packppi_amd.lib import Context
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
sd = make_random_state_dict(20251003)
loads = {"t1124": load_t1124()[0], "s1500": load_s1500()[0], "c5": pack(c5_complexes(0, "cpu"))}
for mode in ("lower_index", "aten_member", "aten_cpu"):
    m = TDiffusionModule(sd, device="cuda:0", knn_ties=mode)
    for name, b in loads.items():
        bd = b.to("cuda:0")
        for rep in range(3):
            ctx = Context(m._plan, bd); del ctx
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for rep in range(20):
            ctx = Context(m._plan, bd); del ctx
        e1.record(); torch.cuda.synchronize()
        print(f"{mode:12s} {name:6s} prepare {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
