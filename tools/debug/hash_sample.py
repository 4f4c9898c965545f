import hashlib, os, sys
sys.path.insert(0, os.getcwd())
import torch
from bench import load_t1124, load_s1500
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
m.schedule = torch.linspace(1, 0, 101)
for name, load in (("T1124", load_t1124), ("S1500", load_s1500)):
    b, init, ref = load()
    out = m.sample_from(b.to("cuda:0"), init.to("cuda:0"))
    print(os.path.basename(os.environ.get("PACKPPI_LIB", "libpackppi_hip.so")), name, hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16])
