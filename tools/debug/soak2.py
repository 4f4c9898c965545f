"""Soak, part 2: padded multi-complex batches, tiny complexes (K < 32), SDE mode -- bit-reproducibility run to run."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from collections import Counter
from packppi_amd import synth
from packppi_amd.batch import collate
from packppi_amd.featurize import protein_to_data, protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
sched = torch.linspace(1, 0, 21)
def check(name, b, mode="ode", reps=8):
    ctx = m._context(b)
    B, L = b.residue_type.shape
    g = torch.Generator().manual_seed(7)
    init = ((torch.rand(B, L, 4, generator=g) * 2 - 1) * 3.0 * b.SC_D_mask.cpu()).to("cuda:0")
    noise = None
    if mode == "sde":
        noise = torch.randn(len(sched) - 1, 2, B * L, 4, generator=g).to("cuda:0")
    outs = [ctx.sample(init, sched, mode=mode, sde_noise=noise).cpu() for _ in range(reps)]
    cnt = Counter(o.numpy().tobytes() for o in outs)
    ok = len(cnt) == 1 and all(torch.isfinite(o).all() for o in outs)
    print("%-34s %s (%d distinct outputs)" % (name, "ok" if ok else "FAIL", len(cnt)), flush=True)
    return ok
ok = True
ok &= check("B=24 x L~U(250,330) padded", collate([protein_to_data(synth.make_complex(250 + 7 * (i % 12), 100 + i)) for i in range(24)]).to("cuda:0"))
ok &= check("B=5 x L=12..28 (K < 32)", collate([protein_to_data(synth.make_complex(12 + 4 * i, 300 + i)) for i in range(5)]).to("cuda:0"))
ok &= check("L=9 single", protein_to_batch(synth.make_complex(9, 5)).to("cuda:0"))
ok &= check("L=900 SDE", protein_to_batch(synth.make_complex(900, 6)).to("cuda:0"), mode="sde")
print("SOAK2", "OK" if ok else "FAIL")
