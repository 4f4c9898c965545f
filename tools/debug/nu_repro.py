"""Bit-reproducibility probe of the node-update kernel variants (MID / SCORE / STEP with and without the embedding tail)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from packppi_amd import synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict

DEV = "cuda:0"
m = TDiffusionModule(make_random_state_dict(20251003), device=DEV)
for L in ((400,) if os.environ.get("NU_QUICK") else (64, 400, 739)):
    b = protein_to_batch(synth.make_complex(L, 11)).to(DEV)
    ctx = m._context(b)
    g = torch.Generator().manual_seed(L)
    init = ((torch.rand(1, L, 4, generator=g) * 2 - 1) * 3.0 * b.SC_D_mask.cpu()).to(DEV)
    for name, fn in (("score", lambda: torch.cat([x.flatten() for x in ctx.score(init, 0.5)])),
                     ("sample1", lambda: ctx.sample(init, torch.linspace(1, 0.9, 2))),
                     ("sample2", lambda: ctx.sample(init, torch.linspace(1, 0.8, 3))),
                     ("sample12", lambda: ctx.sample(init, torch.linspace(1, 0, 13)))):
        ref = fn().cpu()
        bad, worst = 0, 0.0
        for _ in range(20):
            x = fn().cpu()
            if not torch.equal(x, ref):
                bad += 1
                worst = max(worst, float((x - ref).abs().max()))
        print(f"L={L} {name}: {bad}/20 runs differ, worst {worst:.3e}", flush=True)
