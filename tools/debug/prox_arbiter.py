"""Proximal stage vs the fp64 arbiter: |HIP - ref64| against |ref32 - ref64| after 1, 5, 10, 20, 50 Adam steps."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import torch
from tests.conftest import load_golden, wrapped_absdiff
from packppi_amd.functional import proximal_optimizer

DEV = "cuda:0"
for tag in ("L64", "L120", "T1124", "S1500"):
    z = np.load(os.path.join("tests", "golden", f"g6_prox_{tag}.npz"))
    b, g = load_golden(str(z["source_fixture"]))
    chi0 = g[str(z["chi0_key"])].float().to(DEV)
    gb = b.to(DEV)
    chis, losses = proximal_optimizer(gb, chi0, 12.0, 0.5, 1.0, 50)
    l32 = z["losses32"]
    print(f"{tag}: loss rel err vs ref32 max {np.max(np.abs(np.array(losses) - l32) / np.abs(l32)):.2e}"
          + (f", ref64 vs ref32 {np.max(np.abs(z['losses64'] - l32) / np.abs(l32)):.2e}" if "losses64" in z else ""))
    for n in (1, 5, 10, 20, 50):
        h = chis[n - 1].cpu().double()
        r32 = torch.from_numpy(z[f"chi32_step{n}"]).double()
        line = f"  step {n:2d}: |hip-ref32| {float(wrapped_absdiff(h, r32).max()):.2e}"
        if f"chi64_step{n}" in z:
            r64 = torch.from_numpy(z[f"chi64_step{n}"]).double()
            d_h, d_r = wrapped_absdiff(h, r64), wrapped_absdiff(r32, r64)
            line += f"  |hip-ref64| {float(d_h.max()):.2e}  |ref32-ref64| {float(d_r.max()):.2e}  n(>1e-4): hip {int((d_h > 1e-4).sum())} ref32 {int((d_r > 1e-4).sum())}"
        print(line)
