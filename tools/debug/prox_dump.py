"""Dump the HIP proximal trajectory (and per-step clash gradient) of a fixture for offline comparison with the reference."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np
import torch
from tests.conftest import load_golden
from packppi_amd.functional import proximal_optimizer, _ctx_for

DEV = "cuda:0"
tag = sys.argv[1] if len(sys.argv) > 1 else "L120"
z = np.load(os.path.join("tests", "golden", f"g6_prox_{tag}.npz"))
b, g = load_golden(str(z["source_fixture"]))
chi0 = g[str(z["chi0_key"])].float().to(DEV)
gb = b.to(DEV)
chis, losses = proximal_optimizer(gb, chi0, 12.0, 0.5, 1.0, 50)
traj = torch.stack([c.cpu() for c in chis]).numpy()
# clash value + gradient at the reference's own fp32 iterates (steps 10 and 20): isolates the gradient from the trajectory
out = {"traj": traj, "losses": np.array(losses)}
ctx = _ctx_for(gb)
for n in (10, 20):
    x = torch.from_numpy(z[f"chi32_step{n}"]).float().to(DEV)
    pr, dchi = ctx.clash(x, 12.0, 0.5, need_grad=True)
    out[f"pr_at_ref{n}"] = pr.cpu().numpy()
    out[f"dchi_at_ref{n}"] = dchi.cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.savez_compressed(f"gpurun_out/prox_{tag}_hip.npz", **out)
print("wrote", traj.shape)
