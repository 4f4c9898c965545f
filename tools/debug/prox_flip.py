"""Where does a proximal run leave the reference's fp32 run, and which hinge did it?  (tests/test_hip_parity.py::test_proximal
records the figures this prints; run on the GPU box:  python tools/debug/prox_flip.py [L64 L120 T1124 S1500])

For every fixture g6_prox_<tag>: the HIP path's 50 Adam steps against every step of the reference's fp32 run (traj32), the
first step at which the two differ by more than 1e-5 rad ("first flip"), the reference's own first flip between its fp32 and
fp64 runs (first_flip_ref), the distances after 50 steps, and -- at the HIP run's last agreeing iterate -- the atom pairs whose
clash hinge r_a + r_b - tol - d has a different sign on the two trajectories (clash.py:139-149)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)


def wrapped(a, b):
    d = (a.double() - b.double()).abs()
    return torch.minimum(d, (2 * np.pi - d).abs())


def overlaps_that_differ(batch, chi_a, chi_b, tol=0.5, rows=64, band=1e-4):
    """Atom pairs (i, a, j, b) of the between-residue clash term whose hinge argument r_a + r_b - tol - d is positive on one
    set of angles and not on the other; [(i, a, j, b, overlap_a, overlap_b)].  CPU, fp64, residue-pair matrix in row blocks."""
    from oracle import ref_cpu as O
    from packppi_amd import constants as rc
    dt = torch.float64
    bd = {k: (v.double().cpu() if isinstance(v, torch.Tensor) and v.dtype == torch.float32 else (v.cpu() if isinstance(v, torch.Tensor) else v))
          for k, v in batch.items()}
    S, exists, ridx = bd["residue_type"], bd["atom_mask"], bd["residue_index"]
    L = S.shape[1]
    radius = exists * torch.as_tensor(rc.between_radius, dtype=dt)[S]
    bb = torch.zeros(14, 14, dtype=dt); bb[:4, :4] = 1
    cn = torch.zeros(14, 14, dtype=dt); cn[2, 0] = 1
    ss = torch.zeros(14, 14, dtype=dt); ss[5, 5] = 1
    xa = O.atom14_coords(bd["X"], S, bd["BB_D"], chi_a.double().cpu())
    xb = O.atom14_coords(bd["X"], S, bd["BB_D"], chi_b.double().cpu())
    out = []
    for i0 in range(0, L, rows):
        i1 = min(L, i0 + rows)
        m = exists[:, i0:i1, None, :, None] * exists[:, None, :, None, :] * (1 - bb)
        m = m * (ridx[:, i0:i1, None, None, None] < ridx[:, None, :, None, None])
        nb = ((ridx[:, i0:i1, None] + 1) == ridx[:, None, :])[..., None, None]
        m = m * (1 - nb * cn) * (1 - ss)
        lower = radius[:, i0:i1, None, :, None] + radius[:, None, :, None, :] - tol
        ov = []
        for x in (xa, xb):
            d = torch.sqrt(1e-10 + ((x[:, i0:i1, None, :, None, :] - x[:, None, :, None, :, :]) ** 2).sum(-1))
            ov.append(lower - d)
        diff = (m > 0) & ((ov[0] > 0) != (ov[1] > 0)) & (torch.minimum(ov[0].abs(), ov[1].abs()) < band)
        for _, i, j, a, b in torch.nonzero(diff).tolist():
            out.append((i0 + i, a, j, b, float(ov[0][0, i, j, a, b]), float(ov[1][0, i, j, a, b])))
    return out


def report(tag, dev="cuda:0"):
    from tests.conftest import load_golden
    from packppi_amd.functional import proximal_optimizer
    z = np.load(os.path.join(ROOT, "tests", "golden", f"g6_prox_{tag}.npz"))
    b, g = load_golden(str(z["source_fixture"]))
    chi0 = g[str(z["chi0_key"])].float()
    chis, losses = proximal_optimizer(b.to(dev), chi0.to(dev), 12.0, 0.5, 1.0, 50)
    chis = [c.cpu() for c in chis]
    idx = torch.from_numpy(z["traj32_residues"].astype(np.int64))
    ref = torch.from_numpy(z["traj32"])                                  # [50, n_moved, 4]
    d = np.array([float(wrapped(c[0, idx], ref[n]).max()) for n, c in enumerate(chis)])
    flips = np.nonzero(d > 1e-5)[0]
    first = int(flips[0]) + 1 if len(flips) else 51
    d64 = float(wrapped(chis[-1], torch.from_numpy(z["chi64_step50"])).max())
    d32 = float(wrapped(chis[-1], torch.from_numpy(z["chi32_step50"])).max())
    print(f"{tag}: |HIP - ref32| by step " + " ".join(f"{x:.1e}" for x in d[[0, 4, 9, 14, 19, 24, 29, 39, 49]]))
    print(f"{tag}: first step with |HIP - ref32| > 1e-5: {first}; the reference's own fp32 / fp64 runs part at step {int(z['first_flip_ref'])}; "
          f"max before that {d[:max(min(first, int(z['first_flip_ref'])) - 1, 1)].max():.2e}")
    print(f"{tag}: after 50 steps |HIP - ref64| {d64:.2e}  |HIP - ref32| {d32:.2e}  |ref32 - ref64| {float(z['div_32_64'][-1]):.2e}; "
          f"entries of HIP beyond 1e-4 of ref64: {int((wrapped(chis[-1], torch.from_numpy(z['chi64_step50'])) > 1e-4).sum())}")
    if first <= 50 and b.residue_type.shape[1] <= 800:
        n = max(first - 1, 1)                       # the last iterate on which the two runs still agree: compare its hinges
        ref_full = chi0.clone()
        ref_full[0, idx] = ref[n - 1]
        for (i, a, j, bq, oa, ob) in overlaps_that_differ(b, chis[n - 1], ref_full)[:6]:
            print(f"{tag}:   hinge (res {i} atom {a}) - (res {j} atom {bq}) at iterate {n}: overlap {oa:+.2e} A here, {ob:+.2e} A in the reference's run")
    return first, d64


if __name__ == "__main__":
    for tag in (sys.argv[1:] or ["L64", "L120", "T1124", "S1500"]):
        report(tag)
