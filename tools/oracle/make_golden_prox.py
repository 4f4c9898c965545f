"""Container-only: proximal-stage goldens at the benchmark sizes + an fp64 arbiter (tests/golden/g6_*.npz).

    PYTHONDONTWRITEBYTECODE=1 python tools/oracle/make_golden_prox.py [--only L64,L120,T1124,S1500] [--no64 S1500]

Runs the UNMODIFIED reference ``proximal_optimizer`` (optimize.py:21-73) through tools/oracle/refshim.py

  * in fp32 (what ``sampling(use_proximal=True)`` does, TorsionalDiffusion.py:286-298): 50 Adam steps from the
    reference's own 100-step sampling result (the ``chi_ode_100`` of g4_T1124 / g5_S1500; ``init_chi_seed11`` for the
    small synthetic cases), storing the angles after steps 1, 5, 10, 20, 50, the 50 pre-step losses, the accepted angles
    and the metrics of ``analyze_samples`` on them;
  * in fp64 (batch tensors and angles cast to double; same code) -- the arbiter: a test may ask
    |HIP - ref64| <= c * |ref32 - ref64| instead of a bound on |HIP - ref32| that fp32 round-off itself cannot hold.

Only data is written (inputs are already in the g3p/g4/g5 fixtures).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refshim  # noqa: E402
from packppi_amd.batch import Batch, TENSOR_KEYS  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
KEEP = (1, 5, 10, 20, 50)


def load_fixture(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    b = Batch({k: torch.from_numpy(z["batch." + k]) for k in TENSOR_KEYS})
    b["num_proteins"] = int(z["batch.num_proteins"])
    b["max_size"] = int(z["batch.max_size"])
    return z, b


def ref_batch(b, double=False):
    d = {}
    for k, v in b.items():
        if isinstance(v, torch.Tensor):
            v = v.clone()
            if double and v.dtype == torch.float32:
                v = v.double()
        d[k] = v
    return refshim.Data(**d)


def wrapped(a, b):
    d = (a.double() - b.double()).abs()
    return torch.minimum(d, (2 * np.pi - d).abs())


def chunked_fp64_proximal(b, chi0, vtf=12., tol=0.5, lamda=1., num_steps=50, rows=96):
    """fp64 ARBITER for complexes whose (L, L, 14, 14) pair tensors do not fit host memory in double precision (S1500: 35 GB
    each, ten of them in the reference): optimize.py:21-73 with the clash loss of clash.py:102-254 evaluated over row blocks of
    the residue-pair matrix and the gradient accumulated block by block (the loss is a sum over atom pairs, so this is the same
    function).  Built from the pinned oracle's pieces; validated against the REFERENCE's own fp64 run on T1124 by
    append_trajectory before it is used."""
    from oracle import ref_cpu as O
    from packppi_amd import constants as rc
    dt = torch.float64
    bd = {k: (v.double() if isinstance(v, torch.Tensor) and v.dtype == torch.float32 else v) for k, v in b.items()}
    S, exists, ridx = bd["residue_type"], bd["atom_mask"], bd["residue_index"]
    L = S.shape[1]
    n_sc = exists[..., 4:].sum(-1)
    radius = exists * torch.as_tensor(rc.between_radius, dtype=dt)[S]
    lo, up = rc.make_atom14_dists_bounds(overlap_tolerance=tol, bond_length_tolerance_factor=vtf)
    lo, up = torch.as_tensor(lo, dtype=dt)[S], torch.as_tensor(up, dtype=dt)[S]
    w_atom = torch.zeros(1, L, 14, dtype=dt)
    w_atom[..., 4:] = (1.0 / (1e-10 + n_sc))[..., None]              # per-residue score = sum_{a >= 4} per_atom / n_sc
    bb = torch.zeros(14, 14, dtype=dt); bb[:4, :4] = 1
    cn = torch.zeros(14, 14, dtype=dt); cn[2, 0] = 1
    ss = torch.zeros(14, 14, dtype=dt); ss[5, 5] = 1

    def per_res(chi, need_grad):
        """per-residue clash [1, L] and, if asked, d(sum_i c_i per_res_i)/dchi for given c (closure)."""
        x = chi.clone().requires_grad_(need_grad)
        xyz = O.atom14_coords(bd["X"], S, bd["BB_D"], x)
        return x, xyz

    def between_blocks(xyz, coef):
        """sum over atom pairs of err * (coef[i, a] + coef[j, b]) block by block; returns (per_atom [1,L,14] detached,
        d(that sum)/dxyz)."""
        per_atom = torch.zeros(1, L, 14, dtype=dt)
        gxyz = torch.zeros_like(xyz)
        xd = xyz.detach()
        for i0 in range(0, L, rows):
            i1 = min(L, i0 + rows)
            xi = xd[:, i0:i1].clone().requires_grad_(True)
            xj = xd.clone().requires_grad_(True)
            d = torch.sqrt(1e-10 + ((xi[:, :, None, :, None, :] - xj[:, None, :, None, :, :]) ** 2).sum(-1))
            m = exists[:, i0:i1, None, :, None] * exists[:, None, :, None, :]
            m = m * (1 - bb)
            m = m * (ridx[:, i0:i1, None, None, None] < ridx[:, None, :, None, None])
            nb = ((ridx[:, i0:i1, None] + 1) == ridx[:, None, :])[..., None, None]
            m = m * (1 - nb * cn) * (1 - ss)
            lower = m * (radius[:, i0:i1, None, :, None] + radius[:, None, :, None, :])
            err = m * torch.relu(lower - tol - d)
            per_atom[:, i0:i1] += err.detach().sum(dim=(2, 4))
            per_atom += err.detach().sum(dim=(1, 3))
            if coef is not None:
                tot = (err * (coef[:, i0:i1, None, :, None] + coef[:, None, :, None, :])).sum()
                gi, gj = torch.autograd.grad(tot, (xi, xj))
                gxyz[:, i0:i1] += gi
                gxyz += gj
        return per_atom, gxyz

    def clash(chi, need_grad):
        x, xyz = per_res(chi, need_grad)
        coef = (w_atom / L) if need_grad else None                    # loss term: lamda * mean_i per_res_i
        pa_b, gxyz = between_blocks(xyz, coef)
        within = O.within_residue_violation(xyz, exists, lo, up)
        pr = ((pa_b + within.detach())[..., 4:].sum(-1)) / (1e-10 + n_sc)
        if not need_grad:
            return pr, None
        tail = (within * w_atom / L).sum()
        xyz.backward(gxyz, retain_graph=True)
        tail.backward()
        return pr, x.grad

    chi0 = chi0.double()
    pr0, _ = clash(chi0, False)
    mask = (pr0 > pr0.mean())[..., None].expand(-1, -1, 4)
    z = chi0 * mask
    x = z.clone().requires_grad_(True)
    opt = torch.optim.Adam([x], lr=1e-2)
    chis, losses = [], []
    for it in range(num_steps):
        opt.zero_grad()
        xe = torch.where(mask, x * mask, chi0)
        pr, g = clash(xe.detach(), True)
        prox = (torch.abs(xe - z) ** 2).sum(-1).mean()
        prox.backward()
        x.grad += lamda * torch.where(mask, g, torch.zeros_like(g))
        losses.append(float(prox.detach() + lamda * pr.mean()))
        opt.step()
        chis.append(torch.where(mask, x.detach().clone(), chi0))
    return chis, losses


def append_trajectory(args, cases, proximal_optimizer):
    for tag in args.only.split(","):
        path = os.path.join(GOLD, f"g6_prox_{tag}.npz")
        old = dict(np.load(path))
        fx, key = cases[tag]
        z, b = load_fixture(fx)
        chi0 = torch.from_numpy(z[key]).float()
        t0 = time.time()
        chis32, losses32 = proximal_optimizer(ref_batch(b), chi0.clone(), 12., 0.5, 1., 50)
        chis32 = [c.detach() for c in chis32]
        print(f"  {tag} fp32 reference: {time.time() - t0:.0f}s", flush=True)
        for n in KEEP:       # the same run as the one stored (same thread count -> same reduction order)
            assert np.array_equal(chis32[n - 1].numpy(), old[f"chi32_step{n}"]), (tag, n)
        t0 = time.time()
        if tag in args.no64.split(","):
            # validate the chunked arbiter where the reference's fp64 run exists, then use it here.  The pinned oracle in
            # fp64 is not bit-for-bit the reference in fp64 (its loss differs by 1.5e-8 relative at the same point), and 50 Adam
            # steps amplify that: on L64 2.6e-8 rad at step 5, 2.8e-7 at step 10, 3e-4 at step 50 -- an order of magnitude
            # below the |ref32 - ref64| distances the arbiter is used to judge.
            zt, bt = load_fixture(cases["L120"][0])
            ct, _ = chunked_fp64_proximal(bt, torch.from_numpy(zt[cases["L120"][1]]).float(), rows=32)
            ref64 = np.load(os.path.join(GOLD, "g6_prox_L120.npz"))
            dv = {n: float(wrapped(ct[n - 1], torch.from_numpy(ref64[f"chi64_step{n}"])).max()) for n in KEEP}
            print(f"  chunked fp64 arbiter vs the reference's fp64 run on L120: {dv}", flush=True)
            assert dv[10] < 2e-6 and dv[50] < 1e-3, dv
            old["arbiter64_vs_reference_on_L120"] = np.array([dv[n] for n in KEEP])
            chis64, losses64 = chunked_fp64_proximal(b, chi0)
            old["arbiter64"] = np.array("oracle fp64, pair matrix in row blocks (tools/oracle/make_golden_prox.py)")
            for n in KEEP:
                old[f"chi64_step{n}"] = chis64[n - 1].numpy()
            old["losses64"] = np.array(losses64, np.float64)
        else:
            chis64, losses64 = proximal_optimizer(ref_batch(b, True), chi0.double(), 12., 0.5, 1., 50)
            chis64 = [c.detach() for c in chis64]
            for n in KEEP:
                assert np.array_equal(chis64[n - 1].numpy(), old[f"chi64_step{n}"]), (tag, n)
            old["arbiter64"] = np.array("reference fp64")
        print(f"  {tag} fp64: {time.time() - t0:.0f}s", flush=True)
        moved = torch.zeros(chi0.shape[1], dtype=torch.bool)
        for c in chis32:
            moved |= (c != chi0).any(-1)[0]
        idx = torch.nonzero(moved).flatten()
        old["traj32_residues"] = idx.to(torch.int32).numpy()
        old["traj32"] = torch.stack([c[0, idx] for c in chis32]).numpy()                 # [50, n_moved, 4]
        div = np.array([float(wrapped(a, c).max()) for a, c in zip(chis32, chis64)])
        old["div_32_64"] = div
        flips = np.nonzero(div > 1e-5)[0]
        old["first_flip_ref"] = np.int64(flips[0] + 1 if len(flips) else 51)              # 1-based step; 51 = never
        print(f"  {tag}: {len(idx)} residues move; |ref32 - ref64| by step: "
              + " ".join(f"{d:.1e}" for d in div[[0, 4, 9, 14, 19, 29, 39, 49]]) + f"; first > 1e-5 at step {int(old['first_flip_ref'])}",
              flush=True)
        np.savez_compressed(path, **old)
        print(f"  rewrote {os.path.basename(path)} {os.path.getsize(path) / 1e6:.2f} MB", flush=True)


def append_alt_hinge(args, cases, proximal_optimizer):
    """A hinge of the clash loss whose overlap r_a + r_b - tol - d is within fp32 rounding of zero is on in one fp32
    implementation and off in another, and the two runs then end O(lr) apart (L120: the pair (res 23 atom 6)-(res 93 atom 11)
    at iterate 18, +1.6e-6 A here).  The OTHER branch as the reference itself computes it: its fp32 run with the overlap
    tolerance moved by a few 1e-6 A (tol + delta: every hinge threshold shifts by delta, far below anything but such a marginal
    pair).  Stored per delta: every step on the residues of traj32 (alt32_traj.<k>), the step at which the run leaves the
    tol = 0.5 run (alt32_first_jump.<k>) and the end state (alt32_step50.<k>).  A test may require the end state of a run whose
    hinge flipped to be as close to ONE of the reference's end states as fp32 allows, instead of to a recording of itself."""
    deltas = [float(x) for x in args.deltas.split(",")]
    for tag in args.only.split(","):
        path = os.path.join(GOLD, f"g6_prox_{tag}.npz")
        old = dict(np.load(path))
        fx, key = cases[tag]
        z, b = load_fixture(fx)
        chi0 = torch.from_numpy(z[key]).float()
        idx = torch.from_numpy(old["traj32_residues"].astype(np.int64))
        base = torch.from_numpy(old["traj32"])
        old["alt32_deltas"] = np.array(deltas, np.float64)
        for k, dl in enumerate(deltas):
            t0 = time.time()
            chis, losses = proximal_optimizer(ref_batch(b), chi0.clone(), 12., 0.5 + dl, 1., 50)
            chis = [c.detach() for c in chis]
            d = np.array([float(wrapped(c[0, idx], base[n]).max()) for n, c in enumerate(chis)])
            jumps = [n + 1 for n in range(1, 50) if d[n] > 20 * max(d[n - 1], 2e-6)]
            moved_elsewhere = max(float((c[0] != chi0[0]).any(-1)[~torch.isin(torch.arange(chi0.shape[1]), idx)].sum()) for c in chis)
            old[f"alt32_traj.{k}"] = torch.stack([c[0, idx] for c in chis]).numpy()
            old[f"alt32_step50.{k}"] = chis[-1].numpy()
            old[f"alt32_first_jump.{k}"] = np.int64(jumps[0] if jumps else 51)
            old[f"alt32_losses.{k}"] = np.array(losses, np.float64)
            print(f"  {tag} tol 0.5 {dl:+.1e}: {time.time() - t0:.0f}s, leaves the tol = 0.5 run at step {jumps[0] if jumps else None} "
                  f"(distance by step: " + " ".join(f"{x:.1e}" for x in d[[0, 9, 17, 18, 19, 29, 49]]) + f"), residues outside traj32 that move: {moved_elsewhere:.0f}",
                  flush=True)
        np.savez_compressed(path, **old)
        print(f"  rewrote {os.path.basename(path)} {os.path.getsize(path) / 1e6:.2f} MB", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="L64,L120,T1124,S1500")
    ap.add_argument("--append-alt-hinge", action="store_true",
                    help="add the reference's fp32 runs with the overlap tolerance moved by --deltas (the other branch of a marginal hinge)")
    ap.add_argument("--deltas", default="1e-5,2e-5,-1e-5,5e-6")
    ap.add_argument("--no64", default="S1500", help="cases without the fp64 run (host memory)")
    ap.add_argument("--threads", type=int, default=6)
    ap.add_argument("--append-grads", action="store_true",
                    help="load the existing g6 files and add the reference's clash value and autograd gradient at its own fp32 "
                         "iterates (per_res32_stepN, grad32_stepN): the trajectory-independent check of the analytic gradient")
    ap.add_argument("--append-trajectory", action="store_true",
                    help="re-run the reference (fp32: every case; fp64: every case but --no64, which get the chunked fp64 "
                         "arbiter below) and add every step of the fp32 run (traj32 on the residues that move), the per-step "
                         "distance between the two precisions (div_32_64) and the first step at which they part (first_flip_ref)")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    from src.models.components.optimize import proximal_optimizer
    model = refshim.build_reference_module(0)          # only analyze_samples is used: weights do not matter

    cases = {"L64": ("g3_proximal_L64", "init_chi_seed11"), "L120": ("g3_proximal_L120", "init_chi_seed11"),
             "T1124": ("g4_T1124", "chi_ode_100"), "S1500": ("g5_S1500", "chi_ode_100")}
    if args.append_trajectory:
        append_trajectory(args, cases, proximal_optimizer)
        return
    if args.append_alt_hinge:
        append_alt_hinge(args, cases, proximal_optimizer)
        return
    if args.append_grads:
        from src.models.components.clash import compute_residue_clash
        for tag in args.only.split(","):
            path = os.path.join(GOLD, f"g6_prox_{tag}.npz")
            old = dict(np.load(path))
            _, b = load_fixture(cases[tag][0])
            rb = ref_batch(b)
            for n in KEEP:
                t0 = time.time()
                x = torch.from_numpy(old[f"chi32_step{n}"]).float().requires_grad_(True)
                pr = compute_residue_clash(rb, x, 12., 0.5)
                pr.mean().backward()
                old[f"per_res32_step{n}"] = pr.detach().numpy()
                old[f"grad32_step{n}"] = x.grad.numpy()
                print(f"  {tag} step {n}: clash mean {float(pr.mean()):.6f}  {time.time() - t0:.1f}s", flush=True)
            np.savez_compressed(path, **old)
            print(f"  rewrote {os.path.basename(path)} {os.path.getsize(path) / 1e6:.2f} MB", flush=True)
        return
    for tag in args.only.split(","):
        fx, key = cases[tag]
        z, b = load_fixture(fx)
        chi0 = torch.from_numpy(z[key]).float()
        out = {"source_fixture": np.array(fx), "chi0_key": np.array(key)}
        for prec in ("32", "64"):
            if prec == "64" and tag in args.no64.split(","):
                continue
            dbl = prec == "64"
            rb = ref_batch(b, dbl)
            x0 = chi0.double() if dbl else chi0.clone()
            t0 = time.time()
            chis, losses = proximal_optimizer(rb, x0, 12., 0.5, 1., 50)
            print(f"  {tag} fp{prec}: 50 steps {time.time() - t0:.1f}s  loss {losses[0]:.6f} -> {losses[-1]:.6f}", flush=True)
            for n in KEEP:
                out[f"chi{prec}_step{n}"] = chis[n - 1].detach().numpy()
            out[f"losses{prec}"] = np.array(losses, np.float64)
            accepted = chis[-1].detach() if losses[-1] < losses[0] else x0
            out[f"accepted{prec}"] = accepted.numpy()
            with torch.no_grad():
                for k, v in model.analyze_samples(rb, SC_D_sample=accepted).items():
                    out[f"metric{prec}.{k}"] = np.float64(v)
                if not dbl:
                    for k, v in model.analyze_samples(rb, SC_D_sample=x0).items():
                        out[f"metric32_before.{k}"] = np.float64(v)
        path = os.path.join(GOLD, f"g6_prox_{tag}.npz")
        np.savez_compressed(path, **out)
        print(f"  wrote {os.path.basename(path)} {os.path.getsize(path) / 1e6:.2f} MB", flush=True)


if __name__ == "__main__":
    main()
