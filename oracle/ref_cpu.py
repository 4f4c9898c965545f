"""CPU oracle for the PackPPI-MSC sampling / proximal path.  TEST INFRASTRUCTURE ONLY.

A from-the-spec restatement (torch CPU tensors, fp32 by default, fp64 on request) of the
reference algorithm the HIP path must reproduce.  It is imported only by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` -- never by the
product path under ``packppi_amd/``.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against
golden vectors produced by running the unmodified reference in the build container
(``tools/oracle/make_golden.py``; fixtures in ``tests/golden/``).

Each function cites the reference lines (relative to the upstream repo root) it follows.
"""
import math
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from packppi_amd import constants as rc

PI = math.pi
TOP_K = 32
N_POINTS = 8
SIGMA_MIN, SIGMA_MAX = 0.01 * PI, PI           # schedule.py:149-150
ANNEALED_TEMP = 3.0                            # Sampling.yaml:4


# --------------------------------------------------------------------------------------
# small helpers
# --------------------------------------------------------------------------------------
def _linear(x, W, b):
    return F.linear(x, W, b)


def _ln(x, g, b):
    return F.layer_norm(x, (x.shape[-1],), g, b, 1e-5)


def _gather_nodes(nodes, idx):
    """nodes [B,L,C], idx [B,L,K] -> [B,L,K,C]   (components/__init__.py:16-30)."""
    B, L, K = idx.shape
    flat = idx.reshape(B, L * K, 1).expand(-1, -1, nodes.shape[-1])
    return torch.gather(nodes, 1, flat).reshape(B, L, K, nodes.shape[-1])


def _mlp(x, sd, prefix, n_inter):
    """Linear-ReLU stack (layers.py:10-33)."""
    x = F.relu(_linear(x, sd[prefix + ".W_in.weight"], sd[prefix + ".W_in.bias"]))
    for k in range(n_inter):
        x = F.relu(_linear(x, sd[f"{prefix}.W_inter.{k}.weight"], sd[f"{prefix}.W_inter.{k}.bias"]))
    return _linear(x, sd[prefix + ".W_out.weight"], sd[prefix + ".W_out.bias"])


# --------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------
def backbone_frames(X):
    """Gram-Schmidt residue frames from N, CA, C: R [.,3,3] (columns e0,e1,e2), t = CA.

    rigid_utils.py:1127-1179 with fixed=True as called from features.py:90-92
    (e0 along C-CA, e1 from N-CA, eps 1e-8 inside both square roots).
    """
    n, ca, c = X[..., 0, :], X[..., 1, :], X[..., 2, :]
    a = [c[..., k] - ca[..., k] for k in range(3)]
    b = [n[..., k] - ca[..., k] for k in range(3)]
    na = torch.sqrt(0 + a[0] * a[0] + a[1] * a[1] + a[2] * a[2] + 1e-8)
    a = [v / na for v in a]
    dot = 0 + a[0] * b[0] + a[1] * b[1] + a[2] * b[2]
    b = [bv - av * dot for av, bv in zip(a, b)]
    nb = torch.sqrt(0 + b[0] * b[0] + b[1] * b[1] + b[2] * b[2] + 1e-8)
    b = [v / nb for v in b]
    cx = [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]
    R = torch.stack([torch.stack([a[r], b[r], cx[r]], -1) for r in range(3)], -2)
    return R, ca


def _rot_apply(R, v):
    """R [...,3,3] applied to v [...,3] with the reference's left-to-right sums (rigid_utils.py:233-254)."""
    x, y, z = v[..., 0], v[..., 1], v[..., 2]
    return torch.stack([R[..., r, 0] * x + R[..., r, 1] * y + R[..., r, 2] * z for r in range(3)], -1)


def _rot_mul(A, Bm):
    """3x3 product written out as the reference does (rigid_utils.py:181-218)."""
    rows = []
    for i in range(3):
        rows.append(torch.stack([A[..., i, 0] * Bm[..., 0, j] + A[..., i, 1] * Bm[..., 1, j]
                                 + A[..., i, 2] * Bm[..., 2, j] for j in range(3)], -1))
    return torch.stack(rows, -2)


def _compose(Ra, ta, Rb, tb):
    """(Ra,ta) o (Rb,tb)   (rigid_utils.py:990-1003)."""
    return _rot_mul(Ra, Rb), _rot_apply(Ra, tb) + ta


def atom14_coords(X, S, BB_D, SC_D):
    """Torsion angles -> atom14 coordinates.

    components/__init__.py:76-120; features.py:95-158 (frames), :161-194 (atoms).
    Backbone slots 0..3 are overwritten with the input X, so slot 4 (CB) is the ideal one.
    """
    dt = X.dtype
    ang = torch.cat([torch.stack((BB_D.sin(), BB_D.cos()), -1), torch.stack((SC_D.sin(), SC_D.cos()), -1)], -2)
    ang = ang / torch.sqrt(torch.clamp((ang ** 2).sum(-1, keepdim=True), min=1e-12))       # [B,L,7,2]
    Rg, tg = backbone_frames(X)
    dflt = torch.as_tensor(rc.default_frames, dtype=dt)[S]                                 # [B,L,8,4,4]
    Rd, td = dflt[..., :3, :3], dflt[..., :3, 3]
    lead = torch.zeros(ang.shape[:-2] + (1, 2), dtype=dt)
    lead[..., 1] = 1
    a8 = torch.cat([lead, ang], -2)                                                        # [B,L,8,2]
    s, c = a8[..., 0], a8[..., 1]
    rot = torch.zeros(a8.shape[:-1] + (3, 3), dtype=dt)
    rot[..., 0, 0] = 1
    rot[..., 1, 1] = c
    rot[..., 1, 2] = -s
    rot[..., 2, 1] = s
    rot[..., 2, 2] = c
    # default o rot : pure rotation on the right, so translation is R_d * 0 + t_d
    Rf = _rot_mul(Rd, rot)
    tf = _rot_apply(Rd, torch.zeros_like(td)) + td
    Rs, ts = [Rf[..., g, :, :] for g in range(5)], [tf[..., g, :] for g in range(5)]
    Rc, tc = Rs[4], ts[4]
    for g in (5, 6, 7):
        Rc, tc = _compose(Rc, tc, Rf[..., g, :, :], tf[..., g, :])
        Rs.append(Rc)
        ts.append(tc)
    Rall, tall = torch.stack(Rs, -3), torch.stack(ts, -2)                                  # [B,L,8,..]
    Rall, tall = _compose(Rg[..., None, :, :], tg[..., None, :], Rall, tall)
    grp = torch.as_tensor(rc.atom14_to_group)[S]                                           # [B,L,14]
    onehot = F.one_hot(grp, 8).to(dt)                                                      # [B,L,14,8]
    Ra = (Rall[..., None, :, :, :] * onehot[..., None, None]).sum(-3)                      # [B,L,14,3,3]
    ta = (tall[..., None, :, :] * onehot[..., None]).sum(-2)                               # [B,L,14,3]
    lit = torch.as_tensor(rc.lit_positions, dtype=dt)[S]
    xyz = (_rot_apply(Ra, lit) + ta) * torch.as_tensor(rc.atom14_mask, dtype=dt)[S][..., None]
    xyz = xyz.clone()
    xyz[..., :4, :] = X[..., :4, :]
    return xyz


# --------------------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------------------
def knn_graph(X_ca, mask, top_k=TOP_K):
    """K nearest CA neighbours incl. self, masked pairs pushed to 2*rowmax (encoder.py:105-118)."""
    m2 = mask[:, None, :] * mask[:, :, None]
    dX = X_ca[:, None, :, :] - X_ca[:, :, None, :]
    D = m2 * torch.sqrt((dX ** 2).sum(3) + 1e-6)
    Dmax = D.max(-1, keepdim=True)[0]
    Dadj = D + 2 * (1.0 - m2) * Dmax
    _, E_idx = torch.topk(Dadj, min(top_k, X_ca.shape[-2]), dim=-1, largest=False)
    return E_idx


def _pair_dihedral(p0, p1, p2, p3):
    """sign * arccos of the normal dot product, NaN -> 0 (encoder.py:164-174)."""
    u0, u1, u2 = p2 - p1, p0 - p1, p3 - p2

    def unit(v):
        return torch.nan_to_num(v / torch.norm(v, dim=-1, keepdim=True))

    n1 = unit(torch.cross(u0, u1, dim=-1))
    n2 = unit(torch.cross(u0, u2, dim=-1))
    sgn = torch.sign((torch.cross(u1, u2, dim=-1) * u0).sum(-1))
    return torch.nan_to_num(sgn * torch.arccos((n1 * n2).sum(-1)))


def edge_features(X, E_idx, residue_index, chain_indices, zero_self_dihedral=False):
    """468 raw edge features per (i, j in kNN(i))   (encoder.py:34-47,120-153,164-196,232-236).

    ``zero_self_dihedral`` selects the build's convention for the j == i edge (exactly 0
    instead of the reference's arccos rounding noise; DESIGN.md "self-edge dihedrals").
    """
    B, L, K = E_idx.shape
    dt = X.dtype
    N, CA, C, O = X[:, :, 0], X[:, :, 1], X[:, :, 2], X[:, :, 3]
    b, c = CA - N, C - CA
    CB = -0.58273431 * torch.cross(b, c, dim=-1) + 0.56802827 * b - 0.54067466 * c + CA
    atoms = [N, CA, C, O, CB]

    def nbr(v):                                                                   # [B,L,3]->[B,L,K,3]
        return _gather_nodes(v, E_idx)

    off = residue_index[:, :, None] - torch.gather(residue_index[:, None, :].expand(-1, L, -1), 2, E_idx)
    relpos = F.one_hot(torch.clip(off + 32, 0, 64), 65).to(dt)

    mu = torch.linspace(0.0, 20.0, 16).to(dt).view(1, 1, 1, -1)
    sigma = (20.0 - 0.0) / 16
    rbfs = []
    nbrs = [nbr(a) for a in atoms]
    for a in atoms:
        for bj in nbrs:
            d = torch.sqrt(((a[:, :, None, :] - bj) ** 2).sum(-1) + 1e-6)
            rbfs.append(torch.exp(-((d[..., None] - mu) / sigma) ** 2))
    rbf = torch.cat(rbfs, -1)

    same = (chain_indices[:, :, None] == torch.gather(chain_indices[:, None, :].expand(-1, L, -1), 2, E_idx))
    etype = (same.to(dt) + 1)[..., None]

    Ci, Ni, CAi = (v[:, :, None, :].expand(-1, -1, K, -1) for v in (C, N, CA))
    Nj, CAj, Cj = nbrs[0], nbrs[1], nbrs[2]
    phi = _pair_dihedral(Ci, Nj, CAj, Cj)
    psi = _pair_dihedral(Ni, CAi, Ci, Nj)
    if zero_self_dihedral:
        is_self = E_idx == torch.arange(L).view(1, L, 1)
        phi = torch.where(is_self, torch.zeros_like(phi), phi)
        psi = torch.where(is_self, torch.zeros_like(psi), psi)
    return torch.cat([relpos, rbf, etype, torch.stack((phi, psi), -1)], -1)


def time_embedding(t):
    """16-d sinusoidal embedding of t*1e4 (layers.py:257-268; scale 10000 from encoder.py:87-91)."""
    ts = t.clone() * 10000
    freq = torch.exp(torch.arange(8, dtype=torch.float32) * -(math.log(10000) / 7)).to(t.dtype)
    arg = ts[:, None] * freq[None, :]
    return torch.cat([arg.sin(), arg.cos()], 1)


def encode_static(sd, batch, zero_self_dihedral=False):
    """Timestep-invariant part of ProteinEncoder.forward: E_idx and h_E0 (encoder.py:198-246)."""
    E_idx = knn_graph(batch["X"][:, :, 1, :], batch["residue_mask"])
    E = edge_features(batch["X"], E_idx, batch["residue_index"], batch["chain_indices"], zero_self_dihedral)
    h_E = _ln(_linear(E, sd["encoder.edge_embedding.weight"], sd["encoder.edge_embedding.bias"]),
              sd["encoder.norm_edges.weight"], sd["encoder.norm_edges.bias"])
    return E_idx, h_E


def encode_nodes(sd, batch, chi, t):
    """Node features 51 -> Linear -> LayerNorm (TorsionalDiffusion.py:91-92; encoder.py:218-242)."""
    B, L = batch["residue_type"].shape
    dt = batch["X"].dtype
    sc = torch.stack((chi.sin(), chi.cos()), -1) * batch["SC_D_mask"][..., None]
    V = torch.cat([F.one_hot(batch["residue_type"], 21).to(dt),
                   batch["BB_D_sincos"].reshape(B, L, 6), sc.reshape(B, L, 8),
                   time_embedding(t).reshape(B, L, 16)], -1)
    return _ln(_linear(V, sd["encoder.node_embedding.weight"], sd["encoder.node_embedding.bias"]),
               sd["encoder.norm_nodes.weight"], sd["encoder.norm_nodes.bias"])


# --------------------------------------------------------------------------------------
# message passing
# --------------------------------------------------------------------------------------
def _message_input(sd, pfx, which, h_V, h_E, E_idx, R, tr):
    """[h_V_i | h_E_ij | h_V_j | 72 invariant point features] = 456   (layers.py:65-117)."""
    B, L, K = E_idx.shape
    p_loc = _linear(h_V, sd[f"{pfx}points_fn_{which}.weight"], sd[f"{pfx}points_fn_{which}.bias"])
    p_loc = p_loc.reshape(B, L, N_POINTS, 3)
    p_glob = _rot_apply(R[:, :, None], p_loc) + tr[:, :, None]                   # position_scale == 1
    nbr_glob = _gather_nodes(p_glob.reshape(B, L, -1), E_idx).reshape(B, L, K, N_POINTS, 3)
    loc_e = p_loc[:, :, None].expand(-1, -1, K, -1, -1)
    loc_norm = torch.sqrt((loc_e ** 2).sum(-1) + 1e-8)
    Rt = R.transpose(-1, -2)[:, :, None, None]
    nbr_loc = _rot_apply(Rt, nbr_glob - tr[:, :, None, None])
    nbr_loc_norm = torch.sqrt((nbr_loc ** 2).sum(-1) + 1e-8)
    dist = torch.sqrt(((p_glob[:, :, None] - nbr_glob) ** 2).sum(-1) + 1e-8)
    return torch.cat([h_V[:, :, None].expand(-1, -1, K, -1), h_E, _gather_nodes(h_V, E_idx),
                      loc_e.reshape(B, L, K, -1), loc_norm, nbr_loc.reshape(B, L, K, -1), nbr_loc_norm, dist], -1)


def ipmp_layer(sd, l, h_V, h_E, E_idx, R, tr, mask_V, mask_att, edge_update=True):
    """InvariantPointMessagePassing.forward in eval mode (layers.py:119-148)."""
    p = f"mpnn.mpnn_layers.{l}."
    m = _mlp(_message_input(sd, p, "node", h_V, h_E, E_idx, R, tr), sd, p + "node_message_fn", 1)
    m = (m * mask_att[..., None]).mean(-2)                    # divides by K, not by #valid
    h_V = _ln(h_V + m, sd[p + "norm.0.weight"], sd[p + "norm.0.bias"])
    h_V = _ln(h_V + _mlp(h_V, sd, p + "node_dense", 0), sd[p + "norm.1.weight"], sd[p + "norm.1.bias"])
    h_V = h_V * mask_V[..., None]
    if edge_update:
        m = _mlp(_message_input(sd, p, "edge", h_V, h_E, E_idx, R, tr), sd, p + "edge_message_fn", 1)
        h_E = _ln(h_E + m * mask_att[..., None], sd[p + "norm.2.weight"], sd[p + "norm.2.bias"])
        h_E = _ln(h_E + _mlp(h_E, sd, p + "edge_dense", 0), sd[p + "norm.3.weight"], sd[p + "norm.3.bias"])
        h_E = h_E * mask_att[..., None]
    return h_V, h_E


def network(sd, batch, chi, t, static=None, zero_self_dihedral=False):
    """score [B,L,4] and h_V [B,L,128]   (TorsionalDiffusion.py:90-109; mpnn.py:47-62)."""
    if static is None:
        static = encode_static(sd, batch, zero_self_dihedral)
    E_idx, h_E = static
    h_V = encode_nodes(sd, batch, chi, t)
    R, tr = backbone_frames(batch["X"])
    mask = batch["residue_mask"]
    mask_att = mask[..., None] * _gather_nodes(mask[..., None], E_idx)[..., 0]
    for l in range(3):
        h_V, h_E = ipmp_layer(sd, l, h_V, h_E, E_idx, R, tr, mask, mask_att, edge_update=(l < 2))
    s = _mlp(h_V, sd, "decoder_score.0", 0)
    s = _mlp(F.relu(s), sd, "decoder_score.2", 0)
    return s, h_V


# --------------------------------------------------------------------------------------
# SO(2) variance-exploding schedule
# --------------------------------------------------------------------------------------
def t_to_sigma(t):
    """schedule.py:165-174."""
    lo, hi = np.log(SIGMA_MIN), np.log(SIGMA_MAX)
    return torch.exp(lo + (hi - lo) * t)


def wrap_pi(x):
    """(x + pi) % 2pi - pi   (TorsionalDiffusion.py:121,278)."""
    return (x + np.pi) % (2 * np.pi) - np.pi


def initial_noise(batch, generator=None):
    """The two N(0,1) draws ``add_sc_noise`` makes at t = 1, in its order (schedule.py:186)."""
    shape = (batch["SC_D"].numel() // 4, 4)
    return (torch.randn(shape, generator=generator, dtype=torch.float32),
            torch.randn(shape, generator=generator, dtype=torch.float32))


def add_sc_noise(batch, t, noise_pair):
    """chi_true + sigma(t) * noise on the 1pi then 2pi masks, wrapped (TorsionalDiffusion.py:111-124)."""
    x = batch["SC_D"].reshape(-1, 4)
    sig = t_to_sigma(t)[:, None]
    x = x + noise_pair[0].to(x.dtype) * sig * batch["chi_1pi_periodic_mask"].reshape(-1, 4)
    x = x + noise_pair[1].to(x.dtype) * sig * batch["chi_2pi_periodic_mask"].reshape(-1, 4)
    return wrap_pi(x).reshape(batch["SC_D"].shape)


def reverse_step(x, score, time, dt, mask, mode="ode", noise=None):
    """One SO2VESchedule.step on ``mask`` (schedule.py:198-235)."""
    sigma = t_to_sigma(time)
    g = sigma * np.sqrt(2 * np.log(SIGMA_MAX / SIGMA_MIN))
    alpha = 1 - (sigma / np.exp(np.log(SIGMA_MAX))) ** 2
    w = ANNEALED_TEMP / (alpha + (1 - alpha) * ANNEALED_TEMP) if ANNEALED_TEMP else 1      # schedule.py:216-217
    if mode == "ode":
        new = x + 0.5 * g ** 2 * dt * (score * w)
    else:
        new = x + (g ** 2 * dt * (score * w) + g * torch.sqrt(dt) * noise)
    return torch.where(mask, new, x)


def sampling(sd, batch, init_chi, schedule=None, mode="ode", sde_noise=None, hoist=True,
             zero_self_dihedral=False, trajectory=False):
    """Reverse diffusion from ``init_chi`` (TorsionalDiffusion.py:254-280).

    ``hoist=True`` evaluates the timestep-invariant graph/edge embedding once (identical
    values, it is a pure function of the backbone); ``hoist=False`` recomputes it every
    step the way the reference does, which is what the CPU baseline times.
    """
    if schedule is None:
        schedule = torch.linspace(1, 0, 31)
    B, L = batch["residue_type"].shape
    x = init_chi.clone()
    static = encode_static(sd, batch, zero_self_dihedral) if hoist else None
    m1 = batch["chi_1pi_periodic_mask"].reshape(-1, 4)
    m2 = batch["chi_2pi_periodic_mask"].reshape(-1, 4)
    traj = []
    for j in range(len(schedule) - 1):
        time = schedule[j].to(x.dtype)
        dt = (schedule[j] - schedule[j + 1]).to(x.dtype)
        t = time.repeat_interleave(B * L)
        score, _ = network(sd, batch, x, t, static, zero_self_dihedral)
        score = score.reshape(-1, 4)
        x = x.reshape(-1, 4)
        n1 = n2 = None
        if mode == "sde":
            n1, n2 = sde_noise[j]
        x = reverse_step(x, score, time, dt, m1, mode, n1)
        x = reverse_step(x, score, time, dt, m2, mode, n2)
        x = wrap_pi(x).reshape(B, L, 4) * batch["SC_D_mask"]
        if trajectory:
            traj.append(x.clone())
    return (x, traj) if trajectory else x


# --------------------------------------------------------------------------------------
# clash loss and proximal optimisation
# --------------------------------------------------------------------------------------
def between_residue_clash(xyz, exists, radius, residue_index, tol):
    """Per-atom inter-residue hinge sums [B,L,14]   (clash.py:102-254)."""
    dt = xyz.dtype
    d = torch.sqrt(1e-10 + ((xyz[..., :, None, :, None, :] - xyz[..., None, :, None, :, :]) ** 2).sum(-1))
    m = exists[..., :, None, :, None] * exists[..., None, :, None, :]
    bb = torch.zeros(14, 14, dtype=dt)
    bb[:4, :4] = 1
    m = m * (1 - bb)
    m = m * (residue_index[..., :, None, None, None] < residue_index[..., None, :, None, None])
    cn = torch.zeros(14, 14, dtype=dt)
    cn[2, 0] = 1                                               # C(i) - N(i+1) peptide bond
    nb = ((residue_index[..., :, None] + 1) == residue_index[..., None, :])[..., None, None]
    m = m * (1 - nb * cn)
    ss = torch.zeros(14, 14, dtype=dt)
    ss[5, 5] = 1                                               # "SG" slot pair, every residue type
    m = m * (1 - ss)
    lower = m * (radius[..., :, None, :, None] + radius[..., None, :, None, :])
    err = m * F.relu(lower - tol - d)
    return err.sum(dim=(-4, -2)) + err.sum(dim=(-3, -1))


def within_residue_violation(xyz, exists, lower, upper):
    """Per-atom intra-residue bound violations [B,L,14]   (clash.py:7-99)."""
    dt = xyz.dtype
    m = exists[..., :, None] * exists[..., None, :] * (1 - torch.eye(14, dtype=dt))
    bb = torch.zeros(14, 14, dtype=dt)
    bb[:4, :4] = 1
    m = m * (1 - bb)
    d = torch.sqrt(1e-10 + ((xyz[..., :, None, :] - xyz[..., None, :, :]) ** 2).sum(-1))
    loss = m * (F.relu(lower + 0.0 - d) + F.relu(d - (upper - 0.0)))
    return loss.sum(-2) + loss.sum(-1)


def residue_clash(batch, chi, vtf=12.0, tol=0.5):
    """Per-residue side-chain clash score [B,L]   (clash.py:335-365, :257-332)."""
    dt = batch["X"].dtype
    S = batch["residue_type"]
    exists = batch["atom_mask"]
    n_sc = exists[..., 4:].sum(-1)
    xyz = atom14_coords(batch["X"], S, batch["BB_D"], chi)
    radius = exists * torch.as_tensor(rc.between_radius, dtype=dt)[S]
    lo, up = rc.make_atom14_dists_bounds(overlap_tolerance=tol, bond_length_tolerance_factor=vtf)
    per_atom = (between_residue_clash(xyz, exists, radius, batch["residue_index"], tol)
                + within_residue_violation(xyz, exists, torch.as_tensor(lo, dtype=dt)[S],
                                           torch.as_tensor(up, dtype=dt)[S]))
    return per_atom[..., 4:].sum(-1) / (1e-10 + n_sc)


def clash_mask(batch, chi, vtf, tol):
    """Residues whose clash score exceeds the mean, broadcast to their 4 chi (optimize.py:5-18)."""
    pr = residue_clash(batch, chi, vtf, tol)
    return (pr > pr.mean())[..., None].expand(-1, -1, 4)


def proximal_loss(batch, x, chi0, mask, z, vtf, tol, lamda):
    """optimize.py:33-45."""
    x = x * mask
    x = torch.where(mask, x, chi0)
    pr = residue_clash(batch, x, vtf, tol)
    return (torch.abs(x - z) ** 2).sum(-1).mean() + lamda * pr.mean()


def proximal_optimizer(batch, chi0, vtf=12.0, tol=0.5, lamda=1.0, num_steps=50):
    """Adam(lr 1e-2) on the clash-masked chi; returns (per-step chi list, pre-step losses) (optimize.py:21-73)."""
    assert int(batch["num_proteins"]) == 1
    with torch.no_grad():
        mask = clash_mask(batch, chi0, vtf, tol)
    z = chi0 * mask
    x = z.clone().requires_grad_(True)
    opt = torch.optim.Adam([x], lr=1e-2)
    chis, losses = [], []
    for _ in range(num_steps):
        opt.zero_grad()
        loss = proximal_loss(batch, x, chi0, mask, z, vtf, tol, lamda)
        loss.backward()
        opt.step()
        chis.append(torch.where(mask, x.detach().clone(), chi0))
        losses.append(loss.item())
    return chis, losses


def clash_and_grad(batch, chi, vtf=12.0, tol=0.5):
    """per-residue clash [B,L] and d(mean over residues)/dchi [B,L,4] by autograd (checker for K14)."""
    x = chi.clone().requires_grad_(True)
    pr = residue_clash(batch, x, vtf, tol)
    pr.mean().backward()
    return pr.detach(), x.grad.detach()


# --------------------------------------------------------------------------------------
# metrics
# --------------------------------------------------------------------------------------
def atom_msd(true_xyz, pred_xyz, atom_mask, residue_mask, eps=1e-6):
    """Mean *squared* deviation, no square root (TorsionalDiffusion.py:300-309)."""
    w = atom_mask * residue_mask[..., None]
    return (((true_xyz - pred_xyz) ** 2).sum(-1) * w).sum() / (w + eps).sum()


def analyze_samples(batch, chi):
    """Per-chi MAE (rad, deg), acc@20deg and atom_rmsd   (TorsionalDiffusion.py:311-341)."""
    out = {}
    true, m, pi1 = batch["SC_D"], batch["SC_D_mask"], batch["chi_1pi_periodic_mask"]
    for i in range(4):
        n = m[..., i].sum()
        n = n if n != 0 else 1
        diff = (chi[..., i] - true[..., i]).abs()
        acc = torch.logical_and(diff * 180 / np.pi < 20, diff > 0).to(diff.dtype)
        ae = torch.minimum(diff, 2 * np.pi - diff)
        ae = torch.where(pi1[..., i], torch.minimum(ae, np.pi - ae), ae)
        out[f"chi_{i}_ae_rad"] = ae.sum() / n
        out[f"chi_{i}_ae_deg"] = (ae * 180 / np.pi).sum() / n
        out[f"chi_{i}_acc"] = acc.sum() / n
    pred = atom14_coords(batch["X"], batch["residue_type"], batch["BB_D"], chi)
    out["atom_rmsd"] = atom_msd(batch["X"], pred, batch["atom_mask"], batch["residue_mask"])
    return out
