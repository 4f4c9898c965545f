"""Synthetic random-backbone complexes for the benchmark configs C4 / C5.

Backbones are grown by NeRF with ideal geometry and a helix/sheet/coil (phi, psi) mixture,
kept compact by per-residue rejection; side chains are laid down from uniformly random chi
angles with the same rigid-group geometry the sampler uses.  Pure numpy; deterministic in
``seed``.  This is workload generation, not part of the sampling path.
"""
from typing import Dict, List

import numpy as np

from . import constants as rc

# residue-type frequencies of data/T1124_lig.pdb, restype order ARNDCQEGHILKMFPSTWYV
_T1124_COUNTS = np.array([106, 62, 19, 50, 8, 20, 47, 59, 16, 24, 87, 6, 12, 31, 28, 34, 44, 16, 19, 51], float)

_N_CA, _CA_C, _C_N, _C_O = 1.459, 1.525, 1.329, 1.229
_ANG_N_CA_C, _ANG_CA_C_N, _ANG_C_N_CA, _ANG_CA_C_O = np.deg2rad([111.0, 116.2, 121.7, 120.8])


def _place(a, b, c, length, angle, torsion):
    """Point d with |cd| = length, angle(b,c,d) = angle, dihedral(a,b,c,d) = torsion."""
    bc = c - b
    bc /= np.linalg.norm(bc)
    n = np.cross(b - a, bc)
    n /= np.linalg.norm(n)
    m = np.cross(n, bc)
    d2 = np.array([-length * np.cos(angle), length * np.sin(angle) * np.cos(torsion),
                   length * np.sin(angle) * np.sin(torsion)])
    return c + d2[0] * bc + d2[1] * m + d2[2] * n


def _phi_psi(rng):
    u = rng.random()
    if u < 0.4:
        return np.deg2rad(rng.normal((-60.0, -45.0), 10.0))
    if u < 0.7:
        return np.deg2rad(rng.normal((-120.0, 130.0), 15.0))
    return rng.uniform(-np.pi, np.pi, 2)


def _grow_chain(n_res: int, rng) -> np.ndarray:
    """[n_res, 4, 3] N, CA, C, O coordinates."""
    radius = 2.9 * n_res ** 0.38
    bb = np.zeros((n_res, 4, 3))
    bb[0, 0] = (0.0, 0.0, 0.0)
    bb[0, 1] = (_N_CA, 0.0, 0.0)
    bb[0, 2] = bb[0, 1] + _CA_C * np.array([-np.cos(_ANG_N_CA_C), np.sin(_ANG_N_CA_C), 0.0])
    prev_c = bb[0, 2] + np.array([0.3, -1.0, 0.5])        # virtual C(-1): fixes phi_0 frame only
    for i in range(n_res):
        n_i, ca_i = bb[i, 0], bb[i, 1]
        centroid = bb[:i + 1, 1].mean(0)
        for _ in range(20):
            phi, psi = _phi_psi(rng)
            c_i = _place(prev_c, n_i, ca_i, _CA_C, _ANG_N_CA_C, phi) if i > 0 else bb[0, 2]
            n_next = _place(n_i, ca_i, c_i, _C_N, _ANG_CA_C_N, psi)
            ca_next = _place(ca_i, c_i, n_next, _N_CA, _ANG_C_N_CA, np.pi)
            ok = np.linalg.norm(ca_next - centroid) <= radius
            if ok and i >= 1:
                ok = np.min(np.linalg.norm(bb[:i, 1] - ca_next, axis=1)) >= 4.0
            if ok:
                break
        bb[i, 2] = c_i
        bb[i, 3] = _place(n_i, ca_i, c_i, _C_O, _ANG_CA_C_O, psi + np.pi)
        if i + 1 < n_res:
            bb[i + 1, 0], bb[i + 1, 1] = n_next, ca_next
        prev_c = c_i
    return bb


def _rot_x(s, c):
    m = np.zeros(s.shape + (4, 4))
    m[..., 0, 0] = 1.0
    m[..., 1, 1] = c
    m[..., 1, 2] = -s
    m[..., 2, 1] = s
    m[..., 2, 2] = c
    m[..., 3, 3] = 1.0
    return m


def build_atom14(bb: np.ndarray, aatype: np.ndarray, chi: np.ndarray) -> np.ndarray:
    """All-atom [L,14,3] from backbone N/CA/C/O + chi angles (float64 rigid-group chain)."""
    L = bb.shape[0]
    n, ca, c = bb[:, 0], bb[:, 1], bb[:, 2]
    e0 = c - ca
    e0 /= np.linalg.norm(e0, axis=-1, keepdims=True)
    e1 = n - ca
    e1 -= e0 * (e0 * e1).sum(-1, keepdims=True)
    e1 /= np.linalg.norm(e1, axis=-1, keepdims=True)
    e2 = np.cross(e0, e1)
    glob = np.zeros((L, 4, 4))
    glob[:, :3, 0], glob[:, :3, 1], glob[:, :3, 2], glob[:, :3, 3] = e0, e1, e2, ca
    glob[:, 3, 3] = 1.0
    dflt = rc.default_frames[aatype].astype(np.float64)                         # [L,8,4,4]
    frames = np.zeros((L, 8, 4, 4))
    frames[:, 0] = glob
    to_bb = dflt[:, 4] @ _rot_x(np.sin(chi[:, 0]), np.cos(chi[:, 0]))
    frames[:, 4] = glob @ to_bb
    for k in (1, 2, 3):
        to_bb = to_bb @ dflt[:, 4 + k] @ _rot_x(np.sin(chi[:, k]), np.cos(chi[:, k]))
        frames[:, 4 + k] = glob @ to_bb
    for g in (1, 2, 3):
        frames[:, g] = glob @ dflt[:, g]
    grp = rc.atom14_to_group[aatype]                                             # [L,14]
    lit = np.concatenate([rc.lit_positions[aatype].astype(np.float64), np.ones((L, 14, 1))], -1)
    f = np.take_along_axis(frames, grp[:, :, None, None], axis=1)                # [L,14,4,4]
    xyz = np.einsum("laij,laj->lai", f, lit)[..., :3]
    xyz[:, :4] = bb
    return xyz


def make_complex(n_res: int, seed: int, n_chains: int = 2) -> Dict:
    """Protein dict (the layout ``pdb_io.from_pdb_file`` returns) of a compact random complex."""
    rng = np.random.default_rng(seed)
    sizes = [n_res // n_chains + (1 if k < n_res % n_chains else 0) for k in range(n_chains)]
    bbs, chains, ridx = [], [], []
    for k, sz in enumerate(sizes):
        bb = _grow_chain(sz, rng)
        bb -= bb[:, 1].mean(0)
        if k > 0:
            direction = rng.normal(size=3)
            direction /= np.linalg.norm(direction)
            bb += direction * 1.2 * 2.9 * sz ** 0.38 * k
        bbs.append(bb)
        chains += [chr(ord("A") + k)] * sz
        ridx += list(range(1, sz + 1))
    bb = np.concatenate(bbs)
    aatype = rng.choice(20, size=n_res, p=_T1124_COUNTS / _T1124_COUNTS.sum())
    chi = rng.uniform(-np.pi, np.pi, (n_res, 4)) * rc.chi_angles_mask[aatype]
    xyz = build_atom14(bb, aatype, chi)
    mask = rc.atom14_mask[aatype].astype(np.float64)
    xyz = np.where(mask[..., None] > 0, xyz, np.nan).astype(np.float32).astype(np.float64)
    return dict(atom_positions=xyz, atom_mask=mask, aaindex=aatype.astype(np.int64),
                residue_index=np.array(ridx, np.int64), chain_id=np.array(chains),
                b_factors=np.zeros((n_res, 14)))


def c5_lengths(n_complexes: int = 256) -> List[int]:
    """Residue counts of benchmark config C5: L_i ~ U{270..330}, default_rng(256)."""
    return [int(x) for x in np.random.default_rng(256).integers(270, 331, size=n_complexes)]


def _make_one(spec):
    return make_complex(*spec)


def make_complexes(specs, workers: int = 1) -> List[Dict]:
    """``[make_complex(n_res, seed) for n_res, seed in specs]``, on ``workers`` forked processes when that is more than one
    (0.25 s of numpy per ~300-residue complex: the 256 complexes of config C5 take a minute on one core).  Call it BEFORE the
    process touches the GPU: a forked child must not inherit an initialised HIP runtime."""
    specs = [tuple(s) for s in specs]
    if workers <= 1 or len(specs) < 4:
        return [make_complex(*s) for s in specs]
    import multiprocessing as mp
    with mp.get_context("fork").Pool(min(workers, len(specs))) as pool:
        return pool.map(_make_one, specs, chunksize=max(1, len(specs) // (4 * workers)))
