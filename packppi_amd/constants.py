"""Residue-chemistry data tables for the sampling / proximal path.

The numbers live in ``data/residue_constants.npz`` (AlphaFold2 rigid-group geometry and
Engh-Huber bond statistics, dumped from the tables the reference builds at
``src/utils/residue_constants.py:29-240,280-285,459-554,595-677,709-806`` by
``tools/oracle/make_constants.py``).  This module only loads them and derives the
parameterised distance bounds the clash loss needs.
"""
import os
from functools import lru_cache

import numpy as np

_DATA = os.path.join(os.path.dirname(__file__), "data", "residue_constants.npz")
_z = np.load(_DATA)

restypes = [str(x) for x in _z["restypes"]]                     # 20 one-letter codes
resnames = [str(x) for x in _z["resnames"]]                     # 20 three-letter + "UNK"
restype_order = {r: i for i, r in enumerate(restypes)}
resname_to_idx = {r: i for i, r in enumerate(resnames)}
atom14_names = [[str(a) for a in row] for row in _z["atom14_names"]]   # [21][14], "" = empty
sidechain_atoms = set(str(a) for a in _z["sidechain_atoms"])

default_frames = _z["default_frames"]            # [21,8,4,4] f32   rigid-group default frames
atom14_to_group = _z["atom14_to_group"]          # [21,14]    i64   rigid group of each atom14 slot
atom14_mask = _z["atom14_mask"]                  # [21,14]    f32   slot exists for the residue type
lit_positions = _z["lit_positions"]              # [21,14,3]  f32   literature position in its group
chi_angles_mask = _z["chi_angles_mask"]          # [21,4]     f32
chi_pi_periodic = _z["chi_pi_periodic"]          # [21,4]     f32
chi_atom_indices_atom14 = _z["chi_atom_indices_atom14"]   # [21,7] i64
chi_mask_atom14 = _z["chi_mask_atom14"]          # [21,4]     f32
bond_rows = _z["bond_rows"]                      # [n,5] f64  (restype, a, b, length, stddev)
slot_radius = _z["slot_radius"]                  # [21,14] f64 element vdW radius, 0 for empty slot
between_radius = _z["between_radius"]            # [21,14] f64 radius table of clash.py:263-287


@lru_cache(maxsize=16)
def make_atom14_dists_bounds(overlap_tolerance: float = 1.5,
                             bond_length_tolerance_factor: float = 15.0):
    """Per-residue-type lower/upper bounds on intra-residue atom distances.

    Same result as the reference's ``make_atom14_dists_bounds``
    (residue_constants.py:809-869): every existing atom pair gets
    ``lower = r_a + r_b - overlap_tolerance`` and ``upper = 1e10``; bonded and
    angle-related ("virtual bond") pairs are then overwritten with
    ``length -/+ factor * stddev``.  Arithmetic in float64, stored float32.
    """
    lower = np.zeros((21, 14, 14), np.float32)
    upper = np.zeros((21, 14, 14), np.float32)
    exists = slot_radius > 0
    for rt in range(20):
        ex = exists[rt]
        pair = ex[:, None] & ex[None, :] & ~np.eye(14, dtype=bool)
        lo = slot_radius[rt][:, None] + slot_radius[rt][None, :] - overlap_tolerance
        lower[rt][pair] = lo[pair]
        upper[rt][pair] = 1e10
    for rt, a, b, length, sd in bond_rows:
        rt, a, b = int(rt), int(a), int(b)
        lo = length - bond_length_tolerance_factor * sd
        up = length + bond_length_tolerance_factor * sd
        lower[rt, a, b] = lower[rt, b, a] = lo
        upper[rt, a, b] = upper[rt, b, a] = up
    return lower, upper
