"""Protein dict -> batch tensors (host side, torch CPU).

Mirrors what ``ComplexDataset.prot_to_data`` (complex_dataset.py:65-148) and the dihedral
helpers (helper.py:20-101) compute, so that a batch built here is interchangeable with
the reference's.  SURVEY.md §8(f) row 1.
"""
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import constants as rc
from .batch import Batch, as_single


def _unit(v: torch.Tensor) -> torch.Tensor:
    return torch.nan_to_num(v / torch.norm(v, dim=-1, keepdim=True))


def dihedrals_along(points: torch.Tensor) -> torch.Tensor:
    """Signed dihedral of every 4 consecutive points along axis -2 (helper.py:20-36)."""
    bond = _unit(points[..., 1:, :] - points[..., :-1, :])
    b_prev, b_mid, b_next = bond[..., :-2, :], bond[..., 1:-1, :], bond[..., 2:, :]
    n_a = _unit(torch.cross(b_prev, b_mid, dim=-1))
    n_b = _unit(torch.cross(b_mid, b_next, dim=-1))
    cosang = torch.clamp((n_a * n_b).sum(-1), -1 + 1e-8, 1 - 1e-8)
    return torch.sign((b_prev * n_b).sum(-1)) * torch.acos(cosang)


def backbone_dihedrals(X: torch.Tensor, residue_index: torch.Tensor):
    """(pre-omega, phi, psi) per residue and their validity mask (helper.py:39-74)."""
    L = X.shape[0]
    chain = X[:, :3].reshape(3 * L, 3)
    d = F.pad(dihedrals_along(chain), [1, 2], value=float("nan")).reshape(L, 3)   # phi, psi, omega
    nan1 = torch.tensor([float("nan")])
    zero1 = torch.tensor([0.0])
    follows_prev = torch.cat((zero1, (residue_index[1:] - 1 == residue_index[:-1]).float()))
    precedes_next = torch.cat(((residue_index[:-1] + 1 == residue_index[1:]).float(), zero1))
    pre_omega = torch.cat((nan1, d[:-1, 2]))
    bb = torch.stack((pre_omega, d[:, 0], d[:, 1]), dim=-1)
    mask = torch.stack((follows_prev, follows_prev, precedes_next), dim=-1)
    mask = mask * torch.isfinite(bb).float()
    return bb, mask


def sidechain_dihedrals(X: torch.Tensor, aatype: torch.Tensor):
    """chi1..chi4 from atom14 coordinates and their mask (helper.py:77-101)."""
    idx = torch.from_numpy(rc.chi_atom_indices_atom14)[aatype]                    # [L,7]
    cmask = torch.from_numpy(rc.chi_mask_atom14)[aatype]                          # [L,4]
    pts = torch.gather(X, -2, idx[..., None].expand(*idx.shape, 3))
    chi = torch.nan_to_num(dihedrals_along(pts)) * cmask
    return chi, (chi != 0.0).float()


def chain_numbers_and_offset_index(protein: Dict):
    """(chain number 1.. in order of first appearance, residue_index with every later chain pushed +100 past the already
    shifted end of the previous one) -- complex_dataset.py:80-93.  The protein dict is left untouched."""
    L = len(protein["aaindex"])
    rindex = torch.from_numpy(np.asarray(protein["residue_index"])).long().clone()
    seen = {}
    chain_np = np.empty(L, np.int64)
    for i, c in enumerate(list(protein["chain_id"])):
        chain_np[i] = seen.setdefault(c, len(seen) + 1)
    chain = torch.from_numpy(chain_np)
    if len(seen) > 1:
        shift = 0
        for c in range(1, len(seen)):
            shift += int(rindex[chain == c].max()) + 100
            rindex[chain == c + 1] += shift
    return chain, rindex


def protein_to_data(protein: Dict) -> Batch:
    """Per-complex tensors (no batch axis), as ``prot_to_data`` lays them out."""
    X = torch.from_numpy(np.asarray(protein["atom_positions"])).float()
    L = X.shape[0]
    rtype = torch.from_numpy(np.asarray(protein["aaindex"])).long()
    amask = torch.from_numpy(np.asarray(protein["atom_mask"])).float()
    chain, rindex = chain_numbers_and_offset_index(protein)

    rmask = torch.isfinite(X[:, :4].sum(dim=(-1, -2))).float()
    bb, bb_mask = backbone_dihedrals(X, rindex)
    sc, sc_mask = sidechain_dihedrals(X, rtype)
    bb_sc = torch.stack((bb.sin(), bb.cos()), -1) * bb_mask[..., None]
    sc_sc = torch.stack((sc.sin(), sc.cos()), -1) * sc_mask[..., None]
    pi1 = torch.from_numpy(rc.chi_pi_periodic)[rtype].bool()
    pi2 = ~pi1

    m1, m2, m3 = rmask, rmask[:, None], rmask[:, None, None]
    sc_mask = sc_mask * m2
    data = Batch(
        num_nodes=L,
        X=X * m3,
        atom_mask=amask * m2,
        residue_type=(rtype * m1).long(),
        residue_mask=rmask,
        residue_index=(rindex * m1).long(),
        chain_indices=(chain * m1).long(),
        BB_D=bb * m2,
        BB_D_sincos=bb_sc * m3,
        BB_D_mask=bb_mask * m2,
        SC_D=sc * m2,
        SC_D_sincos=sc_sc * m3,
        SC_D_mask=sc_mask,
        chi_1pi_periodic_mask=torch.logical_and(sc_mask, pi1 * m2),
        chi_2pi_periodic_mask=torch.logical_and(sc_mask, pi2 * m2),
    )
    data.apply(lambda v: torch.nan_to_num(v) if isinstance(v, torch.Tensor) and v.is_floating_point() else v)
    return data


def protein_to_batch(protein: Dict) -> Batch:
    """B=1 batch, the form ``ProteinAnalysis.get_prot`` hands to ``sampling`` (protein_analysis.py:103-122)."""
    return as_single(protein_to_data(protein))
