"""The ``batch`` object the sampler consumes.

The reference uses a ``torch_geometric.data.Data`` filled by ``prot_to_data``
(complex_dataset.py:123-139) and padded/stacked by ``collate_fn``
(complex_datamodule.py:196-226).  This is a dependency-free container with the same
attribute / item access and ``.to(device)``.
"""
from typing import Iterable, List

import torch
import torch.nn.functional as F

TENSOR_KEYS = (
    "X", "atom_mask", "residue_type", "residue_mask", "residue_index", "chain_indices",
    "BB_D", "BB_D_sincos", "BB_D_mask", "SC_D", "SC_D_sincos", "SC_D_mask",
    "chi_1pi_periodic_mask", "chi_2pi_periodic_mask",
)


class Batch(dict):
    """Attribute-style dict: ``batch.X``, ``batch['X']``, ``batch.to('cuda')``, ``batch.keys()``."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def to(self, device):
        out = Batch()
        for k, v in self.items():
            out[k] = v.to(device) if isinstance(v, torch.Tensor) else v
        return out

    def clone(self):
        out = Batch()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, torch.Tensor) else v
        return out

    def apply(self, fn):
        for k in list(self.keys()):
            self[k] = fn(self[k])
        return self

    def true_residues(self) -> int:
        return int(self["residue_mask"].sum().item())


def as_single(data: Batch) -> Batch:
    """Add the leading batch axis the way ``ProteinAnalysis.get_prot`` does (protein_analysis.py:115-120)."""
    out = Batch()
    for k, v in data.items():
        out[k] = v.unsqueeze(0) if isinstance(v, torch.Tensor) else v
    out["num_proteins"] = 1
    out["max_size"] = int(data["num_nodes"])
    return out


def collate(proteins: Iterable[Batch]) -> Batch:
    """Pad every per-residue tensor to the longest complex and stack (complex_datamodule.py:196-226)."""
    proteins = list(proteins)
    max_size = max(int(p["num_nodes"]) for p in proteins)

    def pad(p, key):
        t = p[key]
        return F.pad(t, [0, 0] * (t.dim() - 1) + [0, max_size - int(p["num_nodes"])])

    out = Batch(num_proteins=len(proteins), max_size=max_size)
    for key in TENSOR_KEYS:
        out[key] = torch.stack([pad(p, key) for p in proteins])
    return out


def split(batch: Batch) -> List[Batch]:
    """Inverse of ``collate`` up to padding: one B=1 batch per complex (padding kept)."""
    outs = []
    for b in range(int(batch["num_proteins"])):
        o = Batch(num_proteins=1, max_size=int(batch["max_size"]))
        for k in TENSOR_KEYS:
            o[k] = batch[k][b:b + 1]
        outs.append(o)
    return outs
