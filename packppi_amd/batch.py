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


def pack(complexes: Iterable[Batch]) -> Batch:
    """Ragged batch WITHOUT padding rows: the complexes' rows back to back in one [1, sum of lengths, ...] batch, plus
    ``seg_offsets`` (int32 [n + 1]: first row of every complex, then the total; ``seg_offsets_host`` is the same as a list,
    so that nothing has to be read back from the device).

    The reference pads to the longest complex and stacks (``collate``; complex_datamodule.py:196-226) and then computes on
    the padding rows as well; the path has no cross-complex term, so a packed batch gives every complex the result of
    running it alone (``lib.Context`` / ``pp_complex_prepare_packed``).  Accepts per-complex data (``protein_to_data``,
    tensors [L, ...]) or B = 1 batches ([1, L, ...]).  Only TRAILING padding is dropped: a complex keeps its rows up to the
    last true residue, so a residue masked out in the middle of a chain (a missing backbone atom: featurize.py) stays in
    place with its ``residue_mask`` 0, exactly as the reference carries it."""
    complexes = list(complexes)
    rows = {k: [] for k in TENSOR_KEYS}
    offs = [0]
    ends = []
    for c in complexes:
        lead = c["residue_type"].dim() == 2
        if lead and c["residue_type"].shape[0] != 1:
            raise ValueError("pack() takes single complexes (use split() on a padded batch first)")
        keep = (c["residue_mask"][0] if lead else c["residue_mask"]) > 0
        # index of the last true residue + 1 (0 for an empty complex), computed where the mask lives
        ends.append((keep * torch.arange(1, keep.numel() + 1, device=keep.device)).max())
    ends = [int(x) for x in torch.stack(ends).tolist()]            # ONE read-back for the whole batch
    for c, n in zip(complexes, ends):
        if n == 0:
            raise ValueError("empty complex")
        lead = c["residue_type"].dim() == 2
        for k in TENSOR_KEYS:
            t = c[k][0] if lead else c[k]
            rows[k].append(t[:n])
        offs.append(offs[-1] + n)
    out = Batch(num_proteins=1, max_size=offs[-1])
    for k in TENSOR_KEYS:
        out[k] = torch.cat(rows[k], 0).unsqueeze(0)
    out["seg_offsets"] = torch.tensor(offs, dtype=torch.int32).to(out["X"].device, non_blocking=True)
    out["seg_offsets_host"] = offs
    return out


def unpack(packed: Batch, t: torch.Tensor) -> List[torch.Tensor]:
    """Split a per-row result [1, sum of lengths, ...] of a packed batch into one [1, L_i, ...] tensor per complex."""
    offs = packed.get("seg_offsets_host") or [int(x) for x in packed["seg_offsets"].tolist()]
    return [t[:, a:b] for a, b in zip(offs[:-1], offs[1:])]
