"""Container-only: goldens for the writer / metric harness -> tests/golden/g8_io.npz.

    PYTHONDONTWRITEBYTECODE=1 python tools/oracle/make_golden_io.py

  * ``to_pdb``: the UNMODIFIED reference writer (src/utils/protein.py:207-314) on the protein dicts of 1BRS and 2FTL
    stored in g0_protein_*.npz -- the full expected text of 1BRS and the sha256 of both.
  * ``get_metric``: the UNMODIFIED reference ``ProteinAnalysis.get_metric`` (protein_analysis.py:36-91) on a true / predicted
    PDB pair written by that writer from the 1BRS dict and a side-chain-perturbed copy of it.  Three things the image
    lacks are replaced, and only for this fixture: the Biopython PDB reader by the build's own reader (the reference's
    ``prot_to_data`` then runs on its dict, as in g0); ``get_interface_residues`` (Biopython NeighborSearch) by the
    definition it implements, computed by brute force (two residues of different chains with any atom pair within
    10 A); MolProbity's clashscore by a constant.  Everything between -- chi recomputed from the 3-decimal file,
    accuracies, interface mask handling, atom_rmsd -- is the reference's own code.
"""
import hashlib
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refshim  # noqa: E402
from packppi_amd.pdb_io import from_pdb_file  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CLASHSCORE = 12.34


def brute_force_interface(pdb_file, radius=10.0):
    """{chain: sorted residue numbers with an atom within `radius` of an atom of another chain} -- the definition behind
    NeighborSearch.search_all(radius, 'R') as interface.py:11-56 uses it."""
    xyz, chain, res = [], [], []
    for ln in open(pdb_file):
        if ln.startswith("ENDMDL"):
            break
        if ln.startswith("ATOM"):
            xyz.append((float(ln[30:38]), float(ln[38:46]), float(ln[46:54])))
            chain.append(ln[21])
            res.append(int(ln[22:26]))
    xyz, chain, res = np.array(xyz), np.array(chain), np.array(res)
    out = {c: set() for c in np.unique(chain)}
    for i in range(0, len(xyz), 512):
        d = np.linalg.norm(xyz[i:i + 512, None] - xyz[None], axis=-1)
        a, b = np.nonzero((d <= radius) & (chain[i:i + 512, None] != chain[None]))
        for ai, bi in zip(a + i, b):
            out[chain[ai]].add(int(res[ai]))
            out[chain[bi]].add(int(res[bi]))
    return {c: sorted(v) for c, v in out.items()}


def main():
    from src.utils.protein import to_pdb
    out = {}
    prots = {}
    for tag in ("1BRS", "2FTL"):
        z = np.load(os.path.join(GOLD, f"g0_protein_{tag}.npz"), allow_pickle=False)
        prot = {k[5:]: z[k] for k in z.files if k.startswith("prot.")}
        prots[tag] = prot
        text = to_pdb(prot)
        out[f"sha256.{tag}"] = np.array(hashlib.sha256(text.encode()).hexdigest())
        if tag == "1BRS":
            out["text.1BRS"] = np.frombuffer(text.encode(), dtype=np.uint8)
        print(tag, len(text), "bytes", out[f"sha256.{tag}"])

    # ---- get_metric on a written pair ----------------------------------------------------------------------------------
    import src.utils.protein_analysis as PA
    import src.datamodules.components.helper as H
    from src.models.components import get_atom14_coords
    from src.datamodules.components.complex_dataset import ComplexDataset
    prot = prots["1BRS"]
    data = ComplexDataset.prot_to_data({k: (v.copy() if hasattr(v, "copy") else v) for k, v in prot.items()}, cache_processed_data=False)
    g = torch.Generator().manual_seed(8)
    chi_pred = data.SC_D + 0.35 * torch.randn(data.SC_D.shape, generator=g) * data.SC_D_mask     # a plausible prediction
    xyz = get_atom14_coords(data.X[None], data.residue_type[None], data.BB_D[None], chi_pred[None])[0]
    pred = dict(prot)
    pred["atom_positions"] = (xyz * data.atom_mask[..., None]).numpy().astype(prot["atom_positions"].dtype)
    out["pred.atom_positions"] = pred["atom_positions"]
    tmp = tempfile.mkdtemp()
    true_pdb, pred_pdb = os.path.join(tmp, "true.pdb"), os.path.join(tmp, "pred.pdb")
    open(true_pdb, "w").write(to_pdb(prot))
    open(pred_pdb, "w").write(to_pdb(pred))
    PA.from_pdb_file = lambda p, mse_to_met=True: types.SimpleNamespace(**from_pdb_file(p, mse_to_met=mse_to_met))
    H.get_interface_residues = brute_force_interface
    pa = PA.ProteinAnalysis("unused", tmp)
    pa.get_clashscore = lambda pdb: CLASHSCORE
    metric = pa.get_metric(true_pdb, pred_pdb)
    assert metric is not None
    for k, v in metric.items():
        out["metric." + k] = np.float64(v)
        print(f"  {k}: {float(v):.6f}")
    im = pa.get_prot(true_pdb, get_interface=True).interface_mask
    out["interface_mask"] = im.numpy()
    print("  interface residues:", int(im.sum()), "of", im.numel())
    path = os.path.join(GOLD, "g8_io.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) / 1e6, "MB")


if __name__ == "__main__":
    main()
