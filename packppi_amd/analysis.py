"""``ProteinAnalysis``: PDB -> batch, metrics and MolProbity clashscore around the sampler.

Mirrors src/utils/protein_analysis.py: ``get_prot`` (:103-122), ``get_metric`` (:36-91), ``compute_rmsd``
(:93-101), ``get_clashscore`` (:26-34).  ``run_tool`` (SCWRL/FASPR/Rosetta wrappers) is out of scope.
Interface residues (src/utils/interface.py:11-56: any atom within 10 A of an atom of another chain) are found
with a KD-tree instead of Biopython's NeighborSearch; like the PDB reader this step is parity-unpinned.
"""
import os
import subprocess
from pathlib import Path

import numpy as np
import torch

from .featurize import chain_numbers_and_offset_index, protein_to_batch
from .functional import get_atom14_coords
from .pdb_io import from_pdb_file


def interface_residues(pdb_file, radius=10.0):
    """{chain: sorted residue numbers having an atom within ``radius`` of another protein chain} or None."""
    from scipy.spatial import cKDTree
    xyz, chain, resseq, is_std = [], [], [], {}
    with open(pdb_file) as fh:
        for ln in fh:
            if ln.startswith("ENDMDL"):
                break
            if not ln.startswith(("ATOM", "HETATM")):
                continue
            try:
                p = (float(ln[30:38]), float(ln[38:46]), float(ln[46:54]))
                rs = int(ln[22:26])
            except ValueError:
                continue
            c = ln[21]
            xyz.append(p); chain.append(c); resseq.append(rs)
            is_std[c] = is_std.get(c, False) or ln.startswith("ATOM")
    chains = [c for c, ok in is_std.items() if ok]
    if len(chains) < 2:
        return None
    keep = np.array([c in chains for c in chain])
    xyz, chain, resseq = np.array(xyz)[keep], np.array(chain)[keep], np.array(resseq)[keep]
    tree = cKDTree(xyz)
    pairs = tree.query_pairs(radius, output_type="ndarray")
    cross = chain[pairs[:, 0]] != chain[pairs[:, 1]]
    out = {c: set() for c in chains}
    for side in (0, 1):
        idx = pairs[cross, side]
        for c in chains:
            out[c].update(resseq[idx[chain[idx] == c]].tolist())
    return {c: sorted(v) for c, v in out.items()}


def interface_mask(protein, pdb, radius=10.0):
    """helper.py:104-128."""
    if len(np.unique(protein["chain_id"])) == 1:
        return None
    inter = interface_residues(pdb, radius)
    if inter is None:
        return None
    parts = []
    for cid in np.unique(protein["chain_id"]):
        sub = protein["residue_index"][protein["chain_id"] == cid]
        parts.append(np.isin(sub, inter[cid]) if cid in inter else np.zeros(len(sub), bool))
    return torch.from_numpy(np.concatenate(parts)).float()


class ProteinAnalysis:
    def __init__(self, molprobity_clash_loc, tmp_dir, device="cuda"):
        self.molprobity_clash_loc = molprobity_clash_loc
        self.device = device
        self.tmp_dir = tmp_dir
        os.makedirs(self.tmp_dir, exist_ok=True)
        self.tmp_log = os.path.join(tmp_dir, "molprobity_clash.log")
        self.tmp_pdb = os.path.join(tmp_dir, "structure.pdb")

    def get_clashscore(self, pdb):
        """External MolProbity binary; returns None when it is unavailable or prints no score."""
        if not self.molprobity_clash_loc or not os.path.exists(str(self.molprobity_clash_loc).split()[0]):
            return None
        cmd = f"{self.molprobity_clash_loc} model={pdb} keep_hydrogens=True > {self.tmp_log}"
        subprocess.run(cmd, shell=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        try:
            with open(self.tmp_log) as fh:
                for ln in fh:
                    if "clashscore" in ln and "= " in ln:
                        txt = ln.split("= ")[1].strip()
                        return float(txt) if txt.replace(".", "").isdigit() else None
        except OSError:
            pass
        return None

    def get_prot(self, pdb, get_interface=True):
        protein = from_pdb_file(Path(pdb), mse_to_met=True)
        data = protein_to_batch(protein)
        if get_interface:
            # Reference behaviour kept: prot_to_data shifts the residue numbers of every chain after the first IN PLACE in the
            # protein dict it is given (complex_dataset.py:79,91: torch.from_numpy(...).to(int64) shares memory with the
            # int64 array the PDB reader returns), and get_interface_mask, called after it (protein_analysis.py:106-108),
            # compares those shifted numbers with the file's own numbering -- so only residues of the first chain (and
            # accidental number matches) can enter the interface mask, and interface_acc is an accuracy over them.
            shifted = dict(protein, residue_index=chain_numbers_and_offset_index(protein)[1].numpy())
            im = interface_mask(shifted, pdb)
            rm = data.residue_mask[0]
            data["interface_mask"] = ((im * rm) if im is not None else torch.zeros_like(rm)).unsqueeze(0)
        return data

    def compute_rmsd(self, true_coords, pred_coords, atom_mask, residue_mask, eps=1e-6):
        w = atom_mask * residue_mask[..., None]
        return (torch.sum((true_coords - pred_coords) ** 2, dim=-1) * w).sum() / (w + eps).sum()

    def get_metric(self, true_pdb, pred_pdb):
        try:
            true_data = self.get_prot(true_pdb, get_interface=True)
            pred_data = self.get_prot(pred_pdb)
        except Exception as e:  # noqa: BLE001  (the reference reports and returns None)
            print(f"Error: Failed to load or parse PDB files. Details: {str(e)}")
            return None
        if true_data.X.shape[1] != pred_data.X.shape[1]:
            print("Error: Mismatch in the number of residues between true and predicted structures.")
            return None
        clashscore = self.get_clashscore(pred_pdb)
        imask = true_data.interface_mask
        ct, cp, cm, pi1 = true_data.SC_D, pred_data.SC_D, true_data.SC_D_mask, true_data.chi_1pi_periodic_mask
        metric, total_acc, interface_acc = {}, 0, 0
        for i in range(4):
            n = cm[..., i].sum()
            n = n if n != 0 else 1
            ni = (cm[..., i] * imask).sum()
            ni = ni if ni != 0 else 1
            diff = (cp[..., i] - ct[..., i]).abs()
            acc = torch.logical_and(diff * 180 / np.pi < 20, diff > 0).float()
            ae = torch.minimum(diff, 2 * np.pi - diff)
            ae = torch.where(pi1[..., i], torch.minimum(ae, np.pi - ae), ae)
            metric[f"chi_{i}_ae_rad"] = ae.sum() / n
            metric[f"chi_{i}_ae_deg"] = (ae * 180 / np.pi).sum() / n
            metric[f"chi_{i}_acc"] = acc.sum() / n
            total_acc += acc.sum() / n
            interface_acc += (acc * imask).sum() / ni
        d = torch.device(self.device)
        pred_xyz = get_atom14_coords(true_data.X.to(d), true_data.residue_type.to(d), true_data.BB_D.to(d),
                                     pred_data.SC_D.to(d)).cpu()
        metric["atom_rmsd"] = self.compute_rmsd(true_data.X, pred_xyz, true_data.atom_mask, true_data.residue_mask)
        metric["total_acc"] = total_acc / 4
        metric["interface_acc"] = interface_acc / 4
        metric["clashscore"] = clashscore
        return metric
