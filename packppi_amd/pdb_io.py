"""PDB text <-> protein dict, without Biopython.

Read side reproduces what ``from_pdb_file``/``from_pdb_string`` (src/utils/protein.py:55-199)
obtain through Biopython 1.84's permissive ``PDBParser``: only ``ATOM`` records; chains
sorted by id; residues stably sorted by sequence number; alternate locations resolved to
the highest occupancy (first wins ties); a repeated atom name keeps its first record;
non-standard residues and atoms outside the residue's atom14 slots (all hydrogens) are
dropped; insertion codes shift later indices by one each; duplicate indices inside a
chain are bumped.  Biopython itself is not in this image, so this step is
**parity-unpinned**  -- everything downstream of the dict is pinned.

Write side reproduces ``to_pdb`` (protein.py:207-314) column for column.
"""
import gzip
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

from . import constants as rc


def _atom_records(path) -> List[str]:
    path = str(path)
    if path.endswith("pdb.gz"):
        with gzip.open(path, "rb") as fh:
            raw = [ln.decode() for ln in fh]
    elif path.endswith("pdb"):
        with open(path, "r") as fh:
            raw = list(fh)
    else:
        raise ValueError("Unrecognized file type.")
    return [ln.strip() for ln in raw if ln.startswith("ATOM")]


def parse_atom_records(lines: List[str], mse_to_met: bool = True, ignore_non_std: bool = True) -> Dict:
    # chain -> ordered list of residues; residue = dict(resseq, icode, resname, atoms{name: [occ, xyz, b]})
    chains: Dict[str, List[dict]] = {}
    by_id: Dict[tuple, dict] = {}
    for ln in lines:
        ln = ln.ljust(80)
        fullname = ln[12:16]
        parts = fullname.split()
        name = parts[0] if len(parts) == 1 else fullname
        altloc = ln[16]
        resname = ln[17:20].strip()
        chain_id = ln[21]
        try:
            resseq = int(ln[22:26].split()[0])
            xyz = (float(ln[30:38]), float(ln[38:46]), float(ln[46:54]))
        except (ValueError, IndexError):
            continue
        icode = ln[26]
        try:
            occ = float(ln[54:60])
        except ValueError:
            occ = None
        try:
            bfac = float(ln[60:66])
        except ValueError:
            bfac = 0.0
        key = (chain_id, resseq, icode)
        res = by_id.get(key)
        if res is None or res["resname"] != resname:
            if res is not None:
                # same id, different residue name (point-mutation microheterogeneity): keep the first
                continue
            res = dict(resseq=resseq, icode=icode, resname=resname, atoms={})
            by_id[key] = res
            chains.setdefault(chain_id, []).append(res)
        slot = res["atoms"].get(name)
        occ_cmp = -1.0 if occ is None else occ
        if slot is None:
            res["atoms"][name] = [occ_cmp, np.float32(xyz), bfac, altloc != " "]
        elif altloc != " " and slot[3] and occ_cmp > slot[0]:
            res["atoms"][name] = [occ_cmp, np.float32(xyz), bfac, True]
        # else: atom defined twice / lower-or-equal occupancy altloc -> first record stays

    pos_l, aa_l, mask_l, idx_l, chain_l, bf_l = [], [], [], [], [], []
    icode_shift = 0
    for cid in sorted(chains):
        for res in sorted(chains[cid], key=lambda r: r["resseq"]):
            resname, atoms = res["resname"], res["atoms"]
            if resname == "HOH":
                continue
            if mse_to_met and resname == "MSE":
                resname = "MET"
                if "SE" in atoms:
                    atoms = {("SD" if k == "SE" else k): v for k, v in atoms.items()}
            rt = rc.resname_to_idx.get(resname, 20)
            if rt == 20:
                if ignore_non_std:
                    continue
            if res["icode"] != " ":
                icode_shift += 1
            names = rc.atom14_names[rt]
            pos = np.full((14, 3), np.nan)
            mask = np.zeros(14)
            bf = np.zeros(14)
            for aname, (_, xyz, b, _) in atoms.items():
                if aname and aname in names:
                    k = names.index(aname)
                    pos[k], mask[k], bf[k] = xyz, 1.0, b
            if mask.sum() < 0.5:
                continue
            pos_l.append(pos); aa_l.append(rt); mask_l.append(mask)
            idx_l.append(res["resseq"] + icode_shift); chain_l.append(cid); bf_l.append(bf)

    used: Dict[str, set] = {}
    new_idx = []
    for cid, idx in zip(chain_l, idx_l):
        taken = used.setdefault(cid, set())
        while idx in taken:
            idx += 1
        taken.add(idx)
        new_idx.append(idx)

    return dict(atom_positions=np.array(pos_l), atom_mask=np.array(mask_l), aaindex=np.array(aa_l),
                residue_index=np.array(new_idx), chain_id=np.array(chain_l), b_factors=np.array(bf_l))


def from_pdb_file(pdb_file, mse_to_met: bool = True) -> Dict:
    return parse_atom_records(_atom_records(Path(pdb_file)), mse_to_met=mse_to_met)


def contains_sidechains(pdb_file) -> bool:
    """eval_diffusion.py:43-50."""
    with open(pdb_file, "r") as fh:
        for ln in fh:
            if ln.startswith("ATOM") and ln[12:16].strip() in rc.sidechain_atoms:
                return True
    return False


def _ter(serial, resname, chain, resnum) -> str:
    return f"{'TER':<6}{serial:>5}      {resname:>3} {chain:>1}{resnum:>4}"


def to_pdb(prot: Dict, keep_chains: Optional[list] = None) -> str:
    amask, aa = np.asarray(prot["atom_mask"]), np.asarray(prot["aaindex"])
    xyz, ridx = np.asarray(prot["atom_positions"]), np.asarray(prot["residue_index"])
    chain, bfac = np.asarray(prot["chain_id"]), np.asarray(prot["b_factors"])
    if np.any(aa > 20):
        raise ValueError("Invalid aaindexs.")
    if keep_chains is not None:
        sel = np.isin(chain, keep_chains)
        amask, aa, xyz, ridx, chain, bfac = amask[sel], aa[sel], xyz[sel], ridx[sel], chain[sel], bfac[sel]
    if xyz.shape[-2] != 14:
        raise ValueError("Invalid number of atoms per residue.")

    out = ["MODEL     1"]
    serial = 1
    prev_chain = chain[0]
    for i in range(aa.shape[0]):
        if chain[i] != prev_chain:
            out.append(_ter(serial, rc.resnames[aa[i - 1]], chain[i - 1], ridx[i - 1]))
            prev_chain = chain[i]
            serial += 1
        rname = rc.resnames[aa[i]]
        for aname, p, m, b in zip(rc.atom14_names[aa[i]], xyz[i], amask[i], bfac[i]):
            if m < 0.5:
                continue
            shown = aname if len(aname) == 4 else f" {aname}"
            out.append(f"{'ATOM':<6}{serial:>5} {shown:<4}{'':>1}{rname:>3} {chain[i]:>1}"
                       f"{ridx[i]:>4}{'':>1}   {p[0]:>8.3f}{p[1]:>8.3f}{p[2]:>8.3f}"
                       f"{1.0:>6.2f}{b:>6.2f}          {aname[0]:>2}{'':>2}")
            serial += 1
    out.append(_ter(serial, rc.resnames[aa[-1]], chain[-1], ridx[-1]))
    out += ["ENDMDL", "END"]
    return "\n".join(ln.ljust(80) for ln in out) + "\n"
