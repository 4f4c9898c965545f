"""Container-only: dump the residue-chemistry DATA tables the hot path needs.

Reads the reference's `src/utils/residue_constants.py` (AlphaFold2 / Engh-Huber public
constants, rc:29-240,280-285,459-554,595-677,709-806) through `refshim` and writes the
NUMERIC tables to `packppi_amd/data/residue_constants.npz`.  Only data leaves this
script (arrays of numbers and atom names) -- no reference code.

    PYTHONDONTWRITEBYTECODE=1 python tools/oracle/make_constants.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(__file__))
import refshim  # noqa: F401,E402
import src.utils.residue_constants as rc  # noqa: E402

OUT = os.path.join(os.path.dirname(__file__), "..", "..", "packppi_amd", "data",
                   "residue_constants.npz")


def main():
    resnames = [rc.restype_1to3[r] for r in rc.restypes] + ["UNK"]
    atom14_names = np.array([[n for n in rc.restype_name_to_atom14_names[r]] for r in resnames])

    # (restype, a, b, length, stddev) rows for real + virtual bonds, in the order
    # make_atom14_dists_bounds (rc:809-869) applies them.
    bonds, vbonds, _ = rc.load_stereo_chemical_props()
    rows = []
    for ri, rn in enumerate(resnames[:20]):
        names = rc.restype_name_to_atom14_names[rn]
        for b in bonds[rn] + vbonds[rn]:
            rows.append((ri, names.index(b.atom1_name), names.index(b.atom2_name),
                         float(b.length), float(b.stddev)))
    bond_rows = np.array(rows, dtype=np.float64)

    # element vdW radius per atom14 slot (0 where the slot is empty), rc:280-285
    slot_radius = np.zeros((21, 14), np.float64)
    for ri, rn in enumerate(resnames):
        for ai, n in enumerate(rc.restype_name_to_atom14_names[rn]):
            if n:
                slot_radius[ri, ai] = rc.van_der_waals_radius[n[0]]

    # radius table as find_sc_violations builds it (clash.py:263-287): empty slots map to
    # atom37 index 0 ("N", 1.55 A) and are zeroed at run time by atom_exists.
    atomtype_radius = np.array([rc.van_der_waals_radius[n[0]] for n in rc.atom_types])
    a14_to_a37 = np.zeros((21, 14), np.int64)
    for ri, rn in enumerate(resnames[:20]):
        for ai, n in enumerate(rc.restype_name_to_atom14_names[rn]):
            a14_to_a37[ri, ai] = rc.atom_order[n] if n else 0
    between_radius = atomtype_radius[a14_to_a37]

    chi_angles_mask = np.array(list(rc.chi_angles_mask) + [[0.0] * 4], np.float32)
    np.savez_compressed(
        OUT,
        restypes=np.array(rc.restypes),
        resnames=np.array(resnames),
        atom14_names=atom14_names,
        default_frames=rc.restype_rigid_group_default_frame.astype(np.float32),
        atom14_to_group=rc.restype_atom14_to_rigid_group.astype(np.int64),
        atom14_mask=rc.restype_atom14_mask.astype(np.float32),
        lit_positions=rc.restype_atom14_rigid_group_positions.astype(np.float32),
        chi_angles_mask=chi_angles_mask,
        chi_pi_periodic=np.array(rc.chi_pi_periodic, np.float32),
        chi_atom_indices_atom14=np.array(rc.chi_atom_indices_atom14, np.int64),
        chi_mask_atom14=np.array(rc.chi_mask_atom14, np.float32),
        bond_rows=bond_rows,
        slot_radius=slot_radius,
        between_radius=between_radius.astype(np.float64),
        sidechain_atoms=np.array(sorted(rc.sidechain_atoms)),
    )
    # self-check data for the run-time bounds builder (two parameter settings)
    chk = {}
    for tol, vtf in ((0.5, 12.0), (1.5, 15.0), (0.1, 12.0)):
        b = rc.make_atom14_dists_bounds(overlap_tolerance=tol, bond_length_tolerance_factor=vtf)
        chk[f"lower_{tol}_{vtf}"] = b["lower_bound"]
        chk[f"upper_{tol}_{vtf}"] = b["upper_bound"]
    np.savez_compressed(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden",
                                     "g1_dists_bounds.npz"), **chk)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
