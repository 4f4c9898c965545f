"""``proximal_optimize`` command line (src/proximal_optimize.py:26-78): clash-relax the side chains of a PDB.

Flags as in the reference (:69-78) plus --device (the reference script is CPU only; this path is HIP only).
"""
import argparse
from pathlib import Path

from ..analysis import ProteinAnalysis
from ..functional import get_atom14_coords, proximal_optimizer
from ..pdb_io import contains_sidechains, from_pdb_file, to_pdb


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--input", type=str, help="The input pdb file path.", required=True)
    p.add_argument("--outdir", type=str, help="Directory to store outputs.", required=True)
    p.add_argument("--molprobity_clash_loc", type=str, help="Path to /build/bin/molprobity.clashscore.", required=True)
    p.add_argument("--violation_tolerance_factor", type=float, help="The violation tolerance factor.", default=12)
    p.add_argument("--clash_overlap_tolerance", type=float, help="Acceptable deviation between atoms.", default=0.5)
    p.add_argument("--lamda", type=float, help="The influence of the proximal term on the gradient.", default=1)
    p.add_argument("--num_steps", type=int, help="Number of optimize steps.", default=50)
    p.add_argument("--device", type=str, default="cuda")
    args = p.parse_args(argv)

    assert contains_sidechains(args.input), "----- No side chain atoms found in the input PDB -----"
    print("----- Starting optimize! -----")
    analysis = ProteinAnalysis(args.molprobity_clash_loc, args.outdir, args.device)
    print(f"----- The input structure clashscore is {analysis.get_clashscore(args.input)} -----")
    protein = from_pdb_file(Path(args.input), mse_to_met=True)
    batch = analysis.get_prot(args.input).to(args.device)
    chis, losses = proximal_optimizer(batch, batch.SC_D, args.violation_tolerance_factor,
                                      args.clash_overlap_tolerance, args.lamda, args.num_steps)
    SC_D = chis[-1] if losses[-1] < losses[0] else batch.SC_D
    xyz = get_atom14_coords(batch.X, batch.residue_type, batch.BB_D, SC_D)
    protein["atom_positions"] = xyz.cpu().squeeze(0).numpy()
    with open(analysis.tmp_pdb, "w") as fh:
        fh.writelines(to_pdb(protein))
    print(f"----- The optimized structure clashscore is {analysis.get_clashscore(analysis.tmp_pdb)} -----")
    print("----- Finishing optimize! -----")


if __name__ == "__main__":
    main()
