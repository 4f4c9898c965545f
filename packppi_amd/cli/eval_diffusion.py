"""``eval_diffusion`` command line: PDB in -> sampled side chains -> OUTDIR/structure.pdb + metrics.

Same required flags as the reference's src/eval_diffusion.py:86-93 (--input, --outdir, --molprobity_clash_loc,
--use_proximal, --device).  The reference composes configs/eval_diffusion.yaml with Hydra and takes its checkpoint path and
the encoder / model / sampling settings from there (eval_diffusion.py:22-41); here --config_dir (or $PACKPPI_CONFIG_DIR)
names that configs/ directory and the four hot-path YAML files are read as plain YAML (packppi_amd/config.py): sample_cfg
(mode, annealed_temp, proximal parameters) is applied, the dimensions are checked against what the kernels are compiled for.
--ckpt_path / $PACKPPI_CKPT override the tree's ckpt_path; --steps exposes the number of diffusion steps (reference: 30).
"""
import argparse
import os
from pathlib import Path

import torch

from ..analysis import ProteinAnalysis
from ..functional import get_atom14_coords
from ..module import TDiffusionModule
from ..pdb_io import contains_sidechains, from_pdb_file, to_pdb


def load_model(args):
    from ..config import load_hot_path_configs, resolve_ckpt
    config_dir = args.config_dir or os.environ.get("PACKPPI_CONFIG_DIR")
    cfgs = load_hot_path_configs(config_dir) if config_dir else None
    cfg_kw = dict(encoder_cfg=cfgs.encoder_cfg, model_cfg=cfgs.model_cfg, sample_cfg=cfgs.sample_cfg) if cfgs else {}
    if cfgs is not None:
        print(f"----- Using the configuration tree {cfgs.config_dir} -----")
    ckpt = resolve_ckpt(args.ckpt_path, cfgs)
    if args.random_weights is not None:
        from ..weights import make_random_state_dict
        print(f"----- Using seeded random weights (seed {args.random_weights}); no checkpoint given! -----")
        model = TDiffusionModule(make_random_state_dict(args.random_weights), device=args.device, **cfg_kw)
    else:
        assert ckpt is not None and os.path.exists(ckpt), "Invalid checkpoint path!"
        print(f"----- Loading {ckpt} checkpoint! -----")
        model = TDiffusionModule.load_from_checkpoint(ckpt, map_location=args.device, strict=False, **cfg_kw)
    if args.steps is not None:
        model.schedule = torch.linspace(1, 0, args.steps + 1)
    return model.eval()


def evaluate_model(model, args):
    print("----- Starting evaluation! -----")
    analysis = ProteinAnalysis(args.molprobity_clash_loc, args.outdir, args.device)
    protein = from_pdb_file(Path(args.input), mse_to_met=True)
    batch = analysis.get_prot(args.input).to(args.device)
    if args.seed is not None:
        torch.manual_seed(args.seed)
    SC_D_sample = model.sampling(batch, use_proximal=args.use_proximal)
    if model.saturated() & 4:
        print("----- WARNING: NaN / infinity in the input coordinates or angles: the reference would return NaN here -----")
    if model.saturated() & 3:
        # a hidden activation reached the f16 maximum in the split-f16 dense layers: not the reference's arithmetic any more
        print("----- WARNING: f16 saturation in the score network (flag %d); run `python -m packppi_amd.rangecheck` on this "
              "checkpoint -----" % model.saturated())
    xyz = get_atom14_coords(batch.X, batch.residue_type, batch.BB_D, SC_D_sample)
    protein["atom_positions"] = xyz.cpu().squeeze(0).numpy()
    with open(analysis.tmp_pdb, "w") as fh:
        fh.writelines(to_pdb(protein))
    if contains_sidechains(args.input):
        metric = analysis.get_metric(true_pdb=args.input, pred_pdb=analysis.tmp_pdb)
        print(f"----- Metric: ----- {metric}")
    else:
        print("----- No side chain atoms found in the input PDB. Skipping metric calculation. -----")
    print("----- Finishing evaluation! -----")


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--input", type=str, help="The input pdb file path.", required=True)
    p.add_argument("--outdir", type=str, help="Directory to store outputs.", required=True)
    p.add_argument("--molprobity_clash_loc", type=str, help="Path to /build/bin/molprobity.clashscore.", required=True)
    p.add_argument("--use_proximal", action="store_true", help="Use proximal optimize.")
    p.add_argument("--device", type=str, help="cuda (the MI355X HIP device)", default="cuda")
    p.add_argument("--ckpt_path", type=str, default=None, help="Lightning checkpoint (else $PACKPPI_CKPT, else the config tree's ckpt_path).")
    p.add_argument("--config_dir", type=str, default=None, help="The reference's configs/ directory (else $PACKPPI_CONFIG_DIR): "
                   "encoder / model / sampling YAML files are read from it.")
    p.add_argument("--steps", type=int, default=None, help="Diffusion steps (reference schedule: 30).")
    p.add_argument("--seed", type=int, default=None, help="Seed of the device generator for the initial noise.")
    p.add_argument("--random_weights", type=int, default=None, help="Seeded stand-in weights instead of a checkpoint.")
    args = p.parse_args(argv)
    evaluate_model(load_model(args), args)


if __name__ == "__main__":
    main()
