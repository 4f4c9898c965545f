"""Host-side logic (no GPU): PDB IO, featurisation vs the reference's prot_to_data fixtures, batching, checkpoints."""
import os
import re

import numpy as np
import pytest
import torch

from packppi_amd import constants as rc
from packppi_amd import synth
from packppi_amd.batch import Batch, TENSOR_KEYS, collate, split
from packppi_amd.featurize import protein_to_batch, protein_to_data
from packppi_amd.pdb_io import from_pdb_file, to_pdb
from .conftest import GOLD, ROOT


@pytest.mark.parametrize("tag", ["1BRS", "2FTL"])
def test_featurize_matches_reference_prot_to_data(tag):
    """g0 fixtures: protein dict (from our parser) and the reference's prot_to_data output on it."""
    z = np.load(os.path.join(GOLD, f"g0_protein_{tag}.npz"))
    prot = {k[5:]: z[k] for k in z.files if k.startswith("prot.")}
    d = protein_to_data(prot)
    for k in TENSOR_KEYS:
        ref = torch.from_numpy(z["ref." + k])
        assert torch.equal(d[k], ref), k


def test_pdb_write_read_roundtrip(tmp_path):
    prot = synth.make_complex(40, 3)
    path = tmp_path / "x.pdb"
    path.write_text(to_pdb(prot))
    lines = path.read_text().splitlines()
    assert all(len(ln) == 80 for ln in lines)
    assert lines[0].startswith("MODEL     1") and lines[-1].startswith("END") and lines[-2].startswith("ENDMDL")
    assert sum(ln.startswith("TER") for ln in lines) == 2
    back = from_pdb_file(path)
    assert np.array_equal(back["aaindex"], prot["aaindex"])
    assert np.array_equal(back["residue_index"], prot["residue_index"])
    assert list(back["chain_id"]) == list(prot["chain_id"])
    assert np.array_equal(back["atom_mask"], prot["atom_mask"])
    m = prot["atom_mask"] > 0
    assert np.abs(back["atom_positions"][m] - prot["atom_positions"][m]).max() <= 5.1e-4      # 3 decimals


def test_pdb_reader_altloc_hydrogens_nonstandard(tmp_path):
    txt = "\n".join([
        "ATOM      1  N   ALA A   1       0.000   0.000   0.000  1.00 10.00           N",
        "ATOM      2  CA AALA A   1       1.000   0.000   0.000  0.40 10.00           C",
        "ATOM      3  CA BALA A   1       1.500   0.000   0.000  0.60 10.00           C",
        "ATOM      4  C   ALA A   1       2.000   1.000   0.000  1.00 10.00           C",
        "ATOM      5  O   ALA A   1       2.000   2.000   0.000  1.00 10.00           O",
        "ATOM      6  H   ALA A   1       0.000   1.000   0.000  1.00 10.00           H",
        "HETATM    7  O   HOH A   2       9.000   9.000   9.000  1.00 10.00           O",
        "ATOM      8  N   XYZ A   3       5.000   0.000   0.000  1.00 10.00           N",
        "ATOM      9  N   GLY B   1       5.000   5.000   0.000  1.00 10.00           N",
        "ATOM     10  N   GLY B   1A      6.000   5.000   0.000  1.00 10.00           N",
    ]) + "\n"
    p = tmp_path / "t.pdb"
    p.write_text(txt)
    prot = from_pdb_file(p)
    assert list(prot["chain_id"]) == ["A", "B", "B"]
    assert prot["aaindex"].tolist() == [0, 7, 7]
    assert prot["atom_positions"][0, 1, 0] == pytest.approx(1.5)          # altloc B has the higher occupancy
    assert prot["atom_mask"][0].tolist()[:5] == [1, 1, 1, 1, 0]           # H dropped, CB absent
    assert prot["residue_index"].tolist() == [1, 1, 2]                    # insertion code shifts by one


def test_collate_and_split():
    ds = [protein_to_data(synth.make_complex(n, 50 + n)) for n in (12, 20, 16)]
    b = collate(ds)
    assert b.num_proteins == 3 and b.max_size == 20 and b.X.shape == (3, 20, 14, 3)
    assert b.residue_mask.sum(1).tolist() == [12, 20, 16]
    assert torch.equal(b.SC_D[0, :12], ds[0].SC_D) and float(b.X[0, 12:].abs().sum()) == 0.0
    parts = split(b)
    assert len(parts) == 3 and parts[1].num_proteins == 1 and torch.equal(parts[1].SC_D[0], ds[1].SC_D)
    assert b.true_residues() == 48


def test_sidechain_roundtrip_through_oracle_atom14():
    """chi -> atom14 -> chi (SURVEY §4 self-consistency): the oracle builder inverts featurize's dihedrals."""
    from oracle import ref_cpu as O
    b = protein_to_batch(synth.make_complex(64, 11))
    xyz = O.atom14_coords(b.X, b.residue_type, b.BB_D, b.SC_D)
    m = b.atom_mask.bool()
    # not exact: chi is re-measured against the actual N position, the builder's chi1 frame uses the literature one
    assert (xyz - b.X)[m].abs().max() < 0.15
    from packppi_amd.featurize import sidechain_dihedrals
    chi, cm = sidechain_dihedrals(xyz[0], b.residue_type[0])
    d = (chi - b.SC_D[0]).abs()
    d = torch.minimum(d, 2 * np.pi - d)
    assert d[b.SC_D_mask[0].bool()].max() < 0.05


def test_synth_is_deterministic_and_compact():
    a, c = synth.make_complex(120, 9), synth.make_complex(120, 9)
    assert np.array_equal(a["atom_positions"], c["atom_positions"], equal_nan=True)
    ca = a["atom_positions"][:, 1]
    d = np.linalg.norm(ca[:, None] - ca[None], axis=-1)
    assert ((d < 10).sum(1) - 1).mean() > 8
    assert len(synth.c5_lengths()) == 256 and min(synth.c5_lengths()) >= 270 and max(synth.c5_lengths()) <= 330


class _FakeDictConfig(dict):
    """Stands in for omegaconf.DictConfig inside a pickled checkpoint; made un-importable before loading."""


def test_lightning_checkpoint_ingest(tmp_path, weights):
    """A Lightning-style .ckpt whose hyper_parameters pickle classes this image does not have."""
    import sys
    import types
    from packppi_amd.module import read_checkpoint_state_dict
    fake = types.ModuleType("omegaconf_fake_pkg")
    fake.DictConfig = _FakeDictConfig
    old_mod, old_name = _FakeDictConfig.__module__, _FakeDictConfig.__qualname__
    _FakeDictConfig.__module__, _FakeDictConfig.__qualname__ = "omegaconf_fake_pkg", "DictConfig"
    _FakeDictConfig.__name__ = "DictConfig"
    sys.modules["omegaconf_fake_pkg"] = fake
    try:
        torch.save({"state_dict": dict(weights), "hyper_parameters": {"encoder_cfg": _FakeDictConfig(a=1)},
                    "epoch": 3}, tmp_path / "m.ckpt")
    finally:
        del sys.modules["omegaconf_fake_pkg"]
        _FakeDictConfig.__module__, _FakeDictConfig.__qualname__ = old_mod, old_name
    sd = read_checkpoint_state_dict(tmp_path / "m.ckpt")
    assert set(sd) == set(weights) and all(torch.equal(sd[k], weights[k]) for k in weights)


def _write_config_tree(root, **over):
    """A directory laid out like the reference's configs/ with the four hot-path files (values of the reference's own YAMLs,
    overridden per test)."""
    import yaml
    enc = dict(node_in=35, edge_in=468, node_features=128, edge_features=128, time_embedding_type="sinusoidal",
               time_embedding_dim=16, num_positional_embeddings=16, num_rbf=16, top_k=32, af2_relpos=True)
    mdl = dict(hidden_dim=128, num_mpnn_layers=3, n_points=8, dropout=0.1, act="relu", position_scale=1.0, use_ipmp=True,
               k_neighbors=32)
    smp = dict(eval_epochs=1, sample_during_training=True, annealed_temp=3, mode="ode", use_proximal=True,
               violation_tolerance_factor=12., clash_overlap_tolerance=0.5, lamda=1., num_steps=50)
    top = {"task_name": "eval", "tags": ["dev"], "seed": 42, "ckpt_path": "/path/to/PackPPI_pretrain_last.ckpt",
           "defaults": ["_self_", {"model": "TorsionalDiffusion.yaml"}], "note": "${paths.root_dir}/x"}
    for d, name in ((enc, "encoder_cfg"), (mdl, "model_cfg"), (smp, "sample_cfg"), (top, "top")):
        d.update(over.get(name, {}))
    for rel, d in (("model/encoder_cfg/ProteinEncoder.yaml", enc), ("model/model_cfg/MpnnNet.yaml", mdl),
                   ("model/sample_cfg/Sampling.yaml", smp), ("eval_diffusion.yaml", top)):
        f = root / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(yaml.safe_dump(d))
    return root


def test_hot_path_configs_are_read_and_checked(tmp_path, weights):
    """configs/{eval_diffusion, model/encoder_cfg/ProteinEncoder, model/model_cfg/MpnnNet, model/sample_cfg/Sampling}.yaml as
    plain YAML (eval_diffusion.py:22-41 composes them with Hydra): sample_cfg reaches the module, dimensions other than the
    compiled ones raise a RuntimeError that names the key -- before anything touches a device."""
    from packppi_amd import config
    from packppi_amd.module import TDiffusionModule
    cfgs = config.load_hot_path_configs(_write_config_tree(tmp_path / "a", sample_cfg=dict(annealed_temp=2.5, num_steps=7)))
    assert cfgs.sample_cfg["annealed_temp"] == 2.5 and cfgs.sample_cfg["num_steps"] == 7 and cfgs.encoder_cfg["top_k"] == 32
    assert cfgs.ckpt_path is None and cfgs.seed == 42             # the shipped placeholder path is not a checkpoint
    config.check_compiled_dims(cfgs.encoder_cfg, cfgs.model_cfg)
    assert config.resolve_ckpt("x.ckpt", cfgs) == "x.ckpt"
    real = tmp_path / "m.ckpt"
    c2 = config.load_hot_path_configs(_write_config_tree(tmp_path / "b", top=dict(ckpt_path=str(real))))
    assert c2.ckpt_path == str(real) and config.resolve_ckpt(None, c2) == str(real)
    for name, key, val in (("model_cfg", "hidden_dim", 256), ("encoder_cfg", "top_k", 48), ("model_cfg", "n_points", 4),
                           ("encoder_cfg", "num_rbf", 8), ("model_cfg", "num_mpnn_layers", 4), ("encoder_cfg", "af2_relpos", False),
                           ("model_cfg", "act", "gelu"), ("encoder_cfg", "time_embedding_type", "fourier")):
        bad = config.load_hot_path_configs(_write_config_tree(tmp_path / f"bad_{key}", **{name: {key: val}}))
        with pytest.raises(RuntimeError, match=rf"{name}\.{key} = "):
            config.check_compiled_dims(bad.encoder_cfg, bad.model_cfg)
        with pytest.raises(RuntimeError, match=key):                 # the module refuses it as well, before any device call
            TDiffusionModule(weights, encoder_cfg=bad.encoder_cfg, model_cfg=bad.model_cfg, device="cuda")
    config.check_compiled_dims({"num_positional_embeddings": 99}, {"dropout": 0.5, "k_neighbors": 7})       # not read by the path
    with pytest.raises(RuntimeError, match="does not exist"):
        config.load_hot_path_configs(tmp_path / "nowhere")
    with pytest.raises(RuntimeError, match="annealed_temp"):         # not a number at all; 0 and null are the reference's "no annealing"
        TDiffusionModule(weights, sample_cfg={"annealed_temp": float("nan")}, device="cuda")
    ref = "/root/reference/configs"
    if os.path.isdir(ref):                                           # the reference's own tree (absent on the GPU box)
        r = config.load_hot_path_configs(ref)
        config.check_compiled_dims(r.encoder_cfg, r.model_cfg)
        assert r.sample_cfg["annealed_temp"] == 3 and r.sample_cfg["mode"] == "ode" and r.sample_cfg["num_steps"] == 50
        assert r.ckpt_path is None


def test_rebalanced_relu_chains_are_the_same_network(weights):
    """pp_plan_create (split-f16 build) rescales the ReLU chains of the edge-level MLPs by powers of two so that hidden
    activations are O(1) whatever the checkpoint's split of scale between consecutive layers (csrc/pp_api.hip
    rebalance_relu_chains).  Host-side statement of that, through the CPU oracle: the rebalanced weights are the SAME network
    (every scale a power of two: identical in fp32 up to the order of roundings), the seeded fixtures are left alone, and a
    checkpoint with a layer a thousand times too small / a second layer a thousand times too large is brought back."""
    from oracle import ref_cpu as O
    from packppi_amd import synth
    from packppi_amd.featurize import protein_to_batch
    from packppi_amd.lib import rebalanced_state_dict
    from tools.oracle.envelope_weights import envelope_variants, tiny_operand_variants
    same, n0 = rebalanced_state_dict(weights)
    assert n0 == 0 and all(torch.equal(same[k], weights[k]) for k in weights)
    b = protein_to_batch(synth.make_complex(48, 5))
    g = torch.Generator().manual_seed(4)
    chi = (torch.rand(1, 48, 4, generator=g) * 2 - 1) * 3.0 * b.SC_D_mask
    t = torch.full((48,), 0.4)
    variants = dict(tiny_operand_variants(weights))
    variants.update({k: v for k, v in envelope_variants(weights).items() if k in ("linear x1/32", "trained-like")})
    scaled = dict(weights)
    scaled["mpnn.mpnn_layers.1.edge_dense.W_in.weight"] = weights["mpnn.mpnn_layers.1.edge_dense.W_in.weight"] * 3e5
    variants["edge FFN first layer x3e5"] = scaled
    for name, sd in variants.items():
        reb, n = rebalanced_state_dict(sd)
        assert n > 0, name
        for k in sd:                                   # every tensor times ONE power of two (or untouched)
            ratio = (reb[k] / sd[k])[sd[k] != 0]
            r0 = float(ratio.flatten()[0])
            assert torch.all(ratio == r0) and abs(np.log2(r0) - round(np.log2(r0))) < 1e-12, (name, k, r0)
            assert float(reb[k].abs().max()) < 65504.0
        with torch.no_grad():
            s_a, h_a = O.network(sd, b, chi, t)
            s_b, h_b = O.network(reb, b, chi, t)
        assert float((h_a - h_b).abs().max() / h_a.abs().max()) < 2e-6, name
        assert float((s_a - s_b).abs().max() / s_a.abs().max()) < 2e-6, name
    # the tiny-operand variant comes back to the balanced network it was made from (its scales undo the 1/1024, x1024)
    reb, n = rebalanced_state_dict(tiny_operand_variants(weights)["edge FFN, layer 1"])
    k = "mpnn.mpnn_layers.1.edge_dense.W_in.weight"
    assert 0.25 <= float(reb[k].abs().max() / weights[k].abs().max()) <= 4.0


def test_layernorm_operand_scales(weights):
    """The power-of-two operand scales behind small LayerNorm gains (csrc/pp_rebalance.h ln_operand_scales): ones for the seeded
    weights (they keep their bits and their kernels); for gains / biases of 1e-3 .. 1e-2 every scaled operand feature lands within
    [1/16, 16] of unit size, the scale is an exact power of two and the consuming weight column divided by it stays in range."""
    from packppi_amd import lib as L
    from tools.oracle.envelope_weights import small_ln_gain_variants
    if L.load().pp_edge_variant() != 1:
        pytest.skip("exact-fp32 edge kernels are built: no operand scales")
    s0, n0 = L.ln_operand_scales(weights)
    assert n0 == 0 and bool((s0 == 1).all())
    sites = ["encoder.norm_edges", "mpnn.mpnn_layers.0.norm.3", "mpnn.mpnn_layers.1.norm.3", "mpnn.mpnn_layers.0.norm.2", "mpnn.mpnn_layers.1.norm.2"]
    for name, sd in small_ln_gain_variants(weights).items():
        sc, n = L.ln_operand_scales(sd)
        assert n > (500 if "bias too" in name else 100), (name, n)       # ("bias kept": most features are bias-sized, i.e. already O(1))
        assert bool((torch.log2(sc) == torch.log2(sc).round()).all())                       # exact powers of two
        for row, site in enumerate(sites):
            mag = torch.sqrt(sd[site + ".weight"] ** 2 + sd[site + ".bias"] ** 2) * sc[row]
            assert float(mag.min()) > 1 / 16 and float(mag.max()) < 16, (name, site, float(mag.min()), float(mag.max()))
        w = sd["mpnn.mpnn_layers.1.edge_dense.W_in.weight"] / sc[4][None, :]
        assert float(w.abs().max()) < 32768
    # one LayerNorm with a HUGE gain is scaled the other way
    big = dict(weights)
    big["mpnn.mpnn_layers.0.norm.2.weight"] = weights["mpnn.mpnn_layers.0.norm.2.weight"] * 300.0
    sc, n = L.ln_operand_scales(big)
    assert n == 128 and float(sc[3].max()) < 1.0 and bool((sc[[0, 1, 2, 4]] == 1).all())


def test_rangecheck_names_small_layernorm_gains(weights):
    """What the plan-time rebalancing does not cover (a LayerNorm gain is not an exact reparametrisation): gains with a median
    below 2^-4 in front of edge-kernel operands are named by the range-check tool."""
    from packppi_amd.rangecheck import small_gain_layernorms
    assert small_gain_layernorms(weights) == []
    sd = dict(weights)
    sd["mpnn.mpnn_layers.1.norm.3.weight"] = weights["mpnn.mpnn_layers.1.norm.3.weight"] * 0.01
    sd["mpnn.mpnn_layers.1.norm.0.weight"] = weights["mpnn.mpnn_layers.1.norm.0.weight"] * 0.01      # a node LayerNorm: scaled lo', fine
    found = small_gain_layernorms(sd)
    assert [n for n, _ in found] == ["mpnn.mpnn_layers.1.norm.3.weight"] and found[0][1] < 0.02


def test_check_state_dict_rejects_bad_shapes(weights):
    from packppi_amd.weights import check_state_dict
    bad = dict(weights)
    bad["encoder.node_embedding.weight"] = torch.zeros(128, 35)
    with pytest.raises(RuntimeError):
        check_state_dict(bad)
    bad = dict(weights)
    del bad["decoder_score.2.W_out.bias"]
    with pytest.raises(RuntimeError):
        check_state_dict(bad)
    extra = dict(weights)
    extra["train_loss.mean_value"] = torch.zeros(())
    assert len(check_state_dict(extra)) == 112          # strict=False tolerates extra keys


def test_product_code_never_touches_the_oracle():
    """The oracle is a checker: nothing under packppi_amd/ may import it (and there is no CPU fallback)."""
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    for dp, _, fs in os.walk(os.path.join(ROOT, "packppi_amd")):
        for f in fs:
            if f.endswith(".py"):
                assert not pat.search(open(os.path.join(dp, f)).read()), f


def test_module_refuses_cpu_device(weights):
    from packppi_amd.module import TDiffusionModule
    with pytest.raises(RuntimeError):
        TDiffusionModule(weights, device="cpu")


def test_cli_flags_match_reference():
    import argparse
    from packppi_amd.cli import eval_diffusion, proximal_optimize
    for mod, req in ((eval_diffusion, ["--input", "--outdir", "--molprobity_clash_loc", "--use_proximal", "--device"]),
                     (proximal_optimize, ["--input", "--outdir", "--molprobity_clash_loc",
                                          "--violation_tolerance_factor", "--clash_overlap_tolerance", "--lamda",
                                          "--num_steps"])):
        src = open(mod.__file__).read()
        for flag in req:
            assert f'"{flag}"' in src, (mod.__name__, flag)
    with pytest.raises(SystemExit):
        eval_diffusion.main(["--outdir", "x"])


def test_T1124_pdb_parses_to_the_golden_batch():
    """data/T1124_lig.pdb itself (11 208 ATOM records, 5 527 hydrogens, 1 313 alternate-location flags, 132 HETATM) through
    pdb_io + featurize == the batch the reference's prot_to_data produced from it (tests/golden/g4_T1124.npz, asserted
    equal at generation time by tools/oracle/make_golden.py).  The file lives in the reference checkout: skipped where that
    is absent (the GPU box)."""
    path = "/root/reference/data/T1124_lig.pdb"
    if not os.path.exists(path):
        pytest.skip("reference checkout not present")
    from packppi_amd.batch import TENSOR_KEYS
    from packppi_amd.featurize import protein_to_batch
    from packppi_amd.pdb_io import from_pdb_file
    prot = from_pdb_file(path)
    z = np.load(os.path.join(GOLD, "g4_T1124.npz"))
    assert prot["atom_positions"].shape == (739, 14, 3)
    assert "".join(sorted(set(prot["chain_id"].tolist()))) == "AB"
    assert np.array_equal(prot["chain_id"], z["prot.chain_id"]) and np.array_equal(prot["residue_index"], z["prot.residue_index"])
    b = protein_to_batch(prot)
    for k in TENSOR_KEYS:
        assert np.array_equal(b[k].numpy(), z["batch." + k]), k
    assert int(b.max_size) == 739 and int(b.residue_mask.sum()) == 738 and int(b.SC_D_mask.sum()) == 1210     # (1 226 by composition; 16 chi lack an atom)


def test_pack_unpack_roundtrip():
    """batch.pack: complexes back to back without padding rows; unpack splits per-row results again."""
    from packppi_amd import synth
    from packppi_amd.batch import TENSOR_KEYS, collate, pack, split, unpack
    from packppi_amd.featurize import protein_to_batch, protein_to_data
    ds = [protein_to_data(synth.make_complex(n, 3 + n)) for n in (40, 33, 57)]
    pb = pack(ds)
    assert pb.seg_offsets.tolist() == [0, 40, 73, 130] and pb.num_proteins == 1 and pb.max_size == 130
    for k in TENSOR_KEYS:
        parts = unpack(pb, pb[k])
        for d, part in zip(ds, parts):
            assert torch.equal(part[0], d[k])
    # B = 1 batches and padded batches (via split) pack to the same thing: padding rows are dropped
    assert torch.equal(pack([protein_to_batch(synth.make_complex(n, 3 + n)) for n in (40, 33, 57)]).X, pb.X)
    assert torch.equal(pack(split(collate(ds))).X, pb.X)
    with pytest.raises(ValueError):
        pack([collate(ds)])
    assert pb.seg_offsets_host == [0, 40, 73, 130]
    # a residue masked out in the middle of a chain (missing backbone atom: featurize.py) keeps its row and its mask;
    # only trailing padding is dropped
    holes = [protein_to_batch(synth.make_complex(n, 3 + n)) for n in (40, 33)]
    holes[0].residue_mask[0, 11] = 0.0
    padded = split(collate([protein_to_data(synth.make_complex(n, 3 + n)) for n in (40, 33)]))
    padded[0].residue_mask[0, 11] = 0.0
    for src in (holes, padded):
        ph = pack(src)
        assert ph.seg_offsets_host == [0, 40, 73] and ph.residue_mask[0, 11] == 0 and ph.residue_mask.sum() == 72
        assert torch.equal(ph.X, pack(ds[:2]).X)


def test_add_sc_noise_matches_reference_draw():
    """TDiffusionModule.add_sc_noise (the product function, torch glue) == the reference's own draw under the same CPU
    seed (TorsionalDiffusion.py:111-124, schedule.py:176-196): two randn_like draws, 1pi mask first."""
    import types
    from packppi_amd.module import TDiffusionModule
    from .conftest import load_golden
    for name in ("g2_ops_L8", "g2_ops_L33", "g2_ops_L64", "g2_ops_B3"):
        b, g = load_golden(name)
        B, L = b.residue_type.shape
        stub = types.SimpleNamespace(_t_to_sigma=TDiffusionModule._t_to_sigma)
        torch.manual_seed(7)
        x, score = TDiffusionModule.add_sc_noise.__wrapped__(stub, b, torch.ones(B * L))
        assert torch.equal(x, g["init_chi_seed7"]) and score.shape == x.shape


def _g0_prot(tag):
    z = np.load(os.path.join(GOLD, f"g0_protein_{tag}.npz"))
    return {k[5:]: z[k] for k in z.files if k.startswith("prot.")}


def test_to_pdb_is_byte_exact():
    """The writer against the reference's own output (src/utils/protein.py:207-314; tools/oracle/make_golden_io.py)."""
    import hashlib
    z = np.load(os.path.join(GOLD, "g8_io.npz"))
    for tag in ("1BRS", "2FTL"):
        text = to_pdb(_g0_prot(tag))
        assert hashlib.sha256(text.encode()).hexdigest() == str(z[f"sha256.{tag}"]), tag
    assert to_pdb(_g0_prot("1BRS")).encode() == z["text.1BRS"].tobytes()
    # keep_chains (protein.py:241-248)
    only_a = to_pdb(_g0_prot("1BRS"), keep_chains=["A"])
    assert {ln[21] for ln in only_a.splitlines() if ln.startswith("ATOM")} == {"A"}


def test_interface_mask_matches_reference(tmp_path):
    """ProteinAnalysis.get_prot's interface mask on a written complex == the reference's own get_prot (helper.py:104-128 fed by
    the brute-force form of interface.py:11-56's 10 A residue search; fixture tools/oracle/make_golden_io.py) -- including
    the reference's quirk that chains after the first are compared under shifted residue numbers (see analysis.get_prot);
    and the residue search itself (KD-tree) against its definition."""
    from packppi_amd.analysis import ProteinAnalysis, interface_residues
    z = np.load(os.path.join(GOLD, "g8_io.npz"))
    prot = _g0_prot("1BRS")
    pdb = tmp_path / "true.pdb"
    pdb.write_text(to_pdb(prot))
    data = ProteinAnalysis(None, str(tmp_path / "w"), device="cpu").get_prot(str(pdb), get_interface=True)
    assert torch.equal(data.interface_mask, torch.from_numpy(z["interface_mask"]))
    assert 0 < int(data.interface_mask.sum()) < data.interface_mask.numel()
    # the search: residues of either chain with an atom within 10 A of the other chain, by brute force
    xyz = prot["atom_positions"][prot["atom_mask"] > 0.5]
    owner = np.repeat(np.arange(len(prot["aaindex"])), (prot["atom_mask"] > 0.5).sum(1))
    xyz = np.round(xyz.astype(np.float64), 3)                                        # the file holds 3 decimals
    d = np.linalg.norm(xyz[:, None] - xyz[None], axis=-1)
    chain_of = prot["chain_id"][owner]
    hit = ((d <= 10.0) & (chain_of[:, None] != chain_of[None])).any(1)
    want = {c: sorted(set(prot["residue_index"][owner[hit & (chain_of == c)]].tolist())) for c in np.unique(prot["chain_id"])}
    assert interface_residues(str(pdb)) == want


def test_batch_key_accepts_inference_tensors():
    """Lightning's test / predict loops build their batches under torch.inference_mode(): such tensors have no version counter
    (reading ``_version`` raises).  lib.BatchKey must take them, and -- since an in-place edit of one cannot be seen -- must never
    report a match, so that the context is rebuilt per call (advisor, round 4)."""
    from packppi_amd import synth
    from packppi_amd.featurize import protein_to_batch
    from packppi_amd.lib import BatchKey
    p = synth.make_complex(40, 3)
    with torch.inference_mode():
        b = protein_to_batch(p)
        assert b.X.is_inference()
        key = BatchKey(b)                       # used to raise "Inference tensors do not track version counter"
        assert not key.matches(b)
    b2 = protein_to_batch(p)
    k2 = BatchKey(b2)
    assert k2.matches(b2)
    b2.X[0, 3] += 1.0
    assert not k2.matches(b2)


def test_oracle_annealed_temp_falsy_means_weight_one():
    """schedule.py:216-217: ``annealed_temp`` 0 / None switch the annealed weight off (w = 1)."""
    from oracle import ref_cpu as O
    x = torch.zeros(1, 3, 4)
    sc = torch.ones(1, 3, 4)
    mask = torch.ones(1, 3, 4, dtype=torch.bool)
    old = O.ANNEALED_TEMP
    try:
        O.ANNEALED_TEMP = 0
        a = O.reverse_step(x, sc, torch.tensor(0.5), torch.tensor(0.1), mask)
    finally:
        O.ANNEALED_TEMP = old
    if True:
        sigma = np.exp(np.log(O.SIGMA_MIN) + (np.log(O.SIGMA_MAX) - np.log(O.SIGMA_MIN)) * 0.5)
        g = sigma * np.sqrt(2 * np.log(O.SIGMA_MAX / O.SIGMA_MIN))
        assert torch.allclose(a, torch.full_like(a, float(0.5 * g ** 2 * 0.1)), rtol=1e-6)


def test_bench_summary_is_small_and_last():
    """bench.py's `summary` digest: flat, under 1.5 KB even with every secondary entry and regime present."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    prox = {"k_clash": {"kernel_us": 16.21234, "frac": 0.0123456, "culled_fraction_of_residue_pairs": 0.98765, "hbm_frac": 0.0051234}}
    out = {"value": 60967.123456, "ms_per_step": 12.105123, "scaling": "weak", "n_gpus": 1,
           "config": {"workload": "data/T1124_lig.pdb"}, "parity": {"max_abs_dchi_vs_reference_rad": 4.812345e-6},
           "secondary": [{"config": c, "value": 53812.123, "ms_per_step": 13.7123, "max_abs_dchi_vs_reference_rad": 4.8e-6,
                          "proximal": {"end_state_vs_reference_fp32_rad": 3.4e-3} if "prox" in c or c in ("configs[2]", "configs[3]") else False}
                         for c in ("configs[2]", "S1500", "configs[3]", "configs[4] share", "configs[4] whole, one GPU",
                                   "configs[1], exact-fp32 library")],
           "roofline": {"regimes": {k: {"frac": 0.151234, "mfma_busy": 0.391234, "l2_over_algorithmic": 22.8123} for k in ("t1124", "s1500", "c5")},
                        "proximal": {"T1124": prox, "S1500": prox}},
           "cpu_baseline": {"value": 24.991234, "cores": 16, "kind": "port"}}
    s = bench.make_summary(out)
    assert len(json.dumps(s)) < 1500, len(json.dumps(s))
    assert s["cfg"]["c1_t1124"][0] == 60970.0 and "c4_whole_1gpu" in s["cfg"] and set(s["roof"]) == {"t1124", "s1500", "c5"}
