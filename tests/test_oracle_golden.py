"""Pins the CPU oracle (oracle/ref_cpu.py) to golden vectors produced by the unmodified reference.

Every tolerance here is "same fp32 arithmetic, possibly a different summation order".
"""
import os

import numpy as np
import pytest
import torch

from oracle import ref_cpu as O
from packppi_amd import constants as rc
from .conftest import GOLD, load_golden, wrapped_absdiff

OPS = ["g2_ops_L8", "g2_ops_L33", "g2_ops_L64", "g2_ops_B3"]


def test_dists_bounds_match_reference():
    z = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "g1_dists_bounds.npz"))
    for tol, vtf in ((0.5, 12.0), (1.5, 15.0), (0.1, 12.0)):
        lo, up = rc.make_atom14_dists_bounds(tol, vtf)
        assert np.array_equal(lo, z[f"lower_{tol}_{vtf}"])
        assert np.array_equal(up, z[f"upper_{tol}_{vtf}"])


@pytest.mark.parametrize("name", OPS)
def test_knn_and_edge_embedding(name, weights):
    b, g = load_golden(name)
    E_idx = O.knn_graph(b.X[:, :, 1, :], b.residue_mask)
    valid = b.residue_mask.bool()
    assert torch.equal(E_idx[valid], g["E_idx"][valid])
    E_idx, hE = O.encode_static(weights, b)
    assert (hE - g["hE0"])[valid].abs().max() < 2e-5
    if "E_raw" in g:
        E = O.edge_features(b.X, E_idx, b.residue_index, b.chain_indices)
        assert (E - g["E_raw"])[valid].abs().max() < 1e-6


@pytest.mark.parametrize("name", OPS)
def test_init_noise_contract(name):
    """randn_like draws under torch.manual_seed(s) == our explicit generator draws, in order."""
    b, g = load_golden(name)
    gen = torch.Generator().manual_seed(7)
    B, L = b.residue_type.shape
    t = torch.ones(B * L)
    x = O.add_sc_noise(b, t, O.initial_noise(b, gen))
    assert wrapped_absdiff(x, g["init_chi_seed7"]).max() < 1e-6


@pytest.mark.parametrize("name", OPS)
def test_network(name, weights):
    b, g = load_golden(name)
    B, L = b.residue_type.shape
    valid = b.residue_mask.bool()
    chi = g["init_chi_seed7"]
    for tval, tn in ((1.0, "t1"), (0.5, "t05"), (1.0 / 30, "t30")):
        t = torch.tensor([tval]).repeat_interleave(B * L)
        hV0 = O.encode_nodes(weights, b, chi, t)
        assert (hV0 - g[f"hV0_{tn}"])[valid].abs().max() < 2e-5
        score, hV = O.network(weights, b, chi, t)
        assert (hV - g[f"hV_{tn}"])[valid].abs().max() < 5e-5
        assert (score - g[f"score_{tn}"])[valid].abs().max() < 5e-5


@pytest.mark.parametrize("name", ["g2_ops_L8", "g2_ops_L33"])
def test_layerwise_states(name, weights):
    b, g = load_golden(name)
    B, L = b.residue_type.shape
    E_idx, hE = O.encode_static(weights, b)
    hV = O.encode_nodes(weights, b, g["init_chi_seed7"], torch.ones(B * L))
    R, tr = O.backbone_frames(b.X)
    mask = b.residue_mask
    ma = mask[..., None] * O._gather_nodes(mask[..., None], E_idx)[..., 0]
    for l in range(3):
        hV, hE = O.ipmp_layer(weights, l, hV, hE, E_idx, R, tr, mask, ma)
        assert (hV - g[f"hV_l{l}_t1"]).abs().max() < 5e-5
        if l < 2:
            assert (hE - g[f"hE_l{l}_t1"]).abs().max() < 5e-5


@pytest.mark.parametrize("name", OPS)
def test_atom14_and_clash(name):
    b, g = load_golden(name)
    chi = g["init_chi_seed7"]
    xyz = O.atom14_coords(b.X, b.residue_type, b.BB_D, chi)
    assert (xyz - g["atom14_init"]).abs().max() < 1e-5
    assert torch.equal(xyz[..., :4, :], b.X[..., :4, :])
    xyz_t = O.atom14_coords(b.X, b.residue_type, b.BB_D, b.SC_D)
    assert (xyz_t - g["atom14_true"]).abs().max() < 1e-5
    pr = O.residue_clash(b, chi, 12.0, 0.5)
    assert (pr - g["clash_init"]).abs().max() < 1e-5
    pr = O.residue_clash(b, b.SC_D, 12.0, 0.1)
    assert (pr - g["clash_true_tol01"]).abs().max() < 1e-5
    _, grad = O.clash_and_grad(b, chi, 12.0, 0.5)
    assert (grad - g["clash_grad_init"]).abs().max() < 1e-5 + 1e-4 * g["clash_grad_init"].abs().max()


@pytest.mark.parametrize("name", OPS)
def test_metrics(name):
    b, g = load_golden(name)
    m = O.analyze_samples(b, g["init_chi_seed7"])
    for k, v in m.items():
        assert abs(float(v) - float(g["metric." + k])) < 1e-5 * max(1.0, abs(float(v))), k


@pytest.mark.parametrize("name,steps", [("g3_sampling_L64", (30, 100)), ("g3_sampling_B3", (30,))])
def test_sampling_ode(name, steps, weights):
    b, g = load_golden(name)
    for n in steps:
        chi = O.sampling(weights, b, g["init_chi_seed11"], torch.linspace(1, 0, n + 1))
        d = wrapped_absdiff(chi, g[f"chi_ode_{n}"])[b.SC_D_mask.bool()]
        assert d.max() < 2e-5, (n, float(d.max()))


def test_sampling_unhoisted_equals_hoisted(weights):
    b, g = load_golden("g3_sampling_L64")
    s = torch.linspace(1, 0, 6)
    a = O.sampling(weights, b, g["init_chi_seed11"], s, hoist=True)
    c = O.sampling(weights, b, g["init_chi_seed11"], s, hoist=False)
    assert torch.equal(a, c)


def test_sampling_sde(weights):
    b, g = load_golden("g3_sampling_sde_L33")
    n = 30
    torch.manual_seed(99)
    N = b.residue_type.numel()
    noise = []
    for _ in range(n):
        noise.append((torch.normal(mean=0, std=1, size=(N, 4)), torch.normal(mean=0, std=1, size=(N, 4))))
    chi = O.sampling(weights, b, g["init_chi_seed11"], torch.linspace(1, 0, n + 1), mode="sde", sde_noise=noise)
    d = wrapped_absdiff(chi, g["chi_sde_30_seed99"])[b.SC_D_mask.bool()]
    assert d.max() < 5e-5, float(d.max())


def test_T1124_sde_100_steps(weights):
    """The oracle's SDE path at benchmark size against the reference's run (fixture g9: seed and result only)."""
    b, g = load_golden("g4_T1124")
    ref = torch.from_numpy(np.load(os.path.join(GOLD, "g9_T1124_sde.npz"))["chi_sde_100_seed1124"])
    n = 100
    torch.manual_seed(1124)
    N = b.residue_type.numel()
    noise = [(torch.normal(mean=0, std=1, size=(N, 4)), torch.normal(mean=0, std=1, size=(N, 4))) for _ in range(n)]
    chi = O.sampling(weights, b, g["init_chi_seed1124"], torch.linspace(1, 0, n + 1), mode="sde", sde_noise=noise)
    d = wrapped_absdiff(chi, ref)[b.SC_D_mask.bool()]
    assert d.max() < 5e-5, float(d.max())


@pytest.mark.parametrize("name", ["g3_proximal_L64"])
def test_proximal(name):
    b, g = load_golden(name)
    init = g["init_chi_seed11"]
    for n in (5, 50):
        chis, losses = O.proximal_optimizer(b, init.clone(), 12.0, 0.5, 1.0, n)
        assert np.allclose(np.array(losses), g[f"prox_losses_{n}"].numpy(), rtol=1e-4, atol=1e-6)
        assert wrapped_absdiff(chis[-1], g[f"prox_chi_last_{n}"]).max() < 1e-4
        assert wrapped_absdiff(chis[0], g[f"prox_chi_first_{n}"]).max() < 1e-5
        mask = O.clash_mask(b, init, 12.0, 0.5)
        assert torch.equal(chis[-1][~mask], init[~mask])          # untouched where not clashing


def test_sampling_L300_100_steps(weights):
    b, g = load_golden("g3_sampling_L300")
    chi = O.sampling(weights, b, g["init_chi_seed11"], torch.linspace(1, 0, 101))
    d = wrapped_absdiff(chi, g["chi_ode_100"])[b.SC_D_mask.bool()]
    assert d.max() < 2e-5, float(d.max())


def test_T1124_100_steps(weights):
    """BASELINE config 1/2 input: data/T1124_lig.pdb, 100 steps, reference CPU output."""
    b, g = load_golden("g4_T1124")
    assert b.true_residues() == 738 and int(b.SC_D_mask.sum()) == 1210
    chi = O.sampling(weights, b, g["init_chi_seed1124"], torch.linspace(1, 0, 101))
    d = wrapped_absdiff(chi, g["chi_ode_100"])[b.SC_D_mask.bool()]
    assert d.max() < 2e-5, float(d.max())
    m = O.analyze_samples(b, chi)
    assert abs(float(m["atom_rmsd"]) - float(g["metric.atom_rmsd"])) < 1e-4
    pr = O.residue_clash(b, g["chi_ode_100"], 12.0, 0.5)
    assert (pr - g["clash_final"]).abs().max() < 1e-5


@pytest.mark.parametrize("tag", ["L64", "L120"])
def test_proximal_arbiter_fixture(tag):
    """The oracle against the round-2 proximal fixtures (g6): gradient and clash value at the reference's own iterates, the
    loss curve and the first ten steps of the trajectory; and the fixture's own statement of how far the reference's fp32 run
    ends from its fp64 run (the bound the GPU test uses)."""
    import os
    from .conftest import GOLD
    z = np.load(os.path.join(GOLD, f"g6_prox_{tag}.npz"))
    b, g = load_golden(str(z["source_fixture"]))
    chi0 = g[str(z["chi0_key"])].float()
    for n in (1, 10, 50):
        x = torch.from_numpy(z[f"chi32_step{n}"]).float()
        pr, grad = O.clash_and_grad(b, x, 12.0, 0.5)
        assert (pr - torch.from_numpy(z[f"per_res32_step{n}"])).abs().max() < 2e-5
        assert (grad - torch.from_numpy(z[f"grad32_step{n}"])).abs().max() < 3e-7
    chis, losses = O.proximal_optimizer(b, chi0.clone(), 12.0, 0.5, 1.0, 10)
    assert np.allclose(np.array(losses), z["losses32"][:10], rtol=2e-5)
    for n in (1, 5, 10):
        assert wrapped_absdiff(chis[n - 1], torch.from_numpy(z[f"chi32_step{n}"])).max() < 2e-5
    d = wrapped_absdiff(torch.from_numpy(z["chi32_step50"]), torch.from_numpy(z["chi64_step50"])).max()
    assert d <= 3.0e-3 + 1e-6            # REF_FP32_VS_FP64_WORST in tests/test_hip_parity.py is the L64 value


def test_c5_complex_with_knn_tie(weights):
    """Complex 12 of BASELINE config 4's set has an exact CA-distance tie at rank 32 / 33 (row 8): the oracle, which calls
    torch.topk like the reference, reproduces the reference's 100-step output -- the fixture the GPU test leans on when it
    hands the reference's neighbour lists to pp_ctx_set_graph."""
    import os
    from packppi_amd import synth
    from packppi_amd.featurize import protein_to_batch
    from .conftest import GOLD
    z = np.load(os.path.join(GOLD, "g7_c5_rank0.npz"))
    i = 12
    b = protein_to_batch(synth.make_complex(int(z["lengths"][i]), 10000 + i))
    ca = b.X[0, :, 1, :]
    d = torch.sqrt(((ca[:, None] - ca[None]) ** 2).sum(-1) + 1e-6).sort(dim=-1)[0]
    assert torch.nonzero(d[:, 31] == d[:, 32]).flatten().tolist() == [8]
    with torch.no_grad():
        chi = O.sampling(weights, b, torch.from_numpy(z[f"init_{i}"]), torch.linspace(1, 0, 101))
    assert wrapped_absdiff(chi, torch.from_numpy(z[f"chi_ode_100_{i}"]))[b.SC_D_mask.bool()].max() < 2e-5
