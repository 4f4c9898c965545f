"""Per-layer GPU parity (a3 / a5 / a7 / a8 of SURVEY.md section 8a): the tensors BETWEEN the launches of one network evaluation
against the reference's own per-layer tensors (fixtures g2_ops_*: hV0_* after the node embedding, hV_l{0,1,2}_t1 and
hE_l{0,1}_t1 after each InvariantPointMessagePassing layer; mpnn.py:47-62, layers.py:119-148,257-268).

The tensors are read through ``pp_debug_score_prefix`` / ``pp_debug_buffer``, which only ``libpackppi_hip.dbg.so`` exports
(-DPP_DIAG: the default kernels, same results): ``test_diag_library_runs_the_layer_tests`` starts one child test run on it.
"""
import ctypes as C
import os
import subprocess
import sys

import pytest
import torch

from .conftest import load_golden

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# pp_score's launch schedule: embed, NM0, NU0, EU0(+NM1), NU1, EU1(+NM2), NU2
AFTER = {"hV0": 1, "hV_l0": 3, "hE_l0": 4, "hV_l1": 5, "hE_l1": 6, "hV_l2": 7}


def _diag():
    from packppi_amd import lib as L
    l = L.load()
    if not hasattr(l, "pp_debug_score_prefix"):
        pytest.skip("needs libpackppi_hip.dbg.so (run through test_diag_library_runs_the_layer_tests)")
    l.pp_debug_score_prefix.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_void_p]
    l.pp_debug_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    return l


def _prefix(l, ctx, chi, t, n):
    assert l.pp_debug_score_prefix(ctx.handle, C.c_void_p(chi.data_ptr()), float(t), n, None) == 0, l.pp_last_error()


def _buffer(l, ctx, which, shape):
    out = torch.empty(*shape, device=DEV)
    assert l.pp_debug_buffer(ctx.handle, which, C.c_void_p(out.data_ptr()), out.numel()) == 0, l.pp_last_error()
    return out.cpu()


@pytest.mark.parametrize("name", ["g2_ops_L8", "g2_ops_L33", "g2_ops_L64", "g2_ops_B3"])
def test_node_embedding_at_three_times(name, weights):
    """a5: the sinusoidal time embedding (arguments up to 1e4 rad at t = 1) + node embedding + LayerNorm, straight out of the
    embedding launch, vs the reference's h_V0 at t = 1, 0.5, 1/30."""
    from packppi_amd.module import TDiffusionModule
    l = _diag()
    b, g = load_golden(name)
    B, L = b.residue_type.shape
    ctx = TDiffusionModule(weights, device=DEV)._context(b.to(DEV))
    chi = g["init_chi_seed7"].to(DEV).contiguous()
    valid = b.residue_mask.bool()
    for tval, tn in ((1.0, "t1"), (0.5, "t05"), (1.0 / 30, "t30")):
        _prefix(l, ctx, chi, tval, 1)
        hV0 = _buffer(l, ctx, 5, (B, L, 128))
        assert (hV0 - g[f"hV0_{tn}"])[valid].abs().max() < 2e-5, tn


@pytest.mark.parametrize("name", ["g2_ops_L8", "g2_ops_L33", "g2_ops_L64", "g2_ops_B3"])
def test_per_layer_states(name, weights):
    """a3 / a7 / a8: h_V after each of the three layers and h_E after the first two (the third edge update is dead code in the
    reference), each read right after the launch that produces it, vs the reference's tensors at t = 1."""
    from packppi_amd.module import TDiffusionModule
    l = _diag()
    b, g = load_golden(name)
    if "hV_l0_t1" not in g:
        pytest.skip("fixture without per-layer tensors")
    B, L = b.residue_type.shape
    K = min(32, L)
    ctx = TDiffusionModule(weights, device=DEV)._context(b.to(DEV))
    chi = g["init_chi_seed7"].to(DEV).contiguous()
    valid = b.residue_mask.bool()
    worst = {}
    for key, n in AFTER.items():
        if key == "hV0" or key + "_t1" not in g:             # (the larger fixtures carry the h_V tensors only)
            continue
        _prefix(l, ctx, chi, 1.0, n)                      # the prefix always starts from the embedding: launches are idempotent
        if key.startswith("hV"):
            got = _buffer(l, ctx, 5, (B, L, 128))
        else:
            got = _buffer(l, ctx, 0, (B, L, K, 128))
        ref = g[key + "_t1"]
        d = (got - ref)[valid].abs().max().item()
        worst[key] = d
        assert d < (5e-5 if key.startswith("hV") else 1e-4), (key, d)
    print(name, {k: f"{v:.1e}" for k, v in worst.items()})


def test_diag_library_runs_the_layer_tests():
    """One child run of this file (and of the tests that force launch shapes) on libpackppi_hip.dbg.so."""
    from packppi_amd.build import diag_variant_path
    if os.environ.get("PACKPPI_LIB"):
        pytest.skip("already a child run")
    lib = diag_variant_path()
    if not os.path.exists(lib):
        pytest.skip("libpackppi_hip.dbg.so not built (__graft_entry__.build() builds it)")
    env = dict(os.environ, PACKPPI_LIB=lib, PACKPPI_EXPECT_VARIANT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_hip_layers.py"),
                        os.path.join(ROOT, "tests", "test_hip_parity.py"), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", "-k",
                        "test_node_embedding_at_three_times or test_per_layer_states or test_residues_per_workgroup_agree or "
                        "test_library_variant_is_the_requested_one"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-500:]
