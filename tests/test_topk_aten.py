"""The reference CPU path's neighbour choice on equal distances (encoder.py:105-118: torch.topk(..., largest=False)).

ATen's CPU topk runs libstdc++'s nth_element + sort (or partial_sort) over (value, index) pairs with a value-only
comparator; which of several equal values is returned is a property of those algorithms.  csrc/pp_topk_aten.h restates them
so that the kNN kernel can run them on the device.  Here, on the CPU: the restatement against (a) the real std:: algorithms,
whole permutations, (b) torch.topk itself through the library's host entry point, (c) the reference's own neighbour lists
stored with the config-4 goldens."""
import ctypes
import glob
import os
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def tie_heavy_rows(n_rows, seed):
    """Rows of 1 .. 6000 values drawn from few distinct levels, some pre-sorted either way, a few with NaNs."""
    rng = np.random.default_rng(seed)
    for trial in range(n_rows):
        n = int(rng.integers(1, 3000)) if trial % 3 else int(rng.integers(2048, 6000))
        k = min(32, n) if trial % 5 else int(rng.integers(1, min(n, 64) + 1))
        levels = int(rng.choice([2, 5, 20, 100, 1000, 100000]))
        v = (rng.integers(0, levels, n).astype(np.float32) * np.float32(0.37))
        if trial % 7 == 0:
            v = np.sort(v)
        if trial % 11 == 0:
            v = np.sort(v)[::-1].copy()
        if trial % 13 == 0:
            v[rng.integers(0, n, max(1, n // 50))] = np.nan
        yield np.ascontiguousarray(v), k


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("topk") / "libtopk_check.so")
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "native", "topk_aten_check.cpp")],
                   check=True)
    return ctypes.CDLL(so)


def _call(fn, *args, n_out):
    out = np.zeros(n_out, np.int32)
    fn(*args, out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_restatement_equals_libstdcxx(harness):
    vp = ctypes.c_void_p
    for v, k in tie_heavy_rows(1500, 0):
        p, n = v.ctypes.data_as(vp), len(v)
        assert np.array_equal(_call(harness.mine_topk, p, n, k, n_out=k), _call(harness.std_topk, p, n, k, n_out=k)), (n, k)
        for which in (0, 1, 2, 3):       # 3: nth_element through the data-parallel partition formula the kNN kernel evaluates
            a = _call(harness.mine_piece, which, p, n, k, n_out=n)
            b = _call(harness.std_piece, which, p, n, k, n_out=n)
            assert np.array_equal(a, b), (which, n, k)


@pytest.fixture(scope="module")
def lib():
    from packppi_amd.build import build_library
    from packppi_amd import lib as L
    build_library(verbose=False)
    return L.load()


def host_topk(lib, v, k):
    out = np.zeros(k, np.int32)
    st = lib.pp_topk_aten_host(v.ctypes.data_as(ctypes.c_void_p), len(v), k, out.ctypes.data_as(ctypes.c_void_p))
    assert st == 0
    return out


def test_library_host_entry_equals_torch_topk(lib):
    for v, k in tie_heavy_rows(600, 1):
        want = torch.topk(torch.from_numpy(v)[None, None], k, dim=-1, largest=False)[1][0, 0].numpy()
        assert np.array_equal(host_topk(lib, v, k), want), (len(v), k)
    # many rows at once (ATen splits the rows over threads; every row is independent)
    rng = np.random.default_rng(2)
    D = (rng.integers(0, 40, (300, 300)).astype(np.float32) * np.float32(0.25))
    want = torch.topk(torch.from_numpy(D)[None], 32, dim=-1, largest=False)[1][0].numpy()
    for r in range(300):
        assert np.array_equal(host_topk(lib, np.ascontiguousarray(D[r]), 32), want[r])
    bad = np.zeros(4, np.float32)
    assert lib.pp_topk_aten_host(bad.ctypes.data_as(ctypes.c_void_p), 4, 5, bad.ctypes.data_as(ctypes.c_void_p)) == 1


def test_reference_neighbour_lists_of_config4(lib):
    """The reference's own E_idx (stored for the complexes of BASELINE config 4 whose rows hold equal distances) from the
    reference's distance arithmetic + the restated selection: every row, order included."""
    from packppi_amd import synth
    from packppi_amd.featurize import protein_to_batch
    files = sorted(glob.glob(os.path.join(GOLD, "g7_c5_rank*.npz")))
    if not files:
        pytest.skip("no g7 fixtures")
    lens = synth.c5_lengths(256)
    checked = member = 0
    for f in files:
        z = np.load(f)
        ids = [int(k[6:]) for k in z.files if k.startswith("E_idx_")]
        for i in sorted(ids)[:6]:                       # a few per file keep the CPU suite short
            b = protein_to_batch(synth.make_complex(lens[i], 10000 + i))
            X = b.X[:, :, 1, :]
            m2 = b.residue_mask[:, None, :] * b.residue_mask[:, :, None]
            D = m2 * torch.sqrt(((X[:, None] - X[:, :, None]) ** 2).sum(3) + 1e-6)
            Dadj = (D + 2 * (1. - m2) * D.max(-1, keepdim=True)[0])[0].numpy()
            E = z[f"E_idx_{i}"].astype(np.int64)
            srt = np.sort(Dadj, -1)
            for r in z[f"tie_rows_{i}"].astype(np.int64):
                assert np.array_equal(host_topk(lib, np.ascontiguousarray(Dadj[r]), 32), E[r]), (i, int(r))
                member += int(srt[r, 31] == srt[r, 32])
                checked += 1
    if not checked:
        pytest.skip("fixtures hold no stored neighbour lists")
