"""CPU-side checks of the C-ABI library: it builds, loads and exports exactly what include/packppi_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    from packppi_amd.build import build_library
    return build_library(verbose=False)


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "packppi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from packppi_amd.lib import SYMBOLS
    assert sorted(SYMBOLS) == declared_symbols()


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.pp_version.restype = ctypes.c_int
    assert lib.pp_version() >= 100


def test_plan_create_without_gpu_reports_error(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from packppi_amd.lib import Plan
    with pytest.raises(RuntimeError):
        Plan(None, "cpu")


def test_weight_offsets_match_python_spec(lib_path):
    from packppi_amd.weights import weight_spec
    import numpy as np
    total = sum(int(np.prod(s)) for _, s in weight_spec())
    assert total == 1439172 and len(weight_spec()) == 112
