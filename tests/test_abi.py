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


def _header(diag):
    src = open(os.path.join(ROOT, "include", "packppi_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    block = re.search(r"#ifdef PP_DIAG\n(.*?)#endif", src, flags=re.S)
    return block.group(1) if diag else src.replace(block.group(0), "")


def declared_symbols(diag=False):
    """The product ABI (diag=False) or the extra exports of libpackppi_hip.dbg.so (the header's PP_DIAG block)."""
    return sorted(set(re.findall(r"\b(pp_[a-z0-9_]+)\s*\(", _header(diag))))


def test_header_and_binding_agree():
    from packppi_amd.lib import SYMBOLS
    assert sorted(SYMBOLS) == declared_symbols()


def test_library_exports_every_declared_symbol(lib_path):
    lib = ctypes.CDLL(lib_path)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.pp_version.restype = ctypes.c_int
    assert lib.pp_version() >= 100


def test_product_libraries_carry_no_diagnostics():
    """pp_debug_* exports and the launchers' environment switches exist in libpackppi_hip.dbg.so only (-DPP_DIAG); the
    default, .f32 and .chk libraries export exactly the documented ABI."""
    import subprocess
    from packppi_amd import build
    diag = declared_symbols(diag=True)
    assert diag and all(n.startswith("pp_debug_") for n in diag)

    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("pp_") and " T " in ln}
    for path in (build.LIB, build.other_variant_path(), build.check_variant_path()):
        if os.path.exists(path):
            names = exported(path)
            assert not [n for n in names if n.startswith("pp_debug_")], path
            assert set(declared_symbols()) <= names, path
            assert b"PP_NU_SPLIT" not in open(path, "rb").read() and b"PP_EDGE_R" not in open(path, "rb").read()
    dbg = build.build_diag_variant(verbose=False)
    assert set(diag) <= exported(dbg) and set(declared_symbols()) <= exported(dbg)
    assert b"PP_NU_SPLIT" in open(dbg, "rb").read()


def test_lab_switches_need_the_lab_flag_and_a_tag(monkeypatch):
    """-DPP_X_* timing variants (most give wrong results) do not compile without -DPP_LAB, extra flags never reach a product
    library, and the binding refuses a library whose flag stamp is none of the product sets."""
    import subprocess
    from packppi_amd import build, lib as L
    hdr = os.path.join(build.CSRC, "pp_internal.h")
    base = [build._hipcc(), "--offload-arch=gfx950", "-std=c++17", "-x", "hip", "-E", "--cuda-host-only", hdr, "-o", os.devnull]
    bad = subprocess.run(base + ["-DPP_X_NOWLOAD"], capture_output=True, text=True)
    assert bad.returncode != 0 and "PP_LAB" in bad.stderr
    assert subprocess.run(base + ["-DPP_X_NOWLOAD", "-DPP_LAB"], capture_output=True, text=True).returncode == 0
    assert subprocess.run(base, capture_output=True, text=True).returncode == 0
    with pytest.raises(RuntimeError, match="fixed flag set"):
        build.build_library(extra_flags=["-DPP_LAB"], tag="")
    monkeypatch.setenv("PACKPPI_CFLAGS", "-DPP_LAB -DPP_X_NOSAT")
    with pytest.raises(RuntimeError, match="fixed flag set"):
        build.build_library(tag="chk", extra_flags=["-DPP_CHECK_RANGE"])
    monkeypatch.delenv("PACKPPI_CFLAGS")
    stamps = build.product_flag_stamps()
    assert len(stamps) == 4 and build.embedded_build_id(build.LIB).split("-")[1] in stamps
    monkeypatch.setattr(build, "product_flag_stamps", lambda: {})                # "this library is a laboratory build"
    monkeypatch.setattr(L, "_lib", None)
    with pytest.raises(RuntimeError, match="laboratory variant"):
        L.load()
    monkeypatch.setenv("PACKPPI_ALLOW_LAB_LIBRARY", "1")
    assert L.load() is not None


def test_build_stamp_matches_the_sources_on_disk(lib_path, monkeypatch):
    """pp_build_id() = hash of csrc/*.hip|*.h + the public header, and of the flags, taken when the library was compiled:
    build() rebuilds on a mismatch (content, not mtime) and the binding refuses a stale prebuilt library."""
    from packppi_amd import build, lib as L
    lib = ctypes.CDLL(lib_path)
    lib.pp_build_id.restype = ctypes.c_char_p
    stamp = lib.pp_build_id().decode()
    assert re.fullmatch(r"[0-9a-f]{16}-[0-9a-f]{16}", stamp)
    assert stamp == build.build_id(build.FLAGS, build.SOURCES) == build.embedded_build_id(lib_path)
    assert not build.needs_build()
    for other in (build.other_variant_path(), build.check_variant_path(), build.diag_variant_path()):      # variants: same sources, other flags
        if os.path.exists(other):
            assert build.embedded_build_id(other).split("-")[0] == build.source_hash()
            assert build.embedded_build_id(other) != stamp
    monkeypatch.setattr(build, "source_hash", lambda: "0" * 16)                  # "the sources changed"
    assert build.needs_build()
    monkeypatch.setattr(L, "_lib", None)
    with pytest.raises(RuntimeError, match="stale"):
        L.load()
    monkeypatch.setenv("PACKPPI_SKIP_BUILD_CHECK", "1")
    assert L.load() is not None


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


def test_range_check_is_a_build_variant(lib_path):
    """The default library has no f16 range counters (pp_range_check -> PP_ERR_UNSUPPORTED, no device call); the check
    variant, when built, says so."""
    lib = ctypes.CDLL(lib_path)
    lib.pp_last_error.restype = ctypes.c_char_p
    assert lib.pp_has_range_check() == 0
    ev = ctypes.c_ulonglong(7)
    assert lib.pp_range_check(ctypes.byref(ev), 0) == 3 and ev.value == 0
    assert b"PP_CHECK_RANGE" in lib.pp_last_error()
    from packppi_amd.build import check_variant_path
    if os.path.exists(check_variant_path()):
        assert ctypes.CDLL(check_variant_path()).pp_has_range_check() == 1


def test_plan_create_rejects_weights_outside_the_f16_range(lib_path):
    """Checked before any device call: a non-finite weight or one beyond 65504 cannot be split into f16 operands."""
    import numpy as np
    from packppi_amd.lib import PPTables
    lib = ctypes.CDLL(lib_path)
    lib.pp_last_error.restype = ctypes.c_char_p
    lib.pp_plan_create.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(PPTables), ctypes.c_int,
                                   ctypes.POINTER(ctypes.c_void_p)]
    for bad in (np.inf, np.nan, 7.0e4):
        w = np.zeros(1439172, dtype=np.float32)
        w[12345] = bad
        out = ctypes.c_void_p()
        st = lib.pp_plan_create(w.ctypes.data_as(ctypes.c_void_p), w.size, ctypes.byref(PPTables()), 0, ctypes.byref(out))
        assert st == 1 and b"weight 12345" in lib.pp_last_error() and not out.value
