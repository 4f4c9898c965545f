"""Compile the HIP sources into ``csrc/libpackppi_hip.so`` (gfx950 only, in-tree)."""
import hashlib
import os
import re
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libpackppi_hip.so")
# The edge kernels exist twice: pp_edge_f16.hip (default: split-f16 MFMA, fp32-equivalent accuracy, one workgroup per CU)
# and pp_edge.hip (PACKPPI_EDGE=f32: exact-fp32 MFMA, three workgroups per CU).  Same launchers, same results to ~1e-6.
EDGE_F16 = os.environ.get("PACKPPI_EDGE", "f16") != "f32"
SOURCES = ["pp_api.hip", "pp_prepare.hip", "pp_node.hip", "pp_edge_f16.hip" if EDGE_F16 else "pp_edge.hip", "pp_clash.hip"]
# The flags of the PRODUCT libraries are fixed here: PACKPPI_CFLAGS / -D arguments only reach TAGGED variant libraries
# (python -m packppi_amd.build --tag NAME -DPP_LAB -DPP_X_...), which lib.load() refuses unless PACKPPI_ALLOW_LAB_LIBRARY=1.
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-variable",
         "-Wno-unused-but-set-variable"] + (["-DPP_EDGE_F16"] if EDGE_F16 else [])
PRODUCT_TAGS = ("", "f32", "f16", "chk", "dbg")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _dep_files():
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc")))
    deps.append(os.path.normpath(os.path.join(CSRC, "..", "..", "include", "packppi_hip.h")))
    return deps


def source_hash():
    """16 hex digits over the name and content of every kernel source, internal header and the public header."""
    h = hashlib.sha256()
    for d in _dep_files():
        h.update(os.path.basename(d).encode() + b"\0")
        h.update(open(d, "rb").read() + b"\0")
    return h.hexdigest()[:16]


def flags_hash(flags, sources):
    return hashlib.sha256("\0".join(list(flags) + ["|"] + list(sources)).encode()).hexdigest()[:16]


def build_id(flags, sources):
    """What ``pp_build_id()`` of a library compiled now from these sources with these flags returns."""
    return f"{source_hash()}-{flags_hash(flags, sources)}"


def embedded_build_id(path):
    """The stamp inside a built library, read from the file (no dlopen: a stale library must not get loaded by the check)."""
    if not os.path.exists(path):
        return None
    m = re.search(rb"PP_BUILD_ID=([0-9a-f]{16}-[0-9a-f]{16})", open(path, "rb").read())
    return m.group(1).decode() if m else None


def needs_build(lib_path=LIB, flags=None, sources=None):
    """True unless the library exists and carries the stamp of the sources on disk and of these flags (content hashes, not
    mtimes: a prebuilt library that no longer matches csrc/ is rebuilt)."""
    return embedded_build_id(lib_path) != build_id(FLAGS if flags is None else flags, SOURCES if sources is None else sources)


def _compile_and_link(lib_path, sources, flags, tag, verbose, only=()):
    hipcc = _hipcc()
    stamp = build_id(flags, sources)
    objs, jobs = [], []
    for src in sources:
        obj = os.path.join(CSRC, src.replace(".hip", f".{tag}.o" if tag else ".o"))
        # pp_api.hip carries the build stamp (flags included): a tagged build always recompiles it, so that a laboratory library
        # can never link the product object and pass lib.load()'s flag-stamp check
        if tag and only and src not in only and src != "pp_api.hip":
            objs.append(os.path.join(CSRC, src.replace(".hip", ".o")))
            continue
        extra = [f'-DPP_BUILD_ID="{stamp}"'] if src == "pp_api.hip" else []
        jobs.append([hipcc, *flags, *extra, "-c", os.path.join(CSRC, src), "-o", obj])
        objs.append(obj)
    # the translation units are independent: compile them side by side (the edge kernels alone take minutes)
    workers = max(1, min(len(jobs), int(os.environ.get("PACKPPI_BUILD_JOBS", "0")) or min(5, os.cpu_count() or 1)))

    def _run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    if workers > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(_run, jobs))
    else:
        for cmd in jobs:
            _run(cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return lib_path


def build_library(force=False, verbose=True, extra_flags=(), tag=""):
    """Build the library; with ``tag`` a variant ``libpackppi_hip.<tag>.so`` (selected at run time by PACKPPI_LIB).  Extra
    flags (and PACKPPI_CFLAGS) are for tagged laboratory variants only: the product libraries have fixed flag sets."""
    lib_path = LIB if not tag else LIB.replace(".so", f".{tag}.so")
    env_flags = os.environ.get("PACKPPI_CFLAGS", "").split()
    if tag in PRODUCT_TAGS:
        allowed = {"chk": ["-DPP_CHECK_RANGE"], "dbg": ["-DPP_DIAG"]}.get(tag, [])
        if env_flags or list(extra_flags) != allowed:
            raise RuntimeError(f"libpackppi_hip{'.' + tag if tag else ''}.so is a product library with a fixed flag set; extra flags "
                               f"({' '.join([*extra_flags, *env_flags])}) need their own --tag NAME")
    flags = [*FLAGS, *extra_flags, *env_flags]
    if not force and not needs_build(lib_path, flags, SOURCES):
        return lib_path
    only = os.environ.get("PACKPPI_VARIANT_SOURCES", "").split()      # tagged build: recompile only these, reuse the base objects
    return _compile_and_link(lib_path, SOURCES, flags, tag, verbose, only)


def other_variant_path():
    return LIB.replace(".so", ".f32.so" if EDGE_F16 else ".f16.so")


def build_other_variant(verbose=True):
    """Build the library of the edge-kernel variant the default one was NOT built from (``libpackppi_hip.f32.so`` next to
    the default split-f16 build): every source is recompiled, since the weight packing and the edge embedding differ too.
    The GPU tests run the end-to-end parity cases on it as well (PACKPPI_LIB)."""
    out = other_variant_path()
    tag = "f32" if EDGE_F16 else "f16"
    sources = ["pp_api.hip", "pp_prepare.hip", "pp_node.hip", "pp_edge.hip" if EDGE_F16 else "pp_edge_f16.hip", "pp_clash.hip"]
    flags = [f for f in FLAGS if f != "-DPP_EDGE_F16"] + ([] if EDGE_F16 else ["-DPP_EDGE_F16"])
    if not needs_build(out, flags, sources):
        return out
    return _compile_and_link(out, sources, flags, tag, verbose)


def check_variant_path():
    return LIB.replace(".so", ".chk.so")


def build_check_variant(verbose=True):
    """``libpackppi_hip.chk.so``: the default kernels with -DPP_CHECK_RANGE (every value about to be split into f16 operands is
    compared with the f16 limit and counted; packppi_amd/rangecheck.py runs a checkpoint through it)."""
    os.environ.pop("PACKPPI_VARIANT_SOURCES", None)
    return build_library(verbose=verbose, extra_flags=["-DPP_CHECK_RANGE"], tag="chk")


def diag_variant_path():
    return LIB.replace(".so", ".dbg.so")


def build_diag_variant(verbose=True):
    """``libpackppi_hip.dbg.so``: the default kernels plus -DPP_DIAG -- the pp_debug_* exports (single launches, buffer copies,
    a prefix of one evaluation) and the launchers' environment switches (PP_NU_SPLIT, PP_EDGE_R ...).  Same kernels, same
    results; the tests that force launch shapes or read per-layer tensors run on it, the product libraries carry none of it."""
    os.environ.pop("PACKPPI_VARIANT_SOURCES", None)
    return build_library(verbose=verbose, extra_flags=["-DPP_DIAG"], tag="dbg")


def product_flag_stamps():
    """{flags half of pp_build_id(): library name} of the four libraries this file builds without laboratory flags."""
    f32_sources = ["pp_api.hip", "pp_prepare.hip", "pp_node.hip", "pp_edge.hip", "pp_clash.hip"]
    f16_sources = ["pp_api.hip", "pp_prepare.hip", "pp_node.hip", "pp_edge_f16.hip", "pp_clash.hip"]
    base = [f for f in FLAGS if f != "-DPP_EDGE_F16"]
    return {flags_hash(base + ["-DPP_EDGE_F16"], f16_sources): "default (split-f16)",
            flags_hash(base, f32_sources): "f32",
            flags_hash(base + ["-DPP_EDGE_F16", "-DPP_CHECK_RANGE"], f16_sources): "chk",
            flags_hash(base + ["-DPP_EDGE_F16", "-DPP_DIAG"], f16_sources): "dbg"}


if __name__ == "__main__":
    # python -m packppi_amd.build [--force] [--tag NAME -DFLAG ...]   (a tagged build is a variant library for experiments)
    argv = sys.argv[1:]
    tag = argv[argv.index("--tag") + 1] if "--tag" in argv else ""
    build_library(force="--force" in argv, extra_flags=[a for a in argv if a.startswith("-D")], tag=tag)
