"""Timing-only variants of the clash / atom14 kernels (pp_clash.hip) on the GPU box.
usage: python tools/debug/clash_variants.py "name=-DFLAG" ...   (reports the average time of a pp_clash call = atom14 + clash)"""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import packppi_amd.build as b

TIMER = r'''
import sys, time, torch
sys.path.insert(0, %r)
from bench import load_t1124
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
b, init, ref = load_t1124()
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
ctx = m._context(b.to("cuda:0"))
chi = (ref if ref is not None else init).to("cuda:0").float()
for need_grad in (False, True):
    for _ in range(5): ctx.clash(chi, need_grad=need_grad)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): ctx.clash(chi, need_grad=need_grad)
    torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 50 * 1e6
    print("grad=%%d %%.1f us per pp_clash call" %% (need_grad, us), end=" | ")
for _ in range(5): ctx.atom14(chi)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): ctx.atom14(chi)
torch.cuda.synchronize(); print("atom14 alone %%.1f us" %% ((time.perf_counter() - t0) / 50 * 1e6))
''' % ROOT


def rebuild(extra):
    hipcc = b._hipcc()
    subprocess.run([hipcc, *b.FLAGS, *extra, "-c", os.path.join(b.CSRC, "pp_clash.hip"), "-o",
                    os.path.join(b.CSRC, "pp_clash.o")], check=True, stderr=subprocess.DEVNULL)
    objs = [os.path.join(b.CSRC, s.replace(".hip", ".o")) for s in b.SOURCES]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", b.LIB, *objs], check=True)


def main():
    b.build_library(verbose=False)
    variants = [("baseline", [])] + [(v.split("=", 1)[0], v.split("=", 1)[1].split()) for v in sys.argv[1:]]
    for name, flags in variants:
        rebuild(flags)
        out = subprocess.run([sys.executable, "-c", TIMER], capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED: " + out.stderr.strip()[-300:]
        print(f"{name:20s} {line}", flush=True)
    rebuild([])


if __name__ == "__main__":
    main()
