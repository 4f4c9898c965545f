"""Build-and-time variants of the hot kernels on the GPU box (timing only; a variant may be wrong by construction).

usage: python tools/debug/variants.py "name=-DFLAG1 -DFLAG2" "name2=..."     (first run is always the baseline)
Rebuilds pp_edge.o / pp_node.o with the extra flags, relinks, and reports in-situ kernel times and the pass time.
"""
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
init = init.to("cuda:0")
sched = m.schedule
for _ in range(2): ctx.sample(init, sched)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): chi = ctx.sample(init, sched)
torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 5 * 1e3
res = []
for which in (1, 0, 2):
    ctx.profile_kernel(which); ctx.sample(init, sched); res.append(ctx.profile_read()[0] * 1e3)
d = (chi.cpu().double() - ref.double()).abs().max().item() if ref is not None else -1
print("pass %%.2f ms | edge %%.1f us | node_msg %%.1f us | node_upd %%.1f us | max dchi %%.1e" %% (ms, res[0], res[1], res[2], d))
''' % ROOT


def rebuild(extra):
    hipcc = b._hipcc()
    for src in ("pp_edge.hip", "pp_node.hip"):
        subprocess.run([hipcc, *b.FLAGS, *extra, "-c", os.path.join(b.CSRC, src), "-o",
                        os.path.join(b.CSRC, src.replace(".hip", ".o"))], check=True, stderr=subprocess.DEVNULL)
    objs = [os.path.join(b.CSRC, s.replace(".hip", ".o")) for s in b.SOURCES]
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", b.LIB, *objs], check=True)


def main():
    b.build_library(verbose=False)
    variants = [("baseline", [])] + [(v.split("=", 1)[0], v.split("=", 1)[1].split()) for v in sys.argv[1:]]
    variants.append(("baseline-again", []))
    for name, flags in variants:
        rebuild(flags)
        out = subprocess.run([sys.executable, "-c", TIMER], capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED: " + out.stderr.strip()[-300:]
        print(f"{name:28s} {line}", flush=True)
    rebuild([])


if __name__ == "__main__":
    main()
