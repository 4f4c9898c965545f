"""Diagnostic: per-stage s_memtime stamps of the edge-update kernel (build with -DPP_X_STAMP, run with PP_STAMP=1)."""
import os, sys, ctypes
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
os.environ["PP_STAMP"] = "1"
import numpy as np, torch
import packppi_amd.build as b
b.FLAGS.append("-DPP_X_STAMP"); b.build_library(force=True, verbose=False)
from packppi_amd import lib as L
from bench import load_t1124
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
bt, init, ref = load_t1124()
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
ctx = m._context(bt.to("cuda:0"))
chi = ctx.sample(init.to("cuda:0"), torch.linspace(1, 0, 3))
ctx.time_kernel(1, 1)           # last launch: edge update
N = 739
buf = np.zeros(N * 4 * 64 * 4, np.uint64)
lib = L.load()
lib.pp_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
assert lib.pp_debug_stamps(ctx.handle, buf.ctypes.data, buf.size) == 0
t = buf.reshape(N, 4, 64, 4).astype(np.int64)
ns = 45
valid = [n for n in range(N) if t[n, 0, 0, 0] > 0]
t0 = min(t[n, w, 0, 0] for n in valid for w in range(4))
print("WGs with stamps", len(valid), " clock ticks are 100 MHz s_memtime? first/last:", t0)
for n in (valid[0], valid[len(valid)//2], valid[-1]):
    w = 0
    st = t[n, w, :ns]
    print(f"WG {n} wave {w}: start {st[0,0]-t0}  end {st[ns-1,3]-t0}  total {st[ns-1,3]-st[0,0]}")
    comp = st[:, 1] - st[:, 0]; store = st[:, 2] - st[:, 1]; bar = st[:, 3] - st[:, 2]; gap = st[1:, 0] - st[:-1, 3]
    print("   compute mean %.0f  store(wait loads) mean %.0f  barrier mean %.0f  inter-stage gap mean %.0f" % (comp.mean(), store.mean(), bar.mean(), gap.mean()))
    print("   compute per stage:", comp[:46].tolist())
    print("   store per stage:  ", store[:46].tolist())
    print("   barrier per stage:", bar[:46].tolist())
    print("   gap per stage:    ", gap[:45].tolist())
    print("   sum compute %d store %d barrier %d gap %d" % (comp[:45].sum(), store[:45].sum(), bar[:45].sum(), gap[:44].sum()))
starts = np.array([t[n, 0, 0, 0] - t0 for n in valid]); ends = np.array([t[n, 0, ns-1, 3] - t0 for n in valid])
print("start spread: min %d max %d ; end: min %d max %d" % (starts.min(), starts.max(), ends.min(), ends.max()))
