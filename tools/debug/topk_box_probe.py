"""Is torch.topk / the distance arithmetic on THIS machine's CPU the same as where the goldens were made?"""
import ctypes, hashlib, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
from packppi_amd import lib as L
lib = L.load()
print("capability", torch.backends.cpu.get_cpu_capability(), "threads", torch.get_num_threads(), flush=True)
rng = np.random.default_rng(47)
side = 5
cells = rng.permutation(side ** 3)[:40]
ca = np.stack([cells % side, (cells // side) % side, cells // (side * side)], -1).astype(np.float32) * np.float32(3.8)
X = torch.from_numpy(ca)[None]
for nt in (torch.get_num_threads(), 1):
    torch.set_num_threads(nt)
    dX = X[:, None] - X[:, :, None]
    S = (dX ** 2).sum(3)
    D = torch.sqrt(S + 1e-6)
    E = torch.topk(D, 32, dim=-1, largest=False)[1]
    a, b, c = dX[..., 0] ** 2, dX[..., 1] ** 2, dX[..., 2] ** 2
    print("threads", nt, "sum==(a+b)+c", bool(((a + b) + c == S).all()), "D sha", hashlib.sha256(D.numpy().tobytes()).hexdigest()[:12],
          "E sha", hashlib.sha256(E.numpy().tobytes()).hexdigest()[:12], "row0 tail", E[0, 0, -3:].tolist(), flush=True)
    bad = 0
    Dn = D[0].numpy()
    for r in range(40):
        out = np.zeros(32, np.int32)
        lib.pp_topk_aten_host(np.ascontiguousarray(Dn[r]).ctypes.data_as(ctypes.c_void_p), 40, 32, out.ctypes.data_as(ctypes.c_void_p))
        bad += int(not np.array_equal(out, E[0, r].numpy()))
    print("  rows where the restated selection differs from torch.topk:", bad, flush=True)
    # same values, one row at a time / as 2-D / contiguous copies
    E1 = torch.stack([torch.topk(D[0, r].clone(), 32, largest=False)[1] for r in range(40)])
    print("  row-by-row topk equals batched:", bool(torch.equal(E1, E[0])), flush=True)
torch.set_num_threads(1)
def sha(t): return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()[:12]
dX = X[:, None] - X[:, :, None]
P = dX ** 2
S = P.sum(3)
Se = S + 1e-6
D = torch.sqrt(Se)
print("stages: X", sha(X), "dX", sha(dX), "P", sha(P), "P==dX*dX", bool((P == dX * dX).all()), "S", sha(S), "S+eps", sha(Se), "D", sha(D))
D64 = torch.sqrt(Se.double()).float()
print("sqrt correctly rounded:", bool((D == D64).all()), "n diff", int((D != D64).sum()))
Dn = torch.from_numpy(np.sqrt(Se.numpy()))
print("numpy sqrt == torch sqrt:", bool((Dn == D).all()), " numpy == exact:", bool((Dn == D64).all()))
big = torch.rand(1 << 20) * 1000
print("1M random: torch.sqrt vs exact diff", int((torch.sqrt(big) != torch.sqrt(big.double()).float()).sum()))
