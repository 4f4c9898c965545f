import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from packppi_amd import synth
from packppi_amd.featurize import protein_to_batch
from oracle import ref_cpu as O
f32, f64 = np.float32, np.float64
def fma(x, y, z): return (x.astype(f64) * y.astype(f64) + z.astype(f64)).astype(f32)
def cross(a, b):
    out = np.empty_like(a)
    for comp, (i1, j2, i2, j1) in enumerate(((1, 2, 2, 1), (2, 0, 0, 2), (0, 1, 1, 0))):
        q = (a[..., i2] * b[..., j1]).astype(f32)
        out[..., comp] = fma(a[..., i1], b[..., j2], -q)
    return out
def norm(v): return np.sqrt(fma(v[..., 2], v[..., 2], fma(v[..., 1], v[..., 1], (v[..., 0] * v[..., 0]).astype(f32))))
def dot(a, b):
    p = [(a[..., k] * b[..., k]).astype(f32) for k in range(3)]
    return ((p[0] + p[1]).astype(f32) + p[2]).astype(f32)
def unit(v):
    with np.errstate(all="ignore"):
        q = (v / norm(v)[..., None]).astype(f32)
    return np.nan_to_num(q, nan=0.0, posinf=np.finfo(f32).max, neginf=np.finfo(f32).min)
def dih_arg(p0, p1, p2, p3):
    u0, u1, u2 = (p2 - p1).astype(f32), (p0 - p1).astype(f32), (p3 - p2).astype(f32)
    n1, n2 = unit(cross(u0, u1)), unit(cross(u0, u2))
    return dot(n1, n2), np.sign(dot(cross(u1, u2), u0))
tot = bad = 0
for i in (9, 2, 12, 0):
    b = protein_to_batch(synth.make_complex(synth.c5_lengths(256)[i], 10000 + i))
    X = b.X
    E = O.knn_graph(X[:, :, 1, :], b.residue_mask)
    N, CA, C = X[:, :, 0], X[:, :, 1], X[:, :, 2]
    g = lambda v: O._gather_nodes(v, E)
    K = E.shape[-1]
    Ci, Ni, CAi = (v[:, :, None, :].expand(-1, -1, K, -1) for v in (C, N, CA))
    Nj, CAj, Cj = g(N), g(CA), g(C)
    for (p0, p1, p2, p3), name in (((Ci, Nj, CAj, Cj), "phi"), ((Ni, CAi, Ci, Nj), "psi")):
        # torch's own intermediate: the acos argument and sign
        u0, u1, u2 = p2 - p1, p0 - p1, p3 - p2
        unit_t = lambda v: torch.nan_to_num(v / torch.norm(v, dim=-1, keepdim=True))
        arg_t = (unit_t(torch.cross(u0, u1, dim=-1)) * unit_t(torch.cross(u0, u2, dim=-1))).sum(-1).numpy()
        sgn_t = torch.sign((torch.cross(u1, u2, dim=-1) * u0).sum(-1)).numpy()
        arg_e, sgn_e = dih_arg(*(t.numpy().copy() for t in (p0, p1, p2, p3)))
        same = (arg_t == arg_e) | (np.isnan(arg_t) & np.isnan(arg_e))
        tot += arg_t.size; bad += int((~same).sum()) + int((sgn_t != sgn_e).sum())
        print(i, name, "arg mismatches", int((~same).sum()), "sign mismatches", int((sgn_t != sgn_e).sum()), " |arg|>1:", int((np.abs(arg_t) > 1).sum()), " nan args:", int(np.isnan(arg_t).sum()))
print("total", tot, "bad", bad)
import numpy as np, torch
torch.set_num_threads(1)
rng = np.random.default_rng(0)
a = (rng.standard_normal((200000, 3)) * 3).astype(np.float32)
b = (rng.standard_normal((200000, 3)) * 3).astype(np.float32)
ta, tb = torch.from_numpy(a), torch.from_numpy(b)
c = torch.cross(ta, tb, dim=-1).numpy()
f64 = np.float64
def fma(x, y, z):  # exact fused multiply-add rounded once to f32
    return (x.astype(f64) * y.astype(f64) + z.astype(f64)).astype(np.float32)   # double rounding risk negligible? use exactness: products of f32 are exact in f64, sum rounded to f64 then f32 (double rounding possible but rare)
def variants(a1, b2, a2, b1):
    p, q = (a1 * b2).astype(np.float32), (a2 * b1).astype(np.float32)
    return {"sep": (p - q).astype(np.float32), "fms_first": fma(a1, b2, -q), "fnma_second": fma(-a2, b1, p)}
for comp, (i1, j2, i2, j1) in enumerate(((1, 2, 2, 1), (2, 0, 0, 2), (0, 1, 1, 0))):
    v = variants(a[:, i1], b[:, j2], a[:, i2], b[:, j1])
    print("cross comp", comp, {k: int((x != c[:, comp]).sum()) for k, x in v.items()})
# norm
n = torch.norm(ta, dim=-1, keepdim=True).numpy()[:, 0]
x, y, z = a[:, 0], a[:, 1], a[:, 2]
sq = lambda t: (t * t).astype(np.float32)
cands = {"((x2+y2)+z2)": np.sqrt(((sq(x) + sq(y)).astype(np.float32) + sq(z)).astype(np.float32)),
         "fma chain": np.sqrt(fma(z, z, fma(y, y, sq(x)))),
         "f64 accumulate": np.sqrt((x.astype(f64)**2 + y.astype(f64)**2 + z.astype(f64)**2)).astype(np.float32),
         "sqrt in f32 of f64 sum": np.sqrt((x.astype(f64)**2 + y.astype(f64)**2 + z.astype(f64)**2).astype(np.float32))}
print("norm", {k: int((v != n).sum()) for k, v in cands.items()})
# (n1*n2).sum(-1)
s = (ta * tb).sum(-1).numpy()
p = [(a[:, k] * b[:, k]).astype(np.float32) for k in range(3)]
c2 = {"((p0+p1)+p2)": ((p[0] + p[1]).astype(np.float32) + p[2]).astype(np.float32),
      "(p0+(p1+p2))": (p[0] + (p[1] + p[2]).astype(np.float32)).astype(np.float32),
      "f64": (p[0].astype(f64) + p[1] + p[2]).astype(np.float32)}
print("sum3", {k: int((v != s).sum()) for k, v in c2.items()})
# division and acos
q = (ta / torch.from_numpy(n)[:, None]).numpy()
print("div", int((q[:, 0] != (a[:, 0] / n).astype(np.float32)).sum()))
u = torch.rand(200000) * 2 - 1
ac = torch.arccos(u).numpy()
print("acos vs np.arccos f32:", int((ac != np.arccos(u.numpy())).sum()), " vs f64->f32:", int((ac != np.arccos(u.numpy().astype(f64)).astype(np.float32)).sum()))
