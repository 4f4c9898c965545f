"""Accuracy of one network evaluation under rescaled / heavy-tailed weights: HIP vs the fp64 oracle, beside the fp32 oracle's
own distance from fp64 (a checkpoint's dynamic range differs from the seeded xavier draw every fixture uses)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from oracle import ref_cpu as O
from packppi_amd import synth
from packppi_amd.batch import Batch
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict

DEV = "cuda:0"
torch.set_num_threads(16)


def variants(sd):
    g = torch.Generator().manual_seed(1)
    lin = [k for k in sd if k.endswith("weight") and sd[k].dim() == 2]
    norm_w = [k for k in sd if "norm" in k and k.endswith("weight")]
    bias = [k for k in sd if k.endswith("bias") and "norm" not in k]
    out = {"xavier (fixtures)": dict(sd)}
    for name, f in (("linear x4", 4.0), ("linear x1/32", 1 / 32.), ("linear x1/1024", 1 / 1024.), ("linear x16", 16.0)):
        v = dict(sd)
        for k in lin:
            v[k] = sd[k] * f
        out[name] = v
    v = dict(sd)
    for k in norm_w:
        v[k] = sd[k] * 5.0
    for k in bias:
        v[k] = sd[k] + (torch.rand(sd[k].shape, generator=g) * 6 - 3)
    out["LN gain x5, biases +-3"] = v
    v = dict(sd)
    for k in lin:
        m = torch.rand(sd[k].shape, generator=g) < 0.01
        v[k] = torch.where(m, sd[k] * 30.0, sd[k])
    out["heavy tails (1% x30)"] = v
    v = dict(sd)
    for k in lin:
        v[k] = sd[k] * 1e-3 if "W_out" in k or "node_dense.W_out" in k else sd[k]
    out["tiny output layers (x1e-3)"] = v
    return out


def to64(b):
    o = Batch()
    for k, v in b.items():
        o[k] = v.double() if isinstance(v, torch.Tensor) and v.dtype == torch.float32 else v
    return o


b = protein_to_batch(synth.make_complex(96, 5))
g = torch.Generator().manual_seed(2)
chi = (torch.rand(1, 96, 4, generator=g) * 2 - 1) * 3.0 * b.SC_D_mask
t = torch.full((96,), 0.4)
for name, sd in variants(make_random_state_dict(20251003)).items():
    with torch.no_grad():
        s32, h32 = O.network(sd, b, chi, t)
        s64, h64 = O.network({k: v.double() for k, v in sd.items()}, to64(b), chi.double(), t.double())
    m = TDiffusionModule(sd, device=DEV)
    s, h = m.network(b.to(DEV), chi.to(DEV), t)
    s, h = s.cpu().double(), h.cpu().double()
    rel = lambda a, ref: float((a - ref).abs().max() / ref.abs().max())
    print(f"{name:30s} |h_V| {float(h64.abs().max()):9.3e} |score| {float(s64.abs().max()):9.3e}   "
          f"h_V: hip-64 {rel(h, h64):.2e} o32-64 {rel(h32.double(), h64):.2e} hip-o32 {rel(h, h32.double()):.2e}   "
          f"score: hip-64 {rel(s, s64):.2e} o32-64 {rel(s32.double(), s64):.2e} hip-o32 {rel(s, s32.double()):.2e}",
          flush=True)
