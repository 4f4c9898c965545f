"""CPU study: accuracy of bf16-split MFMA emulation (3 or 6 partial products) for the edge-level linears,
against the reference's own T1124 output (100 steps).  Usage: python tools/debug/bf16_split_study.py [steps]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import numpy as np, torch
import torch.nn.functional as F
from oracle import ref_cpu as R
from bench import load_t1124
from packppi_amd.weights import make_random_state_dict

torch.set_num_threads(8)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
batch, init, ref = load_t1124()
sd = make_random_state_dict(20251003)
orig_linear = R._linear

def split3(x):
    h = x.bfloat16().float(); r = x - h
    m = r.bfloat16().float(); r2 = r - m
    l = r2.bfloat16().float()
    return h, m, l

def split2h(x):
    h = x.half().float(); r = x - h
    l = r.half().float()
    return h, l

def make_linear_h(scaled):
    cache = {}
    def lin(x, w, b=None):
        if x.dim() != 4:
            return orig_linear(x, w, b)
        key = id(w)
        if key not in cache:
            cache[key] = split2h(w)
        wh, wl = cache[key]
        xh, xl = split2h(x)
        y = xh @ wh.T + (xh @ wl.T + xl @ wh.T)
        if b is not None:
            y = y + b
        return y
    return lin

def make_linear(nterms):
    cache = {}
    def lin(x, w, b=None):
        if x.dim() != 4:
            return orig_linear(x, w, b)
        key = id(w)
        if key not in cache:
            cache[key] = split3(w)
        wh, wm, wl = cache[key]
        xh, xm, xl = split3(x)
        y = xh @ wh.T + (xh @ wm.T + xm @ wh.T)
        if nterms == 6:
            y = y + (xh @ wl.T + xl @ wh.T + xm @ wm.T)
        if b is not None:
            y = y + b
        return y
    return lin

sched = torch.linspace(1, 0, steps + 1)
for name, n in (("fp32", 0), ("fp16x3", -3), ("bf16x6", 6)):
    R._linear = orig_linear if n == 0 else (make_linear_h(False) if n == -3 else make_linear(n))
    t0 = time.time()
    with torch.no_grad():
        chi = R.sampling(sd, batch, init.clone(), sched)
    if steps == 100:
        d = (chi.double() - ref.double()).abs()
    else:
        if n == 0:
            base = chi.clone()
        d = (chi.double() - base.double()).abs()
    d = torch.minimum(d, (2 * np.pi - d).abs())[batch["SC_D_mask"].bool()]
    print(f"{name}: max |dchi| = {d.max().item():.3e}  mean {d.mean().item():.3e}  ({time.time()-t0:.0f} s)", flush=True)
