"""Soak: repeated sampling on several sizes must be bit-reproducible; T1124 must stay within 1e-4 rad of the reference."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch, numpy as np
from bench import load_t1124
from packppi_amd import synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sizes = [int(a) for a in sys.argv[2:]] or [300, 520, 739, 800, 1100, 1600, 2300]       # soak.py <reps> [sizes ...]
bad = 0
for L in sizes:
    b = protein_to_batch(synth.make_complex(L, 11)).to("cuda:0")
    ctx = m._context(b)
    sched = torch.linspace(1, 0, 21)
    g = torch.Generator().manual_seed(L)
    init = ((torch.rand(1, L, 4, generator=g) * 2 - 1) * 3.0 * b.SC_D_mask.cpu()).to("cuda:0")
    outs = [ctx.sample(init, sched).cpu() for _ in range(reps + 1)]
    # majority output = reference; count the runs that deviate from it
    keys = [o.numpy().tobytes() for o in outs]
    from collections import Counter
    cnt = Counter(keys)
    maj = cnt.most_common(1)[0][1]
    nd = len(outs) - maj
    bad += nd
    print("L=%d: %d of %d runs deviate from the majority output (%d distinct outputs; 20 steps each)" % (L, nd, len(outs), len(cnt)), flush=True)
b, init, ref = load_t1124()
ctx = m._context(b.to("cuda:0"))
worst = 0.0
for _ in range(max(reps // 2, 3)):
    chi = ctx.sample(init.to("cuda:0"), torch.linspace(1, 0, 101)).cpu()
    d = (chi.double() - ref.double()).abs(); d = torch.minimum(d, (2 * np.pi - d).abs())[b.SC_D_mask.bool()]
    worst = max(worst, float(d.max()))
print("T1124 100 steps: worst max|dchi| over repeats %.2e" % worst)
print("SOAK", "FAIL" if bad or worst > 1e-4 else "OK")
