"""Repeat one full network evaluation (8 launches) and compare hV / score bit for bit."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from collections import Counter
from packppi_amd import synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
L = int(sys.argv[1]); reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
b = protein_to_batch(synth.make_complex(L, 5)).to("cuda:0")
t = torch.full((L,), 0.5, device="cuda:0")
outs = []
for _ in range(reps):
    s_, h_ = m.network(b, b.SC_D, t)
    outs.append(h_.cpu().reshape(L, 128))
keys = [o.numpy().tobytes() for o in outs]
cnt = Counter(keys); maj_key = cnt.most_common(1)[0][0]
maj = outs[keys.index(maj_key)]
dev = [i for i, k in enumerate(keys) if k != maj_key]
print("L=%d: %d of %d evaluations deviate from the majority" % (L, len(dev), reps))
for i in dev[:5]:
    bad = torch.nonzero((outs[i] != maj).any(1)).flatten().tolist()
    print("   run", i, ": residues", bad[:10], "(%d)" % len(bad), "max %.2e" % (outs[i] - maj).abs().max())
