"""Soak of the packed multi-complex path and the proximal stage: repeated runs must agree bit for bit."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from packppi_amd import synth
from packppi_amd.batch import pack
from packppi_amd.featurize import protein_to_batch
from packppi_amd.functional import proximal_optimizer
from packppi_amd.lib import Context
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = "cuda:0"
m = TDiffusionModule(make_random_state_dict(20251003), device=dev)
lens = synth.c5_lengths(256)[:32]
cs = [protein_to_batch(synth.make_complex(n, 10000 + i)).to(dev) for i, n in enumerate(lens)]
g = torch.Generator().manual_seed(7)
x0 = torch.cat([(torch.rand(1, n, 4, generator=g) * 2 - 1) * 3.0 for n in lens], 1).to(dev)
pb = pack(cs)
x0 = x0 * pb.SC_D_mask
sched = torch.linspace(1, 0, 31)
ref = Context(m._plan, pb).sample(x0, sched).cpu()
bad = 0
for r in range(reps):
    out = Context(m._plan, pb).sample(x0, sched).cpu()
    bad += int(not torch.equal(out, ref))
print(f"packed 32 complexes ({sum(lens)} rows), 30 steps, fresh context each time: {bad} of {reps} runs deviate", flush=True)
b = protein_to_batch(synth.make_complex(1500, 1500)).to(dev)
chi0 = ((torch.rand(1, 1500, 4, generator=g) * 2 - 1) * 3.0).to(dev) * b.SC_D_mask
c0, l0 = proximal_optimizer(b, chi0, 12.0, 0.5, 1.0, 50)
pbad = 0
for r in range(max(reps // 2, 1)):
    c1, l1 = proximal_optimizer(b, chi0, 12.0, 0.5, 1.0, 50)
    pbad += int(l1 != l0 or not all(torch.equal(a, c) for a, c in zip(c0, c1)))
print(f"proximal, 1500 residues, 50 steps: {pbad} of {max(reps // 2, 1)} runs deviate", flush=True)
print("SOAK OK" if bad + pbad == 0 else "SOAK FAILED")
