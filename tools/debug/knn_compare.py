"""Neighbour sets of the built-in kNN search vs torch.topk (the reference's) on the C5 complexes; prints rows that differ."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import torch
from oracle import ref_cpu as O
from packppi_amd import synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict

m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
lens = synth.c5_lengths(256)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 32):
    b = protein_to_batch(synth.make_complex(lens[i], 10000 + i))
    E = m._context(b.to("cuda:0")).graph()[0].cpu()
    E_ref = O.knn_graph(b.X[:, :, 1, :], b.residue_mask)
    rows = torch.nonzero((E.sort(-1)[0] != E_ref.sort(-1)[0]).any(-1)[0]).flatten().tolist()
    if rows:
        ca = b.X[0, :, 1, :]
        for r in rows:
            mine, ref = set(E[0, r].tolist()), set(E_ref[0, r].tolist())
            a, c = sorted(mine - ref), sorted(ref - mine)
            d = lambda j: float(torch.sqrt(((ca[r] - ca[j]) ** 2).sum() + 1e-6))
            print(f"complex {i} (L={lens[i]}) row {r}: mine-only {a} d={[d(j) for j in a]}  ref-only {c} d={[d(j) for j in c]}", flush=True)
print("done")
