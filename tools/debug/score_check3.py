import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch, numpy as np
from oracle import ref_cpu as R
from packppi_amd import synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
torch.set_num_threads(16)
sd = make_random_state_dict(20251003)
m = TDiffusionModule(sd, device="cuda:0")
for L in [int(a) for a in sys.argv[1:]]:
    b = protein_to_batch(synth.make_complex(L, 5)); init = b.SC_D.clone()
    bd = b.to("cuda:0")
    t = torch.full((L,), 0.02)
    with torch.no_grad():
        so, ho = R.network(sd, b, init, t, None, True)
    bad_all = []
    for rep in range(6):
        sg, hg = m.network(bd, init.to("cuda:0"), t.to("cuda:0"))
        worst = (hg.cpu() - ho).abs().reshape(-1, 128).max(1).values
        bad = torch.nonzero(worst > 1e-4).flatten().tolist()
        bad_all.append(bad)
    print("L=%d bad residue indices per rep:" % L, [(len(b_), (min(b_), max(b_)) if b_ else None) for b_ in bad_all], flush=True)
    if any(bad_all): print("   example:", [b_[:12] for b_ in bad_all if b_][0])
