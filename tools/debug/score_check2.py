"""GPU-vs-GPU reproducibility and GPU-vs-oracle on T1124 and on a synthetic complex of the same size."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch, numpy as np
from bench import load_t1124
from oracle import ref_cpu as R
from packppi_amd import synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
torch.set_num_threads(16)
sd = make_random_state_dict(20251003)
m = TDiffusionModule(sd, device="cuda:0")
b1, init1, _ = load_t1124()
b2 = protein_to_batch(synth.make_complex(739, 5))
for name, b, init in (("T1124", b1, init1), ("synthetic739", b2, b2.SC_D.clone())):
    bd = b.to("cuda:0")
    t = torch.full((b["residue_type"].numel(),), 0.5)
    with torch.no_grad():
        so, ho = R.network(sd, b, init, t, None, True)
    outs = []
    for rep in range(4):
        sg, hg = m.network(bd, init.to("cuda:0"), t.to("cuda:0"))
        outs.append(hg.cpu().clone())
        worst = (hg.cpu() - ho).abs().reshape(-1, 128).max(1).values
        print(name, "rep", rep, "vs oracle: residues > 1e-4:", int((worst > 1e-4).sum()), "max %.2e" % worst.max(),
              "| vs rep0: max %.2e" % (outs[-1] - outs[0]).abs().max(), "nan:", int(torch.isnan(hg).sum()))
    print(name, "masked residues:", int((b["residue_mask"] == 0).sum()), "K=", min(32, b["residue_type"].shape[1]))
