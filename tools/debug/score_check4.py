import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
from oracle import ref_cpu as R
from packppi_amd import synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
torch.set_num_threads(16)
sd = make_random_state_dict(20251003)
m = TDiffusionModule(sd, device="cuda:0")
L = int(sys.argv[1])
b = protein_to_batch(synth.make_complex(L, 5)); init = b.SC_D.clone()
bd = b.to("cuda:0")
t = torch.full((L,), 0.02)
with torch.no_grad():
    so, ho = R.network(sd, b, init, t, None, True)
errs = []
for rep in range(6):
    sg, hg = m.network(bd, init.to("cuda:0"), t.to("cuda:0"))
    errs.append((hg.cpu() - ho).abs().reshape(-1, 128).max(1).values)
E = torch.stack(errs)
always = torch.nonzero((E > 1e-4).all(0)).flatten().tolist()
some = torch.nonzero((E > 1e-4).any(0) & ~(E > 1e-4).all(0)).flatten().tolist()
print("always bad:", always, [("%.2e" % E[:, i].max()) for i in always])
print("sometimes bad:", some[:20], [("%.2e" % E[:, i].max()) for i in some[:20]])
print("typical error (median of per-residue max): %.2e ; 99.9th pct %.2e" % (E.median(), E.flatten().kthvalue(int(E.numel()*0.999)).values))
for i in always[:3]:
    print("residue", i, "|hV| max", float(ho.reshape(-1,128)[i].abs().max()), "neighbours..", )
