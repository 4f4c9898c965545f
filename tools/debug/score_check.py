"""One network evaluation on T1124: HIP vs the CPU oracle; where do they differ?"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch, numpy as np
from bench import load_t1124
from oracle import ref_cpu as R
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
torch.set_num_threads(16)
b, init, ref = load_t1124()
sd = make_random_state_dict(20251003)
m = TDiffusionModule(sd, device="cuda:0")
bd = b.to("cuda:0")
for tval in (1.0, 0.5, 0.02, 0.02, 0.02, 0.02):
    t = torch.full((b["residue_type"].numel(),), tval)
    with torch.no_grad():
        so, ho = R.network(sd, b, init, t, None, True)
    sg, hg = m.network(bd, init.to("cuda:0"), t.to("cuda:0"))
    ds = (sg.cpu() - so).abs(); dh = (hg.cpu() - ho).abs()
    print("t=%.2f score max|d| %.3e (|score| max %.2f)  hV max|d| %.3e (|hV| max %.2f)" % (tval, ds.max(), so.abs().max(), dh.max(), ho.abs().max()))
    worst = dh.reshape(-1, 128).max(1).values
    idx = torch.topk(worst, 8).indices
    print("   worst residues:", idx.tolist(), ["%.1e" % worst[i] for i in idx], "mask", b["residue_mask"].reshape(-1)[idx].tolist())
    print("   residues with hV err > 1e-3:", int((worst > 1e-3).sum()), " > 1e-4:", int((worst > 1e-4).sum()), "of", worst.numel())
