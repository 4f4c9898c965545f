"""h_E0 (static edge embedding) and one network evaluation of the HIP path vs the oracle on a C5 complex."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import numpy as np, torch
from oracle import ref_cpu as O
from packppi_amd import synth
from packppi_amd.featurize import protein_to_batch
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
DEV = "cuda:0"
torch.set_num_threads(16)
sd = make_random_state_dict(20251003)
m = TDiffusionModule(sd, device=DEV)
lens = synth.c5_lengths(256)
z = np.load("tests/golden/g7_c5_rank0.npz")
for i in [int(a) for a in sys.argv[1:]] or [9]:
    b = protein_to_batch(synth.make_complex(lens[i], 10000 + i))
    ctx = m._context(b.to(DEV))
    E, hE = ctx.graph()
    E, hE = E.cpu(), hE.cpu()
    E_o, hE_o = O.encode_static(sd, b, zero_self_dihedral=True)
    assert torch.equal(E.sort(-1)[0], E_o.sort(-1)[0])
    # align slots by neighbour id
    pm, po = E.argsort(-1), E_o.argsort(-1)
    a = torch.gather(hE, 2, pm[..., None].expand(-1, -1, -1, 128))
    c = torch.gather(hE_o, 2, po[..., None].expand(-1, -1, -1, 128))
    d = (a - c).abs().amax(-1)[0]
    r, k = np.unravel_index(int(d.argmax()), d.shape)
    print(f"complex {i}: h_E0 max dev {float(d.max()):.3e} at row {r}, neighbour {int(E.sort(-1)[0][0, r, k])}; edges > 1e-4: {int((d > 1e-4).sum())}")
    init = torch.from_numpy(z[f"init_{i}"])
    for t in (1.0, 0.5, 0.02):
        s_h, hV_h = ctx.score(init.to(DEV), t)
        with torch.no_grad():
            s_o, hV_o = O.network(sd, b, init, torch.full((lens[i],), t), None, True)
        print(f"   t={t}: score dev {float((s_h.cpu() - s_o).abs().max()):.3e}  h_V dev {float((hV_h.cpu() - hV_o).abs().max()):.3e}")
