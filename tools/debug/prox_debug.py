import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch, numpy as np
from tests.conftest import load_golden, wrapped_absdiff
from oracle import ref_cpu as O
from packppi_amd.functional import _ctx_for, proximal_optimizer
from packppi_amd.batch import Batch
DEV = "cuda:0"
for name in ("g3_proximal_L64", "g3_proximal_L120"):
    b, g = load_golden(name); gb = b.to(DEV)
    init = g["init_chi_seed11"]
    pr_o, gr_o = O.clash_and_grad(b, init)
    b64 = Batch({k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in b.items()})
    pr64, gr64 = O.clash_and_grad(b64, init.double())
    pr_g, gr_g = _ctx_for(gb).clash(init.to(DEV), 12., .5, need_grad=True)
    e_gpu = (gr_g.cpu().double() - gr64).abs(); e_o = (gr_o.double() - gr64).abs()
    print(name, "grad max", gr64.abs().max().item(), "err gpu-vs-64", e_gpu.max().item(), "err cpu32-vs-64", e_o.max().item())
    nz = gr64 != 0
    print("   nonzero entries", int(nz.sum()), "gpu zero where ref nonzero", int(((gr_g.cpu()==0) & nz).sum()), "gpu nonzero where ref zero", int(((gr_g.cpu()!=0) & ~nz).sum()))
    rel = e_gpu[nz] / gr64.abs()[nz]
    print("   rel err quantiles", np.quantile(rel.numpy(), [0.5, 0.9, 0.99, 1.0]))
    small = nz & (gr64.abs() < 1e-6)
    print("   entries with |g|<1e-6:", int(small.sum()), gr64[small][:8].tolist(), gr_g.cpu()[small][:8].tolist(), gr_o[small][:8].tolist())
    chis, losses = proximal_optimizer(gb, init.to(DEV), 12., .5, 1., 50)
    co, lo = O.proximal_optimizer(b, init.clone(), 12., .5, 1., 50)
    c64, l64 = O.proximal_optimizer(b64, init.double(), 12., .5, 1., 50)
    for t in (0, 1, 2, 4, 9, 19, 49):
        dg = wrapped_absdiff(chis[t].cpu(), co[t]); d64 = wrapped_absdiff(co[t], c64[t])
        print(f"   step {t}: gpu-vs-cpu32 max {dg.max():.2e} n>1e-4 {int((dg>1e-4).sum())} | cpu32-vs-64 max {d64.max():.2e} n>1e-4 {int((d64>1e-4).sum())}")
    dg = wrapped_absdiff(chis[49].cpu(), co[49]); bad = (dg > 1e-4).nonzero()
    for idx in bad[:12]:
        i = tuple(idx.tolist())
        print("     bad", i, "restype", int(b.residue_type[0, i[1]]), "dchi", float(dg[i]), "g0 ref", float(gr64[i]), "g0 gpu", float(gr_g.cpu()[i]))
b, g = load_golden("g2_ops_B3")
from packppi_amd.module import TDiffusionModule
from packppi_amd.weights import make_random_state_dict
m = TDiffusionModule(make_random_state_dict(20251003), device=DEV)
E, hE = m._context(b.to(DEV)).graph(); E = E.cpu(); hE = hE.cpu()
valid = b.residue_mask.bool()
def valid_sets(idx):
    ok = torch.gather(b.residue_mask[:, None].expand(-1, idx.shape[1], -1), 2, idx) > 0
    return torch.where(ok, idx, torch.full_like(idx, -1)).sort(-1)[0]
print("B3 sets equal", torch.equal(valid_sets(E)[valid], valid_sets(g["E_idx"])[valid]))
ca = b.X[:, :, 1, :]
for E_any in (E, g["E_idx"]):
    d = (ca[:, :, None, :] - torch.gather(ca[:, None].expand(-1, ca.shape[1], -1, -1), 2, E_any[..., None].expand(-1, -1, -1, 3))).norm(dim=-1)
    dd = (d[..., 1:] - d[..., :-1])[valid]
    print("  min diff", dd.min().item())
