"""pp_sample with replayed graph blocks (default) against plain launches (PP_GRAPH=0): bit-identical angles (each variant in
its own process: the switch is read once per process)."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from packppi_amd import synth
    from packppi_amd.batch import pack
    from packppi_amd.featurize import protein_to_batch
    from packppi_amd.module import TDiffusionModule
    from packppi_amd.weights import make_random_state_dict
    sd = make_random_state_dict(20251003)
    m = TDiffusionModule(sd, device="cuda:0")
    outs = {}
    for L, steps in ((40, 25), (300, 37), (739, 100), (1500, 20)):
        b = protein_to_batch(synth.make_complex(L, 11)).to("cuda:0")
        g = torch.Generator().manual_seed(L)
        init = ((torch.rand(1, L, 4, generator=g) * 2 - 1) * 3.0 * b.SC_D_mask.cpu()).to("cuda:0")
        ctx = m._context(b)
        outs[f"ode_{L}"] = ctx.sample(init, torch.linspace(1, 0, steps + 1)).cpu()
        outs[f"ode_{L}_again"] = ctx.sample(init, torch.linspace(1, 0, steps + 1)).cpu()          # cached block
        nz = torch.randn(steps, 2, L, 4, generator=torch.Generator().manual_seed(5)).to("cuda:0")
        outs[f"sde_{L}"] = ctx.sample(init, torch.linspace(1, 0, steps + 1), "sde", nz).cpu()
    cs = [protein_to_batch(synth.make_complex(270 + 7 * k, 600 + k)) for k in range(6)]
    pb = pack(cs).to("cuda:0")
    init = ((torch.rand(1, pb.max_size, 4, generator=torch.Generator().manual_seed(9)) * 2 - 1) * 3.0).to("cuda:0") * pb.SC_D_mask
    outs["packed"] = m._context(pb).sample(init, torch.linspace(1, 0, 31)).cpu()
    torch.save(outs, sys.argv[2])
    sys.exit(0)
import torch
paths = {}
for gmode in ("1", "0"):
    paths[gmode] = os.path.join(ROOT, "gpurun_out", f"graph{gmode}.pt")
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", paths[gmode]], env=dict(os.environ, PP_GRAPH=gmode), check=True)
a, b = torch.load(paths["1"]), torch.load(paths["0"])
ok = True
for k in a:
    same = torch.equal(a[k], b[k]) and bool(torch.isfinite(a[k]).all())
    ok &= same
    print(k, "bit-identical" if same else "DIFFERENT max %.3e" % float((a[k] - b[k]).abs().max()))
print("GRAPH", "OK" if ok else "FAIL")
