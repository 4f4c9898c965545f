"""PP_NU_WAVES=4 (the 4-wave node update, run standalone) against the default 8-wave kernel: bit-identical sampling results
on a few shapes (each variant in its own process: the switch is read once per process)."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from packppi_amd import synth
    from packppi_amd.featurize import protein_to_batch
    from packppi_amd.module import TDiffusionModule
    from packppi_amd.weights import make_random_state_dict
    m = TDiffusionModule(make_random_state_dict(20251003), device="cuda:0")
    outs = {}
    for L in (40, 300, 739, 1500):
        b = protein_to_batch(synth.make_complex(L, 11)).to("cuda:0")
        g = torch.Generator().manual_seed(L)
        init = ((torch.rand(1, L, 4, generator=g) * 2 - 1) * 3.0 * b.SC_D_mask.cpu()).to("cuda:0")
        for mode in ("ode",):
            outs[f"{L}"] = m._context(b).sample(init, torch.linspace(1, 0, 13)).cpu()
        s, h = m.network(b, init, torch.full((L,), 0.4))
        outs[f"{L}_score"] = s.cpu(); outs[f"{L}_hV"] = h.cpu()
    torch.save(outs, sys.argv[2])
    sys.exit(0)
import torch
paths = {}
for w in ("8", "4"):
    paths[w] = os.path.join(ROOT, "gpurun_out", f"nu{w}.pt")
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", paths[w]], env=dict(os.environ, PP_NU_WAVES=w), check=True)
a, b = torch.load(paths["8"]), torch.load(paths["4"])
ok = True
for k in a:
    same = torch.equal(a[k], b[k])
    ok &= same
    print(k, "bit-identical" if same else "DIFFERENT max %.3e" % float((a[k] - b[k]).abs().max()))
print("NU4", "OK" if ok else "FAIL")
