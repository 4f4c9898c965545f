"""Container-only: the UNMODIFIED reference on data/T1124_lig.pdb, 100 steps, under weight variants outside the seeded
xavier statistics (tools/oracle/envelope_weights.py), in fp32 and in fp64 -> tests/golden/g10_envelope_T1124.npz.

    PYTHONDONTWRITEBYTECODE=1 python tools/oracle/make_golden_envelope.py [--threads 8]

The fp64 run is the arbiter, |ref32 - ref64| says how well conditioned 100 reverse steps are under those weights (test:
|HIP - ref64| <= max(1e-4, 3 |ref32 - ref64|)).  Only data is written: the batch and the initial noise are g4_T1124's, the
weights are regenerated from the seed by the test.
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refshim  # noqa: E402
from envelope_weights import envelope_variants, small_ln_gain_variants, tiny_operand_variants  # noqa: E402
from make_golden_prox import load_fixture, ref_batch, wrapped  # noqa: E402
from packppi_amd.weights import make_random_state_dict  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def run(model, rb, init, n_steps, dtype):
    model.schedule = torch.linspace(1, 0, n_steps + 1, dtype=dtype)
    model.add_sc_noise = lambda b, t: (init.clone(), None)
    with torch.no_grad():
        return model.sampling(rb)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--append", action="store_true", help="keep the variants the fixture already holds, run only the missing ones")
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    z, b = load_fixture("g4_T1124")
    init = torch.from_numpy(z["init_chi_seed1124"]).float()
    sd0 = make_random_state_dict(20251003)
    out = {"variants": [], "steps": np.int64(args.steps)}
    variants = dict(envelope_variants(sd0))
    variants.update({"tiny operands: " + k: v for k, v in tiny_operand_variants(sd0).items()})
    variants.update(small_ln_gain_variants(sd0))
    path = os.path.join(GOLD, "g10_envelope_T1124.npz")
    old = dict(np.load(path)) if args.append and os.path.exists(path) else {}
    if old:
        assert int(old["steps"]) == args.steps
    for name, sd in variants.items():
        key = name.replace(" ", "_").replace(",", "").replace("/", "_").replace(":", "").replace("+-", "pm")
        if "chi64." + key in old:
            out["variants"].append(name)
            for pre in ("chi32.", "chi64.", "cond."):
                out[pre + key] = old[pre + key]
            print(f"{name}: kept (|ref32 - ref64| = {float(old['cond.' + key]):.2e} rad)", flush=True)
            continue
        res = {}
        for prec, dt in (("32", torch.float32), ("64", torch.float64)):
            model = refshim.build_reference_module(0)
            model.load_state_dict(sd, strict=True)
            model = model.to(dt).eval()
            t0 = time.time()
            res[prec] = run(model, ref_batch(b, dt == torch.float64), init.to(dt), args.steps, dt)
            print(f"  {name} fp{prec}: {time.time() - t0:.0f}s", flush=True)
        m = b["SC_D_mask"].bool()
        cond = float(wrapped(res["32"], res["64"])[m].max())
        print(f"{name}: |ref32 - ref64| = {cond:.2e} rad", flush=True)
        out["variants"].append(name)
        out["chi32." + key] = res["32"].numpy()
        out["chi64." + key] = res["64"].numpy()
        out["cond." + key] = np.float64(cond)
    out["variants"] = np.array(out["variants"])
    np.savez_compressed(path, **out)
    print(f"wrote {path} {os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()
