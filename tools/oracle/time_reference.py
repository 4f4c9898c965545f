"""Container-only: time the UNMODIFIED reference (imported through refshim) on data/T1124_lig.pdb, the configuration
BASELINE.json's metric is quoted on, and write profiles/reference_cpu_rates.json (bench.py attaches it to its cpu_baseline).

    PYTHONDONTWRITEBYTECODE=1 python tools/oracle/time_reference.py [--no-grad-steps 10] [--grad-steps 3] [--threads 8]

Two figures, both scaled linearly to 100 reverse-diffusion steps (every step costs the same: the reference rebuilds the graph
and the 468-d edge features per step):
  no_grad    -- sampling() under torch.no_grad(): the stricter baseline, what the >= 50x target is judged against
  as_shipped -- sampling() as eval_diffusion.py:62 calls it, autograd recording on (the parameters require grad)
"""
import argparse
import datetime
import json
import os
import platform
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import refshim  # noqa: E402
from make_golden import WEIGHT_SEED, ref_batch  # noqa: E402
from packppi_amd.featurize import protein_to_batch  # noqa: E402
from packppi_amd.pdb_io import from_pdb_file  # noqa: E402
from packppi_amd.weights import make_random_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-grad-steps", type=int, default=10)
    ap.add_argument("--grad-steps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    args = ap.parse_args()
    torch.set_num_threads(args.threads)
    model = refshim.build_reference_module(0)
    model.load_state_dict(make_random_state_dict(WEIGHT_SEED), strict=True)
    model.eval()
    b = protein_to_batch(from_pdb_file(os.path.join(refshim.REF, "data", "T1124_lig.pdb")))
    rb = ref_batch(b)
    res = int(b.residue_mask.sum())
    g4 = np.load(os.path.join(ROOT, "tests", "golden", "g4_T1124.npz"))
    init = torch.from_numpy(g4["init_chi_seed1124"])
    model.add_sc_noise = lambda batch, t: (init.clone(), None)

    def run(n, grad):
        model.schedule = torch.linspace(1, 0, 101)[: n + 1]           # the first n of the 100 steps
        with torch.set_grad_enabled(grad):
            t0 = time.perf_counter()
            model.sampling(rb)
            return time.perf_counter() - t0

    run(1, False)                                                       # warm-up
    t_ng = run(args.no_grad_steps, False)
    t_g = run(args.grad_steps, True)
    cpu = ""
    try:
        cpu = [ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")][0]
    except (OSError, IndexError):
        pass
    out = {"kind": "reference", "what": "Jackz915/PackPPI TDiffusionModule.sampling, unmodified, --device cpu, data/T1124_lig.pdb "
                                        f"({res} residues), seeded xavier weights",
           "host": platform.node(), "cpu": cpu, "cores": args.threads, "torch": torch.__version__,
           "date": datetime.date.today().isoformat(), "unit": "residues/s",
           "no_grad": {"value": res / (t_ng / args.no_grad_steps * 100),
                       "sample": f"{args.no_grad_steps} of 100 steps, {t_ng:.1f} s measured, scaled x{100 / args.no_grad_steps:g}"},
           "as_shipped": {"value": res / (t_g / args.grad_steps * 100),
                          "sample": f"{args.grad_steps} of 100 steps with autograd recording on (eval_diffusion.py:62), "
                                    f"{t_g:.1f} s measured, scaled x{100 / args.grad_steps:g}"}}
    path = os.path.join(ROOT, "profiles", "reference_cpu_rates.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
