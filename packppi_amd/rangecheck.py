"""f16 operand range check of a checkpoint (no reference counterpart).

The default kernels run every dense layer on two-way f16 splits of fp32 operands: accurate to fp32
level as long as every operand stays inside the f16 range; hidden activations beyond 65504 saturate.  The seeded fixtures
are orders of magnitude inside; a trained checkpoint is checked, not trusted:

    python -m packppi_amd.rangecheck --ckpt_path model.ckpt --input complex.pdb      (or --length 300 for a synthetic complex)

runs the score network (and a few sampling steps) through ``libpackppi_hip.chk.so`` -- the same kernels built with
``-DPP_CHECK_RANGE``, which count every value at or beyond the limit that was about to be split -- and prints the number of
events, separately for the edge-level kernels and the node-level kernels.  0 means the split-f16 library is safe for this
checkpoint on this input.  Any other count: ``PACKPPI_LIB=.../libpackppi_hip.f32.so`` is the same ABI with no f16 operand
anywhere (fp32 MFMA edge kernels, the node update in fp32 on the VALU; about 2.2x the time per sampling step at 1 124
residues) and computes such a checkpoint like the fp32 reference does (tests/test_hip_parity.py::
test_checkpoint_outside_the_f16_range).  The check library is loaded in a child process (one library per process).  (The default library also keeps a sticky flag per context when a hidden
activation is clamped: ``Context.saturated()`` / ``pp_ctx_saturated``.)
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys

# LayerNorms whose OUTPUT is an f16-split operand of the edge kernels (h_E0, x1 = LN2 output, h_E = LN3 output): a gain vector
# far below 1 makes those operand vectors small, and below 2^-4 the unscaled low part of the split loses resolution (DESIGN.md
# section 4.1).  ReLU chains are rebalanced when the plan is made, and since round 5 so are these operands: the plan multiplies a
# small (or huge) operand feature by a power of two before the split and divides the consuming weight column by it
# (csrc/pp_rebalance.h ln_operand_scales; `ln_scaled_features` in the report).  `small_gain_layernorms` names the LayerNorms
# concerned; `uncovered_small_operands` lists what the scales could NOT bring within [2^-4, 2^4] (a consuming column that would
# leave the f16 range): only that makes the check fail.
SMALL_GAIN = 1.0 / 16.0
EDGE_OPERAND_NORMS = ["encoder.norm_edges.weight"] + [f"mpnn.mpnn_layers.{l}.norm.{k}.weight" for l in range(3) for k in (2, 3)]
SCALE_ROWS = {"encoder.norm_edges": 0, "mpnn.mpnn_layers.0.norm.3": 1, "mpnn.mpnn_layers.1.norm.3": 2,
              "mpnn.mpnn_layers.0.norm.2": 3, "mpnn.mpnn_layers.1.norm.2": 4}


def uncovered_small_operands(state_dict):
    """[(LayerNorm, features still below 2^-4 after the plan's operand scales)] -- host only (lib.ln_operand_scales)."""
    import torch
    from .lib import ln_operand_scales
    sc, _ = ln_operand_scales(state_dict)
    out = []
    for site, row in SCALE_ROWS.items():
        mag = torch.sqrt(torch.as_tensor(state_dict[site + ".weight"]).float() ** 2 + torch.as_tensor(state_dict[site + ".bias"]).float() ** 2) * sc[row]
        bad = int(((mag > 0) & (mag < SMALL_GAIN)).sum())
        if bad:
            out.append((site, bad))
    return out


def small_gain_layernorms(state_dict):
    """[(name, median |gain|)] of the LayerNorms in front of edge-kernel operands whose median gain is below 2^-4."""
    import torch
    out = []
    for name in EDGE_OPERAND_NORMS:
        if name in state_dict:
            med = float(torch.as_tensor(state_dict[name]).float().abs().median())
            if med < SMALL_GAIN:
                out.append((name, med))
    return out


def _run(args):
    import torch
    from . import lib, synth
    from .featurize import protein_to_batch
    from .module import TDiffusionModule
    from .weights import make_random_state_dict
    L = lib.load()
    if not L.pp_has_range_check():
        raise RuntimeError("this library was built without -DPP_CHECK_RANGE")
    dev = args.device
    if args.ckpt_path:
        model = TDiffusionModule.load_from_checkpoint(args.ckpt_path, map_location=dev, strict=False)
    else:
        sd = make_random_state_dict(args.seed)
        for spec in args.scale or []:                      # name=factor: testing aid
            name, f = spec.split("=")
            sd[name] = sd[name] * float(f)
        for spec in args.outlier or []:                    # name=factor: the first eight ROWS of a weight scaled (testing aid: a few
            name, f = spec.split("=")                      # huge rows are what the plan's rebalancing does NOT remove)
            sd[name] = sd[name].clone()
            sd[name][:8] *= float(f)
        model = TDiffusionModule(sd, device=dev)
    if args.input:
        from .pdb_io import from_pdb_file
        batch = protein_to_batch(from_pdb_file(args.input))
    else:
        batch = protein_to_batch(synth.make_complex(args.length, 5))
    batch = batch.to(dev)
    n = batch.X.shape[1]
    ev_e, ev_n = C.c_ulonglong(0), C.c_ulonglong(0)

    def take():
        assert L.pp_range_check_parts(C.byref(ev_e), C.byref(ev_n), 1) == 0
        return int(ev_e.value), int(ev_n.value)

    take()
    g = torch.Generator().manual_seed(0)
    chi = ((torch.rand(1, n, 4, generator=g) * 2 - 1) * torch.pi).to(dev) * batch.SC_D_mask
    report, edge, node = {}, 0, 0
    for t in (1.0, 0.5, 0.02):
        model.network(batch, chi, torch.full((n,), t, device=dev))
        e, nd = take()
        report[f"network t={t}"] = e + nd
        edge, node = edge + e, node + nd
    model.schedule = torch.linspace(1.0, 0.0, args.steps + 1)
    model.sampling(batch)
    e, nd = take()
    report[f"sampling {args.steps} steps"] = e + nd
    edge, node = edge + e, node + nd
    report["edge_kernels"], report["node_kernels"] = edge, node
    report["total"] = edge + node
    report["sticky_flag"] = model._ctx.saturated() if model._ctx is not None else 0
    report["rebalanced_relu_chains"] = model._plan.rebalanced_chains()
    report["small_gain_layernorms"] = small_gain_layernorms(model.state_dict())
    report["ln_scaled_features"] = model._plan.ln_scaled_features()
    report["uncovered_small_operands"] = uncovered_small_operands(model.state_dict())
    print("RANGECHECK " + json.dumps(report))
    return report


def check(argv):
    """Run the check in a child process on the chk library; returns the report dict."""
    from .build import check_variant_path
    path = check_variant_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: build it with `python -c 'from packppi_amd.build import build_check_variant as b; b()'`")
    env = dict(os.environ, PACKPPI_LIB=path, PACKPPI_RANGECHECK_CHILD="1")
    r = subprocess.run([sys.executable, "-m", "packppi_amd.rangecheck", *argv], env=env, capture_output=True, text=True)
    for line in r.stdout.splitlines():
        if line.startswith("RANGECHECK "):
            return json.loads(line[len("RANGECHECK "):])
    raise RuntimeError("range check failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])


def main(argv=None):
    p = argparse.ArgumentParser(description="f16 operand range check of a checkpoint on one complex")
    p.add_argument("--ckpt_path", type=str, default=None, help="Lightning checkpoint (else seeded stand-in weights).")
    p.add_argument("--input", type=str, default=None, help="PDB file (else a synthetic complex of --length residues).")
    p.add_argument("--length", type=int, default=300)
    p.add_argument("--steps", type=int, default=10, help="sampling steps run after the three network evaluations")
    p.add_argument("--seed", type=int, default=20251003)
    p.add_argument("--scale", action="append", help="NAME=FACTOR: scale one seeded weight (testing aid)")
    p.add_argument("--outlier", action="append", help="NAME=FACTOR: scale the first eight rows of one seeded weight (testing aid)")
    p.add_argument("--device", type=str, default="cuda:0")
    argv = sys.argv[1:] if argv is None else argv
    args = p.parse_args(argv)
    if os.environ.get("PACKPPI_RANGECHECK_CHILD"):
        _run(args)
        return 0
    rep = check(argv)
    for k, v in rep.items():
        print(f"{k:24s} {v}")
    if rep.get("small_gain_layernorms"):
        print("NOTE: LayerNorm gains with a median below 2^-4 in front of split-f16 operands: " + ", ".join(f"{n} ({m:.3g})" for n, m in rep["small_gain_layernorms"])
              + f" -- the plan carries power-of-two operand scales for {rep.get('ln_scaled_features', 0)} features (exact; csrc/pp_rebalance.h)")
    uncovered = rep.get("uncovered_small_operands") or []
    if uncovered:
        print("f16 operand RESOLUTION not covered: " + ", ".join(f"{n}: {k} features" for n, k in uncovered) + " stay below 2^-4 after the "
              "operand scales (a consuming weight column would leave the f16 range): run this checkpoint with PACKPPI_LIB=.../libpackppi_hip.f32.so")
        return 3
    if rep["total"] == 0:
        print("f16 operand range: OK")
    else:
        where = "edge kernels" if rep["node_kernels"] == 0 else "node kernels" if rep["edge_kernels"] == 0 else "edge and node kernels"
        print(f"f16 operand range EXCEEDED in the {where}: run this checkpoint with PACKPPI_LIB=.../libpackppi_hip.f32.so "
              "(every dense layer in fp32, same ABI, about 2.2x the time per step)")
    return 0 if rep["total"] == 0 else 2


if __name__ == "__main__":
    sys.exit(main())
