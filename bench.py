#!/usr/bin/env python
"""Benchmark of the PackPPI-MSC sampling path on MI355X (contract: see the task brief / DESIGN.md §Measurement).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full ``sampling()`` pass (100 reverse-diffusion network evaluations, no proximal) over the
rank's batch.  Workload at every N: ``data/T1124_lig.pdb`` (738 true residues, fixture
tests/golden/g4_T1124.npz) per GPU -- BASELINE.json configs[1]; ranks hold independent complexes (weak scaling,
no data-path collective); the only collective is the all-gather of per-complex metric rows.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_DIFFUSION_STEPS = 100
# algorithmic FLOPs (2/MAC, dense projections only) of ONE launch of the dominant kernel (edge update), per edge:
# edge message MLP 456->128->128->128 + FFN 128->512->128   (SURVEY.md §8d itemisation)
EDGE_UPDATE_FLOP_PER_EDGE = 2 * (456 * 128 + 128 * 128 + 128 * 128) + 2 * (128 * 512 + 512 * 128)
NODE_MSG_FLOP_PER_EDGE = 2 * (456 * 128 + 128 * 128 + 128 * 128)
FP32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"


def load_t1124():
    from packppi_amd.batch import Batch
    z = np.load(os.path.join(ROOT, "tests", "golden", "g4_T1124.npz"))
    b = Batch()
    for k in z.files:
        if k.startswith("batch."):
            key = k[6:]
            b[key] = int(z[k]) if key in ("num_proteins", "max_size") else torch.from_numpy(z[k])
    return b, torch.from_numpy(z["init_chi_seed1124"]), torch.from_numpy(z["chi_ode_100"])


def load_s1500():
    """The 1500-residue synthetic complex with the reference's own 100-step output on the same noise, if the fixture
    is present (tests/golden/g5_S1500.npz, tools/oracle/make_golden.py --only g5)."""
    from packppi_amd.batch import Batch
    path = os.path.join(ROOT, "tests", "golden", "g5_S1500.npz")
    if not os.path.exists(path):
        return None
    z = np.load(path)
    b = Batch()
    for k in z.files:
        if k.startswith("batch."):
            key = k[6:]
            b[key] = int(z[k]) if key in ("num_proteins", "max_size") else torch.from_numpy(z[k])
    return b, torch.from_numpy(z["init_chi_seed1500"]), torch.from_numpy(z["chi_ode_100"])


def synth_workload(kind, rank):
    from packppi_amd import synth
    from packppi_amd.batch import collate
    from packppi_amd.featurize import protein_to_batch, protein_to_data
    if kind == "s1500":
        return protein_to_batch(synth.make_complex(1500, 1500))
    lens = synth.c5_lengths(256)
    mine = list(range(rank * 32, rank * 32 + 32))
    return collate([protein_to_data(synth.make_complex(lens[i], 10000 + i)) for i in mine])


def cpu_baseline(batch, init, weights, n_sample_steps):
    """Oracle (CPU port of the reference algorithm, recomputing the graph every step like the reference) on a
    bounded sample of the same workload; extrapolated linearly to 100 diffusion steps."""
    from oracle import ref_cpu as O
    # the GPU box gives one GPU a 16-core CPU share; torch's default (all 128 hardware threads) oversubscribes
    # these small ops and is ~6x slower
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sched = torch.linspace(1, 0, N_DIFFUSION_STEPS + 1)[: n_sample_steps + 1]
    with torch.no_grad():
        O.sampling(weights, batch, init, sched[:2], hoist=False)          # warm-up
        t0 = time.perf_counter()
        O.sampling(weights, batch, init, sched, hoist=False)
        dt = time.perf_counter() - t0
    res = batch.true_residues()
    per100 = dt / n_sample_steps * N_DIFFUSION_STEPS
    return {"value": res / per100, "unit": "residues/s", "cores": cores, "kind": "port",
            "sample": f"{n_sample_steps} of {N_DIFFUSION_STEPS} diffusion steps of the same complex under "
                      f"torch.no_grad, graph recomputed per step as the reference does; {dt:.1f} s measured, "
                      f"scaled x{N_DIFFUSION_STEPS / n_sample_steps:g}"}


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this same command (tools/profile/run_profiles.sh;
    counters cannot be read from inside the process).  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))["kernels"][kernel]["hbm_bytes"]
    except (KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="t1124", choices=["t1124", "s1500", "c5"])
    ap.add_argument("--proximal", action="store_true", help="add the 50-step proximal optimisation (configs[2])")
    ap.add_argument("--cpu-steps", type=int, default=20, help="diffusion steps of the CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")       # "nccl" IS RCCL on ROCm; gloo only for rehearsals
        local_rank %= max(torch.cuda.device_count(), 1)                # rehearsal of N ranks on fewer GPUs
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))   # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from packppi_amd.module import TDiffusionModule
    from packppi_amd.weights import make_random_state_dict
    weights = make_random_state_dict(20251003)
    ref_chi = None
    if args.workload == "t1124":
        batch, init, ref_chi = load_t1124()
        name = "data/T1124_lig.pdb (L=739, 738 true residues), 1 complex per GPU"
    else:
        batch = synth_workload(args.workload, rank)
        init = None
        if args.workload == "s1500" and load_s1500() is not None:      # same complex, with the reference's output
            batch, init, ref_chi = load_s1500()
        name = {"s1500": "synthetic 1500-residue 2-chain complex (default_rng(1500)), 1 per GPU",
                "c5": "32 synthetic complexes L~U{270..330} per GPU (default_rng(256))"}[args.workload]
    residues = batch.true_residues()
    model = TDiffusionModule(weights, device=dev)
    model.schedule = torch.linspace(1, 0, N_DIFFUSION_STEPS + 1)
    gb = batch.to(dev)
    if init is None:
        torch.manual_seed(1000 + rank)
        init_d, _ = model.add_sc_noise(gb, torch.ones(batch.residue_type.numel(), device=dev))
        init = init_d.cpu()
    else:
        init_d = init.to(dev)
    ctx = model._context(gb)

    # One timed pass = what TDiffusionModule.sampling() costs on a batch it has not seen: the per-complex preparation
    # (kNN graph, frames, edge embedding, layer-0 static products: pp_complex_prepare, a fresh context) + 100 evaluations.
    from packppi_amd.lib import Context

    def one_pass():
        chi = Context(model._plan, gb).sample(init_d, model.schedule)
        if args.proximal:
            from packppi_amd.functional import proximal_optimizer
            chis, losses = proximal_optimizer(gb, chi, 12.0, 0.5, 1.0, 50)
            chi = chis[-1] if losses[-1] < losses[0] else chi
        return chi

    for _ in range(args.warmup):
        chi = one_pass()

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        chi = one_pass()
    fence()
    elapsed = time.perf_counter() - t0
    total_res = residues
    if dist is not None:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([residues], device=dev, dtype=torch.float64)
        dist.all_reduce(rr, op=dist.ReduceOp.SUM)
        total_res = int(rr.item())

    # per-complex metric row (the one real collective of the path: RCCL all-gather of metric rows)
    m = model.analyze_samples(gb, chi)
    row = torch.stack([torch.as_tensor(float(v), device=dev) for v in m.values()]).float()
    rows = [row]
    if dist is not None:
        rows = [torch.empty_like(row) for _ in range(world)]
        dist.all_gather(rows, row)
    max_dchi = None
    if ref_chi is not None and not args.proximal:
        d = (chi.cpu().double() - ref_chi.double()).abs()
        d = torch.minimum(d, (2 * np.pi - d).abs())[batch.SC_D_mask.bool()]
        max_dchi = float(d.max())

    # dominant-kernel roofline, measured live: one more pass of the same workload with every launch of the kernel
    # bracketed by HIP events on the launch stream (pp_profile_kernel).  Back-to-back launches of the edge kernel alone
    # (pp_time_kernel) run ~10 % slower than in situ (sustained-MFMA clocks), so they are reported only as a cross-check.
    insitu = {}
    for which, kname in ((1, "k_edge_update"), (0, "k_node_message"), (2, "k_node_update")):
        ctx.profile_kernel(which)
        ctx.sample(init_d, model.schedule)
        insitu[kname] = ctx.profile_read()
    t_edge = insitu["k_edge_update"][0] * 1e-3
    t_node = insitu["k_node_message"][0] * 1e-3
    t_edge_b2b = ctx.time_kernel(1, 20) * 1e-3
    n_edges = residues * ctx.K
    # k_edge_update(l) also computes the node message of layer l + 1 (fused): its algorithmic work is both MLP chains
    # of the reference (layers.py:119-148), 2 FLOP per MAC of the dense layers, per edge.  Executed MFMA work is lower:
    # layer 0's W_B h_E0 products are timestep-invariant and computed once per complex.
    fused_flop_per_edge = EDGE_UPDATE_FLOP_PER_EDGE + NODE_MSG_FLOP_PER_EDGE
    achieved = fused_flop_per_edge * n_edges / t_edge / 1e12
    # which edge kernels the library was built with: 1 = split-f16 (default), 0 = exact fp32 (PACKPPI_EDGE=f32)
    from packppi_amd import lib as _lib
    split_f16 = _lib.load().pp_edge_variant() == 1
    if split_f16:
        # 342 (layer 1) / 318 (layer 0) v_mfma_f32_32x32x16_f16 per residue and launch, 32768 FLOP each: every fp32
        # product is three f16 products (hi hi + hi lo + lo hi)
        executed_mfma = 0.5 * (342 + 318) * 32768.0 * residues
        peak, dtype = F16_MFMA_PEAK_TFLOPS, "f32 (dense layers as split-f16: two f16 per operand, three f16 MFMAs per product, fp32 accumulate)"
    else:
        executed_mfma = (2960 + 656 - 128) * 4096.0 * residues    # average of the layer-0 and layer-1 launches
        peak, dtype = FP32_MFMA_PEAK_TFLOPS, "f32"

    if rank == 0:
        out = {
            "metric": "sampled residues/sec at 100 diffusion steps",
            "value": total_res * args.steps / elapsed,
            "unit": "residues/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic (seeded random weights; T1124 backbone fixture, seeded initial noise)"
            if args.workload == "t1124" else "synthetic",
            "config": {"workload": name, "diffusion_steps": N_DIFFUSION_STEPS, "proximal": bool(args.proximal),
                       "residues_per_gpu": residues, "mode": "ode"},
            "roofline": {"bound": "mfma", "kernel": "k_edge_update", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                         "peak_is": "dense F16 MFMA (the pipe the kernel runs on)" if split_f16 else "FP32 matrix",
                         "achieved_is": "the reference's fp32 dense-layer arithmetic (2 FLOP per MAC) per second; the kernel "
                                        "issues 3 f16 MFMAs per product, see executed_mfma_tflops",
                         "achieved_over_fp32_matrix_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "edge_kernels": "split-f16" if split_f16 else "fp32",
                         "traffic": pmc_traffic("k_edge_update") if args.workload == "t1124" else None,
                         "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc "
                                         "passes of this command; profiles/*_pmc_traffic.json)",
                         "kernel_ms": t_edge * 1e3,
                         "kernel_does": "edge update of layer l + node message of layer l+1, one launch",
                         "algorithmic_flop_per_launch": fused_flop_per_edge * n_edges,
                         "executed_mfma_tflops": executed_mfma / t_edge / 1e12,
                         # whole pass against SURVEY 8(d): 46 792 576 algorithmic FLOP per residue per network evaluation
                         "whole_pass_algorithmic_tflops": 46792576.0 * total_res * N_DIFFUSION_STEPS * args.steps / elapsed / 1e12
                                                          / max(args.gpus, 1),
                         "whole_pass_frac_of_peak": 46792576.0 * total_res * N_DIFFUSION_STEPS * args.steps / elapsed / 1e12
                                                    / max(args.gpus, 1) / FP32_MFMA_PEAK_TFLOPS,
                         "kernel_launches_timed": insitu["k_edge_update"][1],
                         "kernel_ms_back_to_back": t_edge_b2b * 1e3,
                         "node_update_kernel_ms": insitu["k_node_update"][0],
                         "node_message_kernel_ms": t_node * 1e3,
                         "node_message_layer0_tflops": NODE_MSG_FLOP_PER_EDGE * n_edges / t_node / 1e12},
            "parity": {"max_abs_dchi_vs_reference_rad": max_dchi, "atom_rmsd": float(m["atom_rmsd"])},
            "metrics_rows_gathered": len(rows),
        }
        if args.cpu_steps > 0 and args.gpus == 1:
            out["cpu_baseline"] = cpu_baseline(batch, init, weights, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
