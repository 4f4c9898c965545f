#!/usr/bin/env python
"""Benchmark of the PackPPI-MSC sampling path on MI355X (contract: see the task brief / DESIGN.md section 5).

    python bench.py --gpus N --steps K --warmup W

One "step" = one full ``sampling()`` pass (100 reverse-diffusion network evaluations, no proximal) over the
rank's batch.
  N = 1 (default workload "t1124"): ``data/T1124_lig.pdb`` (738 true residues, fixture tests/golden/g4_T1124.npz) --
        BASELINE.json configs[1], the configuration the metric is quoted on.  `secondary` carries every other BASELINE
        config, including ALL 256 complexes of configs[4] on this one GPU (the anchor of the strong-scaling curve).
  N > 1 (default workload "c5"): BASELINE.json configs[4] as stated -- the 256 synthetic ~300-residue complexes dealt to
        the N ranks by parallel.shard_complexes, every rank runs parallel.sample_sharded on its share (packed ragged
        batches, per-complex metrics) and the 256 metric rows are all-gathered INSIDE the timed pass (RCCL when the
        backend is "nccl"): total work is fixed, `scaling` = "strong".  No data-path collective.  A single complex does not
        shard ("replicas only"): one T1124 replica per rank is reported as `secondary`.
Prints ONE JSON line on rank 0; its LAST key, `summary`, is a flat digest of every config and regime (survives a tail).

N > 1: the driver launches one rank per GPU with torch.distributed.run; run by hand without WORLD_SIZE, this script
starts that launcher itself as a child process BEFORE anything touches the GPU and exits with its code.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import packppi_amd  # noqa: E402,F401  (sets HIP_FORCE_DEV_KERNARG before the HIP runtime initialises)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_DIFFUSION_STEPS = 100
# algorithmic FLOPs (2/MAC, dense projections only) of ONE launch of the dominant kernel (edge update), per edge:
# edge message MLP 456->128->128->128 + FFN 128->512->128   (SURVEY.md section 8d itemisation)
EDGE_UPDATE_FLOP_PER_EDGE = 2 * (456 * 128 + 128 * 128 + 128 * 128) + 2 * (128 * 512 + 512 * 128)
NODE_MSG_FLOP_PER_EDGE = 2 * (456 * 128 + 128 * 128 + 128 * 128)
# node update, per residue and launch (layers 0, 1): W_out + FFN + the four 128x128 message-input projections + 2 x 24 points
NODE_UPDATE_FLOP_PER_RES = 2 * (128 * 128 + 2 * 128 * 512 + 4 * 128 * 128 + 2 * 24 * 128)
NODE_UPDATE_STREAM_BYTES = 56 * 8 * 2048       # one workgroup's weight stream (pp_internal.h PP_NU_SLOTS_MID slots x 8 waves x 2 KB)
FP32_MFMA_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
F16_MFMA_PEAK_TFLOPS = 2500.0       # MI355X_MICROARCH.md, "Peak BF16/FP16 MFMA ~2.5 PF dense"
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md, "HBM3E peak BW 8.0 TB/s spec"
L2_PEAK_TBS = 34.5                  # MI355X_MICROARCH.md, "L2 (per XCD)": 4 MiB per XCD, ~34.5 TB/s aggregate
CU_VMEM_PEAK_GBS = 64 * 2.4         # one CU's vector-memory path: 64 B/clk at 2.4 GHz
PROFILE_TAG = "r05_v2"             # the committed rocprofv3 summaries of THIS command (profiles/<tag>_*.{csv,json})


def _load_fixture(name, init_key):
    from packppi_amd.batch import Batch
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    b = Batch()
    for k in z.files:
        if k.startswith("batch."):
            key = k[6:]
            b[key] = int(z[k]) if key in ("num_proteins", "max_size") else torch.from_numpy(z[k])
    return b, torch.from_numpy(z[init_key]), torch.from_numpy(z["chi_ode_100"])


def load_t1124():
    return _load_fixture("g4_T1124", "init_chi_seed1124")


def load_s1500():
    """The 1500-residue synthetic complex with the reference's own 100-step output on the same noise
    (tests/golden/g5_S1500.npz, tools/oracle/make_golden.py --only g5)."""
    return _load_fixture("g5_S1500", "init_chi_seed1500")


def c5_proteins(ids, workers):
    """Host side of BASELINE config 4: the protein dicts of complexes `ids` (numpy, synth.make_complex(L_i, 10000 + i)), built on
    `workers` forked processes.  Called before this process touches the GPU."""
    from packppi_amd import synth
    lens = synth.c5_lengths(256)
    ids = list(ids)
    return dict(zip(ids, synth.make_complexes([(lens[i], 10000 + i) for i in ids], workers)))


def c5_share(rank, world, dev, proteins=None):
    """BASELINE config 4: the 256 synthetic complexes (L ~ U{270..330}, default_rng(256), seeds 10000 + i) dealt to `world`
    ranks by parallel.shard_complexes (longest first); returns (lengths of all 256, {complex id: batch} of THIS rank's share).
    `proteins`: {id: protein dict} built earlier (c5_proteins); what is missing is built here."""
    from packppi_amd import synth
    from packppi_amd.featurize import protein_to_batch
    from packppi_amd.parallel import shard_complexes
    lens = synth.c5_lengths(256)
    mine = shard_complexes(lens, world)[rank]
    proteins = proteins or {}
    return lens, {i: protein_to_batch(proteins[i] if i in proteins else synth.make_complex(lens[i], 10000 + i)).to(dev) for i in mine}


def c5_complexes(rank, dev):
    """One GPU's share of config 4 when it runs on 8 GPUs (32 complexes), as a list (tools/, tests)."""
    return list(c5_share(rank % 8, 8, dev)[1].values())


def c5_inits(share, seed):
    g = torch.Generator().manual_seed(seed)
    return {i: (torch.rand(1, int(c["max_size"]), 4, generator=g) * 2 - 1) * np.pi * c.SC_D_mask.cpu() for i, c in share.items()}


def reference_container_rates():
    """The UNMODIFIED reference timed in the build container (tools/oracle/time_reference.py; the reference cannot travel to the
    GPU box): its rate under torch.no_grad and as eval_diffusion.py:62 calls it, with host, cores and date.  None if absent."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "reference_cpu_rates.json")))
    except (OSError, ValueError):
        return None


def full_protocol_record():
    """BASELINE.md section 3 by the letter, measured once on a GPU box (`python bench.py --cpu-protocol full`, ~2.5 minutes of host time)
    and committed: profiles/<round>_cpu_baseline_full.json.  None if absent."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r05_cpu_baseline_full.json")))
    except (OSError, ValueError):
        return None


def cpu_baseline_full(batch, init, weights, cores):
    """BASELINE.md section 3 by the letter: the median of 5 FULL 100-step sampling() calls of the oracle after 2 full warm-up calls,
    under torch.no_grad, graph recomputed per step as the reference does."""
    from oracle import ref_cpu as O
    torch.set_num_threads(cores)
    sched = torch.linspace(1, 0, N_DIFFUSION_STEPS + 1)
    res = batch.true_residues()
    secs = []
    with torch.no_grad():
        for i in range(7):
            t0 = time.perf_counter()
            O.sampling(weights, batch, init, sched, hoist=False)
            dt = time.perf_counter() - t0
            if i >= 2:
                secs.append(dt)
            print(f"cpu full protocol: call {i + 1}/7 {dt:.1f} s", file=sys.stderr, flush=True)
    secs.sort()
    return {"value": res / secs[2], "unit": "residues/s", "cores": cores, "kind": "port", "min": res / secs[-1], "max": res / secs[0],
            "seconds_per_call": secs, "host_cpu_count": os.cpu_count(),
            "sample": f"median of 5 full {N_DIFFUSION_STEPS}-step sampling() calls after 2 full warm-up calls (BASELINE.md section 3), "
                      f"torch.no_grad, graph recomputed per step, {cores} threads"}


def cpu_baseline(batch, init, weights, n_sample_steps, n_grad_steps):
    """Oracle (CPU port of the reference algorithm, recomputing the graph every step like the reference) on a
    bounded sample of the same workload; extrapolated linearly to 100 diffusion steps.  `value` is the torch.no_grad
    figure (what the >= 50x target is judged against); `as_shipped` repeats it with autograd recording, which
    is how eval_diffusion.py:62 calls sampling()."""
    from oracle import ref_cpu as O
    # the GPU box gives one GPU a 16-core CPU share; torch's default (all 128 hardware threads) oversubscribes
    # these small ops and is ~6x slower
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    sched = torch.linspace(1, 0, N_DIFFUSION_STEPS + 1)
    res = batch.true_residues()
    # BASELINE.md section 3 asks for the median of >= 5 full sampling() calls after 2 warm-ups; a full 100-step call costs ~22 s
    # here, so the bounded form is the median of 5 SAMPLES of n_sample_steps / 5 consecutive steps each (every step of this
    # loop costs the same: the graph is recomputed per step), after 2 one-step warm-ups -- stated in `sample`
    n_rep = 5
    per = max(n_sample_steps // n_rep, 1)
    rates, total_s = [], 0.0
    with torch.no_grad():
        for _ in range(2):
            O.sampling(weights, batch, init, sched[:2], hoist=False)          # warm-ups
        x = init
        for r in range(n_rep):
            t0 = time.perf_counter()
            x = O.sampling(weights, batch, x, sched[r * per: (r + 1) * per + 1], hoist=False)
            dt = time.perf_counter() - t0
            total_s += dt
            rates.append(res / (dt / per * N_DIFFUSION_STEPS))
    rates.sort()
    out = {"value": rates[n_rep // 2], "unit": "residues/s", "cores": cores, "kind": "port",
           "min": rates[0], "max": rates[-1],
           "sample": f"median of {n_rep} samples of {per} consecutive diffusion steps each ({n_rep * per} of {N_DIFFUSION_STEPS} steps "
                     f"of the same complex) under torch.no_grad after 2 warm-ups, graph recomputed per step as the reference does, "
                     f"{cores} threads = this box's CPU share for one GPU; {total_s:.1f} s measured, each sample scaled "
                     f"x{N_DIFFUSION_STEPS / per:g}.  Deviation from BASELINE.md section 3 (median of >= 5 FULL 100-step calls, ~110 s): "
                     f"bounded to keep the default run within minutes"}
    if n_grad_steps > 0:
        wg = {k: v.clone().requires_grad_(True) for k, v in weights.items()}       # nn.Parameters, no torch.no_grad
        x = init
        t0 = time.perf_counter()
        for j in range(n_grad_steps):
            x = O.sampling(wg, batch, x, sched[j: j + 2], hoist=False).detach()   # the reference's step() is @no_grad
        dtg = time.perf_counter() - t0
        out["as_shipped"] = {"value": res / (dtg / n_grad_steps * N_DIFFUSION_STEPS), "unit": "residues/s", "kind": "port",
                             "sample": f"{n_grad_steps} steps with autograd recording (eval_diffusion.py:62 calls sampling() "
                                       f"without torch.no_grad); {dtg:.1f} s measured, scaled x{N_DIFFUSION_STEPS / n_grad_steps:g}",
                             "note": "autograd recording costs this loop nothing measurable, in the port and in the reference itself "
                                     "(reference_in_build_container: measured, not quoted)"}
    ref = reference_container_rates()
    if ref is not None:
        out["reference_in_build_container"] = ref      # kind "reference", measured where the reference can run
    full = full_protocol_record()
    if full is not None:
        out["full_protocol"] = full                    # BASELINE.md section 3 by the letter, measured once on a GPU box (committed)
    return out


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC summary of this same command (tools/profile/run_profiles.sh;
    counters cannot be read from inside the process): profiles/<PROFILE_TAG>_pmc_traffic.json.  None when it is absent."""
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_pmc_traffic.json")
    try:
        ks = json.load(open(path))["kernels"]
    except (OSError, KeyError, ValueError):
        return None
    # the launch of this kernel the timed passes use (k_edge_update_mix at T1124): the entry with the most launches
    hits = [v for k, v in ks.items() if k.startswith(kernel)]
    return max(hits, key=lambda v: v["launches"])["hbm_bytes"] if hits else None


def l2_requests(kernel):
    """128-byte L2 requests per launch of `kernel` from the committed TCC counter pass of this command
    (tools/profile/run_tcc.sh -> profiles/<PROFILE_TAG>_tcc_t1124.json).  None when it is absent."""
    try:
        ks = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_tcc_t1124.json")))["kernels"]
    except (OSError, KeyError, ValueError):
        return None
    hits = [v for k, v in ks.items() if k.startswith(kernel)]
    return max(hits, key=lambda v: v["launches"]) if hits else None


RENDEZVOUS_ENV = ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID", "GROUP_RANK",
                  "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE")


def spawn_ranks(n):
    """python bench.py --gpus N by hand: start N ranks (one per GPU) through torch.distributed.run as a CHILD process.
    Nothing in this process has touched the GPU yet (import torch does not), so no initialised process is replaced."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def regime_counters():
    """MFMA-busy share and L2 request bytes per edge-update launch in the three regimes (T1124: one round; S1500: 1.5 rounds; the
    configs[4] share: 38 residues per CU) from the committed counter passes of this command (tools/profile/regimes.py ->
    profiles/<PROFILE_TAG>_regimes.json; counters cannot be read from inside the process).  {} when the file is absent."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_regimes.json")))["regimes"]
    except (OSError, KeyError, ValueError):
        return {}


def insitu_edge(ctx, init_d, sched):
    """Mean begin-to-end interval (seconds) of the k_edge_update dispatches of one more sampling pass on `ctx`."""
    ctx.profile_kernel(1)
    ctx.sample(init_d, sched)
    ms, n = ctx.profile_read()
    return ms * 1e-3, n


def regime_entry(name, residues, K, t_edge, launches, counters):
    """One regime of the dominant kernel: algorithmic FLOP / live kernel time against the dense F16 pipe + the committed counters."""
    flop = (EDGE_UPDATE_FLOP_PER_EDGE + NODE_MSG_FLOP_PER_EDGE) * residues * K
    alg_bytes = 2.0 * residues * K * 512              # h_E read once + written once, fp32
    c = counters.get(name, {})
    e = {"residues": residues, "kernel_ms": t_edge * 1e3, "launches_timed": launches,
         "achieved_tflops": flop / t_edge / 1e12, "frac": flop / t_edge / 1e12 / F16_MFMA_PEAK_TFLOPS,
         "algorithmic_hbm_bytes_per_launch": alg_bytes, "hbm_frac": alg_bytes / t_edge / 1e9 / HBM_PEAK_GBS,
         "mfma_busy": c.get("mfma_busy"), "l2_request_bytes_per_launch": c.get("l2_request_bytes"),
         "l2_over_algorithmic": (c["l2_request_bytes"] / alg_bytes) if c.get("l2_request_bytes") else None,
         "kernel": c.get("kernel")}
    return e


def clash_work(gb, chi, tol):
    """What one k_clash launch tests at these angles, recomputed with torch from the same records the kernel culls with
    (pp_clash.hip k_atom14 / k_clash: bounding sphere about the centroid of the atoms present, radius x 1.0001 + 1e-3; a partner
    survives when the spheres come closer than 3.6 - tol A and the residue numbers differ; of a surviving residue pair every
    atom pair except backbone-backbone and CB-CB goes through the squared-distance test).  Host-side bookkeeping of the
    measurement, not part of the path."""
    from packppi_amd.functional import _ctx_for
    xyz = _ctx_for(gb).atom14(chi)[0]                       # [L, 14, 3]
    ex = gb["atom_mask"][0] > 0                            # [L, 14]
    L = xyz.shape[0]
    cnt = ex.sum(1).clamp(min=1)
    cen = (xyz * ex[..., None]).sum(1) / cnt[:, None]
    rad = (((xyz - cen[:, None]) ** 2).sum(-1) * ex).max(1)[0].sqrt() * 1.0001 + 1e-3
    d2 = ((cen[:, None] - cen[None]) ** 2).sum(-1)
    lim = rad[:, None] + rad[None] + (3.6 - tol)
    ri = gb["residue_index"][0]
    keep = (d2 < lim * lim) & (ri[:, None] != ri[None])
    keep.fill_diagonal_(False)
    na = ex.sum(1).double()
    nbb = ex[:, :4].sum(1).double()
    ncb = ex[:, 4].double()
    pairs = (keep.double() * (na[:, None] * na[None] - nbb[:, None] * nbb[None] - ncb[:, None] * ncb[None])).sum()
    return {"sphere_tests": L * (L - 1), "candidate_residue_pairs": int(keep.sum()), "atom_pair_tests": float(pairs),
            "culled_fraction_of_residue_pairs": 1.0 - float(keep.sum()) / (L * (L - 1)),
            "dense_atom_pairs_i_lt_j": L * (L - 1) // 2 * 196}


def proximal_roofline(gb, chi, n_steps=50):
    """The ONE launch of a proximal Adam step (k_clash<*, true>: clash loss + gradient at the current angles, the step on the workgroup's own
    residue, its reconstruction at the new angles), timed in situ (start/stop HIP events on every such launch inside pp_proximal) and
    priced twice: squared-distance atom-pair tests per second against the fp32 VALU peak, and the bytes of the step + reconstruction tail
    against HBM."""
    from packppi_amd.functional import _ctx_for
    ctx = _ctx_for(gb)
    ctx.profile_kernel(3)
    ctx.proximal(chi, 12.0, 0.5, 1.0, n_steps, want_traj=False)
    ms, n = ctx.profile_read()
    out = {"k_clash": {"kernel_us": ms * 1e3, "launches_timed": n,
                       "kernel_does": "clash loss + analytic gradient, Adam step and atom14 reconstruction of the workgroup's residue: one launch per "
                                      "Adam step (rounds 2-4: two, k_clash + k_atom14<true>)"}}
    w = clash_work(gb, chi, 0.5)
    t = out["k_clash"]["kernel_us"] * 1e-6
    res = int(gb["residue_mask"].sum())
    # per tested atom pair: 3 sub, 3 mul/fma, 1 add (d2), 1 add + 1 sub (threshold), 1 mul, 1 compare = 11 FLOP; the few per cent that
    # overlap add sqrt, the hinge and the gradient (~45 FLOP) -- not counted
    out["k_clash"].update(w)
    # tail, per residue: reads X 168 + BB_D 12 + seven Adam operands 112 + tables (L2); writes xyz 168 + records 256 + axes 96 + Adam state / angles 80
    by = res * (168 + 12 + 112 + 168 + 256 + 96 + 80)
    out["k_clash"].update({"atom_pair_tests_per_s": w["atom_pair_tests"] / t, "bound": "valu", "unit": "TFLOP/s",
                           "achieved": 11.0 * w["atom_pair_tests"] / t / 1e12, "peak": FP32_MFMA_PEAK_TFLOPS,
                           "frac": 11.0 * w["atom_pair_tests"] / t / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                           "flop_per_atom_pair_test": 11,
                           "sphere_test_bytes": 32.0 * w["sphere_tests"],
                           "candidate_record_bytes": 240.0 * w["candidate_residue_pairs"],
                           "l2_read_GBs": (32.0 * w["sphere_tests"] + 240.0 * w["candidate_residue_pairs"]) / t / 1e9,
                           "step_and_reconstruction_bytes": by, "hbm_frac": by / t / 1e9 / HBM_PEAK_GBS,
                           "limiter": "latency: candidate list -> partner records -> pair tests -> reductions -> Adam operands and tables -> rigid-group "
                                      "chain -> stores, one dependent chain per workgroup; work and bytes are orders of magnitude below either peak"})
    return out


PROX_HELD_TO = ("end state within 2 x the reference's own fp32<->fp64 distance after 50 Adam steps (tests/test_hip_parity.py::test_proximal); "
                "1e-4 rad is unreachable for 50 Adam steps on hinges: the reference misses it against itself by 24-51x")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=["t1124", "s1500", "c5", "c5share"],
                    help="default: t1124 (BASELINE configs[1]) on one GPU, c5 (configs[4]: the 256 complexes sharded over the ranks) on several; "
                         "c5share = one GPU's share of configs[4] at 8 GPUs as one packed batch (the profiling workload of that regime)")
    ap.add_argument("--proximal", action="store_true", help="add the 50-step proximal optimisation (configs[2] / configs[3])")
    ap.add_argument("--cpu-steps", type=int, default=50, help="diffusion steps of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-grad-steps", type=int, default=4, help="steps of the autograd-on CPU sample (0 = skip)")
    ap.add_argument("--cpu-protocol", default="bounded", choices=["bounded", "full"],
                    help="full: BASELINE.md section 3 by the letter (5 full 100-step calls after 2 full warm-ups, ~2.5 minutes of host time) "
                         "instead of the bounded sample; its committed record rides on every line as cpu_baseline.full_protocol")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = this box's CPU share for one GPU: 16)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the other BASELINE configs (`secondary` list)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the in-situ kernel timing passes (child runs)")
    ap.add_argument("--c5-max-rows", type=int, default=0, help="rows per packed batch of the sharded path (0 = parallel.sample_sharded's default)")
    ap.add_argument("--build-workers", type=int, default=0, help="processes that build the synthetic complexes (0 = this box's CPU share; the "
                    "profiling scripts pass 1: no fork under rocprofv3's preloaded library)")
    ap.add_argument("--diffusion-steps", type=int, default=100, help="network evaluations per pass (the metric is DEFINED at 100; counter passes "
                    "under rocprofv3 --pmc use a handful: per-launch figures are the same)")
    args = ap.parse_args()
    global N_DIFFUSION_STEPS
    N_DIFFUSION_STEPS = args.diffusion_steps

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: one rank per GPU (torch.distributed.run --nproc-per-node {args.gpus})")
    if args.workload is None:
        # the driver's command line carries no --workload: one GPU measures the configuration the metric is quoted on, several GPUs
        # measure the configuration BASELINE.json states for them (a single complex does not shard)
        args.workload = "t1124" if world == 1 else "c5"
    want_secondary = not args.no_secondary and not args.proximal
    # ---- host-side inputs that take a process pool: BEFORE anything touches the GPU (forked workers) -----------------------
    cores = min(16, os.cpu_count() or 1)
    if args.build_workers > 0:
        cores = args.build_workers
    c5_prot = {}
    if args.workload == "c5":
        from packppi_amd import synth
        from packppi_amd.parallel import shard_complexes
        c5_prot = c5_proteins(shard_complexes(synth.c5_lengths(256), world)[rank], max(1, min(cores, (os.cpu_count() or 1) // world)))
    elif args.workload == "c5share":
        from packppi_amd import synth
        from packppi_amd.parallel import shard_complexes
        c5_prot = c5_proteins(shard_complexes(synth.c5_lengths(256), 8)[rank % 8], max(1, min(cores, (os.cpu_count() or 1) // world)))
    elif want_secondary and world == 1 and args.workload == "t1124":
        c5_prot = c5_proteins(range(256), cores)          # the share AND the whole job on this GPU
    elif want_secondary:
        from packppi_amd import synth
        from packppi_amd.parallel import shard_complexes
        c5_prot = c5_proteins(shard_complexes(synth.c5_lengths(256), 8)[rank % 8], max(1, min(cores, (os.cpu_count() or 1) // world)))
    dist = None
    backend = None
    # a rank started by torch.distributed.run joins a process group even when it is the only one: `--nproc-per-node 1` runs the
    # barriers, the max-over-ranks and the metric all-gather through RCCL on a one-GPU box (tests/test_hip_cli.py)
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:
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
    # configs[1] once more on the exact-fp32 library (libpackppi_hip.f32.so: fp32-MFMA edge kernels, fp32 VALU node update) in a
    # child process of its own, started and finished BEFORE this process touches the GPU (the library is chosen at load time)
    f32_entry = None
    if (world == 1 and dist is None and args.workload == "t1124" and want_secondary and not os.environ.get("PACKPPI_LIB")):
        f32_entry = run_f32_child()

    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from packppi_amd.lib import Context
    from packppi_amd.module import TDiffusionModule
    from packppi_amd.weights import make_random_state_dict
    weights = make_random_state_dict(20251003)
    model = TDiffusionModule(weights, device=dev)
    from packppi_amd import lib as _lib0
    lib_is_split_f16 = _lib0.load().pp_edge_variant() == 1      # 1 = split-f16 (default), 0 = exact fp32 (libpackppi_hip.f32.so)
    model.schedule = torch.linspace(1, 0, N_DIFFUSION_STEPS + 1)
    counters = regime_counters()
    ref_chi, complexes = None, None

    def fence():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    def timed(fn, steps, warmup):
        out = None
        for _ in range(warmup):
            out = fn()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = fn()
        fence()
        el = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, out

    def allsum(v):
        if dist is None:
            return v
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    def c5_job(share_world, share_rank):
        """configs[4] over `share_world` ranks, this process playing `share_rank`: (residues of the share, one_pass, last)."""
        from packppi_amd.parallel import sample_sharded
        lens, cx = c5_share(share_rank, share_world, dev, c5_prot)
        inits = {i: v.to(dev) for i, v in c5_inits(cx, 1000 + share_rank).items()}      # resident before the timed region, like the batch
        last = {}

        def one_pass():
            chis, ids_all, rows_all = sample_sharded(model, cx, init_chi=inits, lengths=lens,
                                                     rank=share_rank if share_world != world else None,
                                                     world=share_world if share_world != world else None,
                                                     **({"max_rows": args.c5_max_rows} if args.c5_max_rows else {}))
            last["ids"], last["rows"] = ids_all, rows_all
            return chis
        return sum(c.true_residues() for c in cx.values()), one_pass, last, cx

    if args.workload == "c5":
        # BASELINE configs[4] as stated: ALL 256 complexes, dealt to the N ranks by parallel.shard_complexes, every rank runs
        # parallel.sample_sharded end to end on its share (packing, preparation, 100 evaluations, per-complex metrics) and the
        # metric rows of all 256 complexes are all-gathered (RCCL): total work is fixed -> "strong" scaling
        if args.proximal:
            raise SystemExit("--proximal with --workload c5: use the per-complex workloads")
        residues, one_pass, c5_last, complexes = c5_job(world, rank)
        name = ("BASELINE configs[4]: 256 synthetic complexes L~U{270..330} (default_rng(256)) sharded over the GPUs by "
                "parallel.shard_complexes, packed ragged batches, per-complex metrics + all-gather of the metric rows inside the timed pass")
    elif args.workload == "c5share":
        # one GPU's share of configs[4] at 8 GPUs (32 complexes) as ONE packed ragged batch: the profiling workload of the throughput regime
        from packppi_amd.batch import pack
        _, share = c5_share(rank % 8, 8, dev, c5_prot)
        ini = c5_inits(share, 1000 + rank)
        init_d = torch.cat([ini[i][:, : c.true_residues()] for i, c in share.items()], 1).to(dev)
        gb = pack(list(share.values()))
        batch, init = None, None
        residues = sum(c.true_residues() for c in share.values())
        name = "BASELINE configs[4], one GPU's share at 8 GPUs: 32 synthetic complexes L~U{270..330} as one packed ragged batch"
        ctx = Context(model._plan, gb)
        if args.proximal:
            raise SystemExit("--proximal with --workload c5share: use the per-complex workloads")

        def one_pass():
            return Context(model._plan, gb).sample(init_d, model.schedule)
    else:
        batch, init, ref_chi = load_t1124() if args.workload == "t1124" else load_s1500()
        name = {"t1124": "data/T1124_lig.pdb (L=739, 738 true residues), 1 complex per GPU",
                "s1500": "synthetic 1500-residue 2-chain complex (default_rng(1500)), 1 per GPU"}[args.workload]
        residues = batch.true_residues()
        gb, init_d = batch.to(dev), init.to(dev)
        ctx = model._context(gb)

        # One timed pass = what TDiffusionModule.sampling() costs on a batch it has not seen: the per-complex preparation
        # (kNN graph, frames, edge embedding, layer-0 static products: pp_complex_prepare, a fresh context) + 100 evaluations.
        def one_pass():
            chi = Context(model._plan, gb).sample(init_d, model.schedule)
            if args.proximal:
                from packppi_amd.functional import proximal_optimizer
                chis, losses = proximal_optimizer(gb, chi, 12.0, 0.5, 1.0, 50)
                chi = chis[-1] if losses[-1] < losses[0] else chi
            return chi

    elapsed, chi = timed(one_pass, args.steps, args.warmup)
    total_res = allsum(residues)

    # per-complex metric rows: the one real collective of the path (RCCL all-gather of fixed-width rows)
    if complexes is None and batch is None:
        atom_rmsd, row = None, torch.zeros(13, device=dev)
        rows = [row]
        ranks_seen, rows_gathered = 1, 0
    elif complexes is None:
        m = model.analyze_samples(gb, chi)
        row = torch.stack([torch.as_tensor(float(v), device=dev) for v in m.values()]).float()
        atom_rmsd = float(m["atom_rmsd"])
        rows = [row]
        if dist is not None:
            rows = [torch.empty_like(row) for _ in range(world)]
            dist.all_gather(rows, row)
        ranks_seen, rows_gathered = len(rows), len(rows)
    else:
        from packppi_amd.parallel import METRIC_KEYS
        # the gather happened inside every timed pass (parallel.gather_metric_rows): all 256 rows are on every rank
        rows_gathered = int(c5_last["ids"].numel())
        assert c5_last["ids"].tolist() == list(range(256)) and bool(torch.isfinite(c5_last["rows"]).all())
        atom_rmsd = float(c5_last["rows"][:, METRIC_KEYS.index("atom_rmsd")].mean())
        cnt = torch.tensor([1.0], device=dev)
        if dist is not None:
            dist.all_reduce(cnt)
        ranks_seen = int(cnt.item())
    max_dchi = None
    if ref_chi is not None and not args.proximal:
        d = (chi.cpu().double() - ref_chi.double()).abs()
        d = torch.minimum(d, (2 * np.pi - d).abs())[batch.SC_D_mask.bool()]
        max_dchi = float(d.max())

    # The other BASELINE configs on the same line (`secondary`, a list): configs[2] (T1124 + 50 proximal steps), S1500 without
    # and with proximal (configs[3]), one GPU's share of configs[4] as a packed batch, the WHOLE of configs[4] on this GPU, and
    # configs[1] on the exact-fp32 library.  At N > 1 (headline = configs[4] sharded): one T1124 replica per rank.
    secondary = None
    regimes = {}
    prox_roof = None
    if want_secondary and args.workload != "c5share":
        secondary = []

        def single_entry(tag, label, b_, init_, ref_, proximal, fixture, regime=None):
            gb_, init_d_ = b_.to(dev), init_.to(dev)
            last = {}

            def fn():
                chi_s = Context(model._plan, gb_).sample(init_d_, model.schedule)
                last["sampled"] = chi_s
                if proximal:
                    from packppi_amd.functional import proximal_optimizer
                    chis, losses = proximal_optimizer(gb_, chi_s, 12.0, 0.5, 1.0, 50)
                    last["losses"], last["end"] = losses, chis[-1]
                    return chis[-1] if losses[-1] < losses[0] else chi_s
                return chi_s
            k, w = 5, 2
            el_, chi_ = timed(fn, k, w)
            res_ = allsum(b_.true_residues())
            d = (last["sampled"].cpu().double() - ref_.double()).abs()
            d = torch.minimum(d, (2 * np.pi - d).abs())[b_.SC_D_mask.bool()]
            e = {"config": tag, "workload": label, "value": res_ * k / el_, "unit": "residues/s", "ms_per_step": el_ / k * 1e3,
                 "residues": res_, "steps": k, "warmup": w, "proximal": proximal,
                 "max_abs_dchi_vs_reference_rad": float(d.max()),
                 "max_abs_dchi_is": "the 100-step sample (before the proximal stage) vs the reference's chi_ode_100 on the same noise"}
            if proximal:
                z = np.load(os.path.join(ROOT, "tests", "golden", fixture + ".npz"))
                m_ = model.analyze_samples(gb_, chi_)
                end = last["end"].cpu().double()
                e["proximal"] = {"loss_first": last["losses"][0], "loss_last": last["losses"][-1],
                                 "reference_loss_first": float(z["losses32"][0]), "reference_loss_last": float(z["losses32"][-1]),
                                 "accepted": bool(last["losses"][-1] < last["losses"][0]),
                                 "atom_rmsd": float(m_["atom_rmsd"]), "reference_atom_rmsd": float(z["metric32.atom_rmsd"]),
                                 # the 50-step end state against the reference's own end states, and those against each other
                                 "end_state_vs_reference_fp32_rad": float((end - torch.from_numpy(z["chi32_step50"]).double()).abs().max()),
                                 "end_state_vs_reference_fp64_rad": float((end - torch.from_numpy(z["chi64_step50"]).double()).abs().max()),
                                 "reference_fp32_vs_fp64_rad": float(z["div_32_64"][-1]),
                                 "held_to": PROX_HELD_TO}
            if regime and not args.no_roofline:
                cx_ = Context(model._plan, gb_)
                t_e, n_e = insitu_edge(cx_, init_d_, model.schedule)
                regimes[regime] = regime_entry(regime, b_.true_residues(), cx_.K, t_e, n_e, counters)
            return e, gb_, last

        if world == 1 and args.workload == "t1124":
            e2, gb2, last2 = single_entry("configs[2]", "data/T1124_lig.pdb, 100 steps + 50 proximal Adam steps (vtf 12, tol 0.5, lamda 1)",
                                          batch, init, ref_chi, True, "g6_prox_T1124")
            secondary.append(e2)
            if not args.no_roofline:
                prox_roof = {"T1124": proximal_roofline(gb2, last2["sampled"])}
            bs, inits_, refs = load_s1500()
            secondary.append(single_entry("S1500", "synthetic 1500-residue 2-chain complex, 100 steps, no proximal", bs, inits_, refs,
                                          False, None, regime="s1500")[0])
            e3, gb3, last3 = single_entry("configs[3]", "synthetic 1500-residue complex, 100 steps + 50 proximal Adam steps",
                                          bs, inits_, refs, True, "g6_prox_S1500")
            secondary.append(e3)
            if not args.no_roofline:
                prox_roof["S1500"] = proximal_roofline(gb3, last3["sampled"])
        if complexes is None:
            # one GPU's share of configs[4] when it runs on 8 GPUs: rank r takes shard (r mod 8) of parallel.shard_complexes(256
            # lengths, 8) and runs the sampling part of parallel.sample_sharded on it (one packed ragged batch)
            _, share = c5_share(rank % 8, 8, dev, c5_prot)
            c5 = list(share.values())
            c5_init = c5_inits(share, 1000 + rank)
            c5_x0 = torch.cat([c5_init[i][:, : c.true_residues()] for i, c in share.items()], 1).to(dev)
            el5, _ = timed(lambda: sample_sharded_local(model, c5, c5_x0), 3, 1)
            res5 = allsum(sum(c.true_residues() for c in c5))
            secondary.append({"config": "configs[4] share",
                              "workload": "BASELINE configs[4], one GPU's share at 8 GPUs (parallel.shard_complexes(256 lengths, 8)[rank mod 8]: "
                                          "32 synthetic complexes L~U{270..330}) as one packed ragged batch (no padding rows), 100 steps, no "
                                          "proximal",
                              "value": res5 * 3 / el5, "unit": "residues/s", "ms_per_step": el5 / 3 * 1e3, "residues": res5,
                              "steps": 3, "warmup": 1, "complexes": len(c5) * world,
                              "max_abs_dchi_vs_reference_rad": None,
                              "parity_is": "tests/test_hip_parity.py::test_c5_all_256_complexes_match_reference (all 256 vs the reference, worst 8.6e-6 rad)"})
            if not args.no_roofline:
                from packppi_amd.batch import pack
                cx5 = Context(model._plan, pack(c5))
                t_e, n_e = insitu_edge(cx5, c5_x0, model.schedule)
                regimes["c5"] = regime_entry("c5", sum(c.true_residues() for c in c5), cx5.K, t_e, n_e, counters)
                del cx5
        if world == 1 and args.workload == "t1124":
            # ALL of configs[4] on this one GPU, through the sharded path end to end (parallel.sample_sharded with world = 1: packing,
            # preparation, 100 evaluations, per-complex metric rows, the gather): the N = 1 point of the strong-scaling curve
            resw, passw, lastw, _ = c5_job(1, 0)
            elw, _ = timed(passw, 2, 1)
            assert lastw["ids"].tolist() == list(range(256))
            secondary.append({"config": "configs[4] whole, one GPU",
                              "workload": "BASELINE configs[4]: all 256 complexes on ONE GPU through parallel.sample_sharded (what `--gpus N` "
                                          "measures over N ranks): the anchor of the strong-scaling curve",
                              "value": resw * 2 / elw, "unit": "residues/s", "ms_per_step": elw / 2 * 1e3, "residues": resw,
                              "steps": 2, "warmup": 1, "complexes": 256, "metrics_rows_gathered": int(lastw["ids"].numel()),
                              "max_abs_dchi_vs_reference_rad": None,
                              "parity_is": "tests/test_hip_parity.py::test_c5_all_256_complexes_match_reference"})
        if complexes is not None:
            # headline = configs[4] sharded; a single complex does not shard: one T1124 replica per rank, aggregate rate ("weak")
            bt, it, rt = load_t1124()
            et, _, _ = single_entry("configs[1] replicas", "data/T1124_lig.pdb, 100 steps, one replica per rank (a single complex does not "
                                    "shard: replicas only)", bt, it, rt, False, None)
            et["scaling"] = "weak"
            secondary.append(et)
        if f32_entry is not None:
            secondary.append(f32_entry)

    # kernel roofline, measured live: one more pass of the same workload in which every launch of the kernel carries a
    # start / stop HIP event pair on the launch stream (pp_profile_kernel -> hipExtLaunchKernelGGL: the dispatch's own
    # begin and end), the interval rocprofv3's kernel trace of this command reports (profiles/<PROFILE_TAG>_kernel_stats.csv).
    roof = None
    dtype = None
    if complexes is None and not args.no_roofline:
        insitu = {}
        for which, kname in ((1, "k_edge_update"), (0, "k_node_message"), (2, "k_node_update")):
            ctx.profile_kernel(which)
            ctx.sample(init_d, model.schedule)
            insitu[kname] = ctx.profile_read()
        t_edge = insitu["k_edge_update"][0] * 1e-3
        t_node = insitu["k_node_message"][0] * 1e-3
        t_nu = insitu["k_node_update"][0] * 1e-3
        n_edges = residues * ctx.K
        regimes["c5" if args.workload == "c5share" else args.workload] = regime_entry("c5" if args.workload == "c5share" else args.workload,
                                                                                     residues, ctx.K, t_edge, insitu["k_edge_update"][1], counters)
        # k_edge_update(l) also computes the node message of layer l + 1 (fused): its algorithmic work is both MLP chains
        # of the reference (layers.py:119-148), 2 FLOP per MAC of the dense layers, per edge.  Executed MFMA work is lower:
        # layer 0's W_B h_E0 products are timestep-invariant and computed once per complex.
        fused_flop_per_edge = EDGE_UPDATE_FLOP_PER_EDGE + NODE_MSG_FLOP_PER_EDGE
        achieved = fused_flop_per_edge * n_edges / t_edge / 1e12
        split_f16 = lib_is_split_f16
        if split_f16:
            # every WAVE of a workgroup issues 342 (layer 1) / 318 (layer 0) v_mfma_f32_32x32x16_f16 per launch (SQ_INSTS_MFMA /
            # residues = 4 x that, profiles/*_sq_counters.txt), 32768 FLOP each; every fp32 product is three f16 products
            executed_mfma = 4 * 0.5 * (342 + 318) * 32768.0 * residues
            peak, dtype = F16_MFMA_PEAK_TFLOPS, "f32 (dense layers as split-f16: two f16 per operand, three f16 MFMAs per product, fp32 accumulate)"
        else:
            executed_mfma = (2960 + 656 - 128) * 4096.0 * residues    # average of the layer-0 and layer-1 launches
            peak, dtype = FP32_MFMA_PEAK_TFLOPS, "f32"
        whole = 46792576.0 * total_res * N_DIFFUSION_STEPS * args.steps / elapsed / 1e12 / max(args.gpus, 1)
        # what the kernel is closest to: the L2 -> CU operand stream (every workgroup pulls the layer's packed weight set)
        l2 = l2_requests("k_edge_update") if args.workload == "t1124" else None
        l2_stream = None
        if l2 is not None:
            l2_stream = {"achieved": l2["l2_request_bytes"] / t_edge / 1e12, "unit": "TB/s", "peak": L2_PEAK_TBS,
                         "frac": l2["l2_request_bytes"] / t_edge / 1e12 / L2_PEAK_TBS,
                         "practical": "16.8-18.8 TB/s is what a kernel that only pulls L2-resident rows reaches (MI355X_MICROARCH.md, indexed rows)",
                         "l2_hit_rate": l2["l2_hit_rate"], "requests_per_launch": l2["l2_requests"],
                         "of_which_weight_stream_bytes": 917504.0 * 1.0 * ((residues + 2) // 3 + max(residues - 2 * ((residues + 2) // 3), 0)),
                         "source": f"profiles/{PROFILE_TAG}_tcc_t1124.json (rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum, own pass) / live kernel time"}
        roof = {"bound": "mfma", "kernel": "k_edge_update", "achieved": achieved,
                "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "peak_is": "dense F16 MFMA (the pipe the kernel runs on)" if split_f16 else "FP32 matrix",
                "limiter": "not the matrix pipe (SQ counters: `regimes[*].mfma_busy`): in a one-round launch the per-CU vector-memory "
                           "path (64 B/clk), spent on the weight stream -- every workgroup pulls the layer's 0.93 MB packed weight set, 1.86 MB "
                           "per CU and launch at three residues per CU; the workgroup kind of the mixed launch that is dispatched second "
                           "queues its prologue loads behind the other's stream.  `l2_stream` is the same quantity seen from the L2 "
                           "(DESIGN.md 4.5, 4.6); `bound` names the roofline that would bind at the limit",
                "achieved_is": "the reference's fp32 dense-layer arithmetic (2 FLOP per MAC) per second; the kernel "
                               "issues 3 f16 MFMAs per product, see executed_mfma_tflops",
                "achieved_over_fp32_matrix_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                "edge_kernels": "split-f16" if split_f16 else "fp32",
                "traffic": pmc_traffic("k_edge_update") if args.workload == "t1124" else None,
                "traffic_unit": f"HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes of this "
                                f"command; profiles/{PROFILE_TAG}_pmc_traffic.json)",
                "l2_stream": l2_stream,
                "kernel_ms": t_edge * 1e3,
                "kernel_ms_is": "mean begin-to-end interval of the dispatches inside the sampling loop (start/stop HIP events attached to each launch)",
                "kernel_does": "edge update of layer l + node message of layer l+1, one launch",
                "algorithmic_flop_per_launch": fused_flop_per_edge * n_edges,
                "executed_mfma_tflops": executed_mfma / t_edge / 1e12,
                # whole pass against SURVEY 8(d): 46 792 576 algorithmic FLOP per residue per network evaluation
                "whole_pass_algorithmic_tflops": whole,
                "whole_pass_frac_of_peak": whole / FP32_MFMA_PEAK_TFLOPS,
                "kernel_launches_timed": insitu["k_edge_update"][1],
                "node_message_kernel_ms": t_node * 1e3,
                "node_message_layer0_tflops": NODE_MSG_FLOP_PER_EDGE * n_edges / t_node / 1e12,
                # the same kernel in the other regimes (S1500; the configs[4] share) + the committed counters of each
                "regimes": regimes,
                "regimes_counters_source": f"profiles/{PROFILE_TAG}_regimes.json (rocprofv3 --pmc passes per workload, tools/profile/regimes.py)",
                # second kernel: the node update (mean over its three launches per evaluation)
                "node_update": {
                    "kernel": "k_node_update", "kernel_ms": t_nu * 1e3, "launches_timed": insitu["k_node_update"][1],
                    "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                    "algorithmic_bytes_per_launch": residues * (2 * 512 + 5 * 512 + 2 * 192 + 16) + NODE_UPDATE_STREAM_BYTES,
                    "achieved": (residues * (2 * 512 + 5 * 512 + 2 * 192 + 16) + NODE_UPDATE_STREAM_BYTES) / t_nu / 1e9,
                    "frac": (residues * (2 * 512 + 5 * 512 + 2 * 192 + 16) + NODE_UPDATE_STREAM_BYTES) / t_nu / 1e9 / HBM_PEAK_GBS,
                    "algorithmic_tflops": NODE_UPDATE_FLOP_PER_RES * residues / t_nu / 1e12,
                    "limiter": "neither HBM nor the matrix pipe: (residues / 16) workgroups each pull the layer's whole "
                               "packed weight set through ONE CU's vector-memory path, the launch lasts as long as one CU "
                               "needs for that stream plus launch and input latency",
                    "weight_stream_bytes_per_workgroup": NODE_UPDATE_STREAM_BYTES,
                    "weight_stream_GBs_per_CU_lower_bound": NODE_UPDATE_STREAM_BYTES / t_nu / 1e9,
                    "CU_vector_memory_peak_GBs": CU_VMEM_PEAK_GBS}}
        if prox_roof is not None:
            roof["proximal"] = prox_roof

    if rank == 0:
        out = {
            "metric": "sampled residues/sec at 100 diffusion steps",
            "value": total_res * args.steps / elapsed,
            "unit": "residues/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.workload == "c5" else "weak", "vs_baseline": None,
            "dtype": ("f32 (dense layers as split-f16: two f16 per operand, three f16 MFMAs per product, fp32 accumulate)"
                      if lib_is_split_f16 else "f32"),
            "library": os.path.basename(os.environ.get("PACKPPI_LIB") or "libpackppi_hip.so"),
            "data": "synthetic (seeded random weights; T1124 backbone fixture, seeded initial noise)"
            if args.workload == "t1124" else "synthetic (seeded random weights, random-backbone complexes, seeded initial noise)",
            "config": {"workload": name, "diffusion_steps": N_DIFFUSION_STEPS, "proximal": bool(args.proximal),
                       "residues_per_gpu": residues, "residues_rank0": residues, "residues_total": total_res, "mode": "ode"},
            "parity": {"max_abs_dchi_vs_reference_rad": max_dchi, "atom_rmsd": atom_rmsd,
                       "proximal_configs_held_to": PROX_HELD_TO},
            "ranks_seen": ranks_seen, "metrics_rows_gathered": rows_gathered, "dist_backend": backend,
        }
        if complexes is not None:
            out["config"]["complexes_total"] = 256
            out["config"]["complexes_this_rank"] = len(complexes)
            out["parity"]["parity_is"] = "tests/test_hip_parity.py::test_c5_all_256_complexes_match_reference (all 256 vs the reference, <= 1e-4 rad)"
        if roof is not None:
            out["dtype"] = dtype
            out["roofline"] = roof
        if secondary is not None:
            out["secondary"] = secondary
        if args.cpu_steps > 0 and args.gpus == 1 and complexes is None and batch is not None:
            if args.cpu_protocol == "full":
                out["cpu_baseline"] = cpu_baseline_full(batch, init, weights, args.cpu_threads or min(16, os.cpu_count() or 1))
            else:
                out["cpu_baseline"] = cpu_baseline(batch, init, weights, args.cpu_steps, args.cpu_grad_steps)
        out["summary"] = make_summary(out)          # LAST key: a flat digest that survives a 2 000-character tail
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def make_summary(out):
    """Every config and regime of the line in <= 1.5 KB: per config [residues/s, ms per pass, max |dchi| vs the reference in rad]; per
    regime of the dominant kernel [roofline frac, MFMA-busy share, L2 request bytes / algorithmic HBM bytes]; the proximal kernels."""
    def r3(x):
        return None if x is None else float(f"{x:.4g}")
    short = {"configs[2]": "c2_t1124_prox", "S1500": "s1500", "configs[3]": "c3_s1500_prox", "configs[4] share": "c4_share_32cx",
             "configs[4] whole, one GPU": "c4_whole_1gpu", "configs[1], exact-fp32 library": "c1_f32lib", "configs[1] replicas": "c1_replicas"}
    head = "c4_sharded" if out["scaling"] == "strong" else ("c1_t1124" if "T1124" in out["config"]["workload"] else "headline")
    cfg = {head: [r3(out["value"]), r3(out["ms_per_step"]), r3(out["parity"]["max_abs_dchi_vs_reference_rad"])]}
    for e in out.get("secondary") or []:
        if "value" in e:
            cfg[short.get(e["config"], e["config"])] = [r3(e["value"]), r3(e["ms_per_step"]), r3(e.get("max_abs_dchi_vs_reference_rad"))]
            if isinstance(e.get("proximal"), dict):
                cfg[short.get(e["config"], e["config"])].append(r3(e["proximal"]["end_state_vs_reference_fp32_rad"]))
    s = {"n_gpus": out["n_gpus"], "cfg": cfg, "cfg_is": "[res/s, ms/pass, max|dchi| rad (, proximal end state vs reference fp32 rad)]"}
    roof = out.get("roofline")
    if roof:
        s["roof"] = {k: [r3(v["frac"]), r3(v["mfma_busy"]), r3(v["l2_over_algorithmic"])] for k, v in roof.get("regimes", {}).items()}
        s["roof_is"] = "k_edge_update [frac of dense F16 peak, MFMA busy, L2 request bytes / algorithmic HBM bytes]"
        if roof.get("proximal"):
            s["prox"] = {k: [r3(v["k_clash"]["kernel_us"]), r3(v["k_clash"]["frac"]), r3(v["k_clash"]["culled_fraction_of_residue_pairs"]),
                             r3(v["k_clash"].get("hbm_frac"))] for k, v in roof["proximal"].items()}
            s["prox_is"] = "the one launch per Adam step (clash + gradient + step + reconstruction): [us, frac of fp32 VALU peak, culled residue pairs, frac of HBM]"
    cb = out.get("cpu_baseline")
    if cb:
        s["cpu"] = [r3(cb["value"]), cb["cores"], cb["kind"]]
    return s


def run_f32_child():
    """configs[1] on libpackppi_hip.f32.so (no f16 operand in any kernel): `python bench.py --steps 5 --warmup 2` as a child with
    PACKPPI_LIB set; returns the `secondary` entry (or one that says why there is none)."""
    lib = os.path.join(ROOT, "packppi_amd", "csrc", "libpackppi_hip.f32.so")
    entry = {"config": "configs[1], exact-fp32 library", "library": "libpackppi_hip.f32.so",
             "workload": "data/T1124_lig.pdb, 100 steps, no proximal, every dense layer in fp32 (fp32-MFMA edge kernels, fp32 VALU "
                         "node update): what the split-f16 arithmetic of the headline buys"}
    if not os.path.exists(lib):
        entry["error"] = "libpackppi_hip.f32.so is not built (python __graft_entry__.py builds it)"
        return entry
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "5", "--warmup", "2", "--cpu-steps", "0", "--no-secondary", "--no-roofline"]
    try:
        env = {k: v for k, v in os.environ.items() if k not in RENDEZVOUS_ENV}
        r = subprocess.run(cmd, env=dict(env, PACKPPI_LIB=lib), capture_output=True, text=True, timeout=300)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        o = json.loads(line)
        entry.update({"value": o["value"], "unit": o["unit"], "ms_per_step": o["ms_per_step"], "residues": o["config"]["residues_per_gpu"],
                      "steps": o["steps"], "warmup": o["warmup"], "dtype": "f32",
                      "max_abs_dchi_vs_reference_rad": o["parity"]["max_abs_dchi_vs_reference_rad"]})
    except Exception as exc:          # the headline must not depend on this leg
        entry["error"] = f"{type(exc).__name__}: {exc}"[:300]
    return entry


def sample_sharded_local(model, complexes, inits):
    """The sampling part of parallel.sample_sharded for complexes this rank already owns (no metric gather): one packed
    ragged batch (packing and context preparation inside the timed pass, like the preparation of the single-complex
    workloads); ``inits`` = the initial noised angles of the packed rows, already on the device."""
    from packppi_amd.batch import pack
    from packppi_amd.lib import Context
    pb = pack(complexes)
    return Context(model._plan, pb).sample(inits, model.schedule)


if __name__ == "__main__":
    main()
