"""Sharding independent complexes over the GPUs of a node.

The sampling path has no cross-complex term (``network`` is per batch row; ``proximal_optimizer`` asserts B == 1),
so complexes are dealt to ranks, every rank runs the whole path on its own shard, and the ONLY collective is an
all-gather of fixed-width per-complex metric rows (RCCL over xGMI when the process group backend is "nccl";
KB-sized, latency bound).  One process per GPU, launched by ``torchrun``.
"""
from typing import Dict, List, Sequence, Tuple

import torch
import torch.distributed as dist

METRIC_KEYS = tuple(f"chi_{i}_{s}" for i in range(4) for s in ("ae_rad", "ae_deg", "acc")) + ("atom_rmsd",)


def shard_complexes(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of complex indices to ranks (cost ~ residues)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    out: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(s) for s in out]


def metrics_to_row(metric: Dict[str, torch.Tensor]) -> torch.Tensor:
    return torch.stack([torch.as_tensor(metric[k], dtype=torch.float32).reshape(()) for k in METRIC_KEYS])


def packed_metric_rows(model, pb, chi, sizes) -> torch.Tensor:
    """``analyze_samples`` (TorsionalDiffusion.py:311-341) for every complex of a packed batch at once: one atom14 launch
    and segment sums instead of one context, one launch and a dozen small reductions per complex.  ``sizes`` = the complexes'
    own ``max_size`` (the reference's atom_rmsd denominator adds eps for every atom slot of the complex, padding included).
    Returns [n_complexes, len(METRIC_KEYS)] on the batch's device, rows in packing order."""
    import numpy as np
    offs = pb["seg_offsets_host"]
    n = len(offs) - 1
    dev = chi.device
    lens = torch.tensor([b - a for a, b in zip(offs[:-1], offs[1:])], device=dev)
    seg = torch.repeat_interleave(torch.arange(n, device=dev), lens)              # complex of every packed row

    def seg_sum(x):                                                                  # [1, N, ...] -> [n] sums per complex
        x = x.reshape(x.shape[1], -1).sum(-1)
        return torch.zeros(n, device=dev, dtype=x.dtype).index_add_(0, seg, x)

    true, m, pi1 = pb["SC_D"], pb["SC_D_mask"], pb["chi_1pi_periodic_mask"]
    cols = []
    for i in range(4):
        cnt = seg_sum(m[..., i])
        cnt = torch.where(cnt != 0, cnt, torch.ones_like(cnt))
        diff = (chi[..., i] - true[..., i]).abs()
        acc = torch.logical_and(diff * 180 / np.pi < 20, diff > 0).float()
        ae = torch.minimum(diff, 2 * np.pi - diff)
        ae = torch.where(pi1[..., i], torch.minimum(ae, np.pi - ae), ae)
        cols += [seg_sum(ae) / cnt, seg_sum(ae * 180 / np.pi) / cnt, seg_sum(acc) / cnt]
    pred = model.get_atom14_coords(pb, chi)
    w = pb["atom_mask"] * pb["residue_mask"][..., None]
    num = seg_sum(torch.sum((pb["X"] - pred) ** 2, dim=-1) * w)
    den = seg_sum(w) + model.eps * 14.0 * torch.tensor([float(sz) for sz in sizes], device=dev)
    cols.append(num / den)
    return torch.stack(cols, 1)


def gather_metric_rows(ids: torch.Tensor, rows: torch.Tensor, group=None) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather ragged (ids [n_r], rows [n_r, W]) from every rank; returns them sorted by complex id.

    Counts are exchanged first so each rank pads its block to the maximum; works for gloo (CPU tensors) and
    nccl/RCCL (device tensors) alike."""
    if not dist.is_available() or not dist.is_initialized():
        order = torch.argsort(ids)
        return ids[order], rows[order]
    world = dist.get_world_size(group)
    dev = rows.device
    n = torch.tensor([ids.numel()], device=dev, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c) for c in torch.cat(counts).tolist()]            # one read-back for all ranks
    cap = max(max(counts), 1)
    W = rows.shape[1] if rows.dim() == 2 else len(METRIC_KEYS)
    # ids travel as int64 in their own block (a float32 column is exact to 2^24 only), the rows as float32
    id_block = torch.zeros(cap, device=dev, dtype=torch.int64)
    block = torch.zeros(cap, W, device=dev, dtype=torch.float32)
    if ids.numel():
        id_block[: ids.numel()] = ids.to(torch.int64)
        block[: ids.numel()] = rows.to(torch.float32)
    id_blocks = [torch.empty_like(id_block) for _ in range(world)]
    blocks = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(id_blocks, id_block, group=group)
    dist.all_gather(blocks, block, group=group)
    all_ids = torch.cat([b[:c] for b, c in zip(id_blocks, counts)], 0)
    all_rows = torch.cat([b[:c] for b, c in zip(blocks, counts)], 0)
    order = torch.argsort(all_ids)
    return all_ids[order], all_rows[order]


def sample_sharded(model, complexes, use_proximal=False, group=None, init_chi=None, max_rows=200_000, lengths=None,
                   rank=None, world=None):
    """Run the sampling path on this rank's share of ``complexes`` (list of B = 1 batches already on the rank's device)
    and gather every complex's metric row on every rank.  With ``lengths`` (the residue counts of ALL complexes, known to
    every rank) ``complexes`` may be a dict {complex id: batch} that holds only this rank's share -- a rank need not build
    the other ranks' inputs.  ``rank`` / ``world`` override the process group's (a single process rehearsing the shards of
    an N-rank job one after the other; the gather then only orders the local rows).

    The shard is sampled as ragged PACKED batches (``batch.pack``: no padding rows are launched; complexes shorter than 32
    residues go alone because K = min(32, L)), at most ``max_rows`` residues per batch; the proximal stage, which the
    reference defines for one complex at a time (optimize.py:27), and the metrics then run per complex.
    ``init_chi`` (optional, {complex id: [1, L, 4]}) injects the initial noised angles instead of drawing them.
    Returns (chi per local complex id, ids_all, rows_all)."""
    from .batch import pack, unpack
    from .functional import proximal_optimizer
    if rank is None:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if lengths is None:
        lengths = [int(c["max_size"]) for c in complexes]
    mine = shard_complexes(lengths, world)[rank]
    cfg = model.hparams.sample_cfg
    groups, cur, rows_in = [], [], 0
    true_counts = (torch.stack([(complexes[i]["residue_mask"] > 0).sum() for i in mine]).tolist() if mine else [])   # one read-back
    for i, n_true in zip(mine, true_counts):
        n = int(complexes[i]["max_size"])
        # K = min(32, L) is a property of the batch (encoder.py:115): a complex that is shorter than 32 rows once its trailing
        # padding is dropped goes through the plain B = 1 path with its own tensors, as the reference would run it
        if n_true < 32 or n < 32:
            groups.append([i])
            continue
        if cur and rows_in + n > max_rows:
            groups.append(cur)
            cur, rows_in = [], 0
        cur.append(i)
        rows_in += n
    if cur:
        groups.append(cur)
    chis, row_of = {}, {}
    for grp in groups:
        if len(grp) == 1:
            i = grp[0]
            if init_chi is not None:
                chis[i] = model.sample_from(complexes[i], init_chi[i].to(model.device))
            else:
                chis[i] = model.sampling(complexes[i])
            continue
        pb = pack([complexes[i] for i in grp])
        if init_chi is not None:
            offs = pb["seg_offsets_host"]
            x0 = torch.cat([init_chi[i][:, :b - a] for i, a, b in zip(grp, offs[:-1], offs[1:])], 1).to(model.device)
            out = model.sample_from(pb, x0)
        else:
            out = model.sampling(pb)
        if not use_proximal:          # the metrics of the whole group in one go (the proximal stage changes the angles first)
            for i, row in zip(grp, packed_metric_rows(model, pb, out, [int(complexes[i]["max_size"]) for i in grp])):
                row_of[i] = row
        for i, chi in zip(grp, unpack(pb, out)):
            L = int(complexes[i]["max_size"])
            if chi.shape[1] == L:
                chis[i] = chi
            else:
                full = torch.zeros(1, L, 4, device=chi.device, dtype=chi.dtype)
                full[:, :chi.shape[1]] = chi
                chis[i] = full
    rows = []
    for i in mine:
        if use_proximal:
            lst, losses = proximal_optimizer(complexes[i], chis[i], cfg.violation_tolerance_factor,
                                             cfg.clash_overlap_tolerance, cfg.lamda, cfg.num_steps)
            if losses[-1] < losses[0]:
                chis[i] = lst[-1]
        rows.append(row_of[i] if i in row_of else metrics_to_row(model.analyze_samples(complexes[i], chis[i])))
    dev = model.device
    ids = torch.tensor(mine, device=dev, dtype=torch.int64)
    rows_t = torch.stack(rows).to(dev) if rows else torch.zeros(0, len(METRIC_KEYS), device=dev)
    ids_all, rows_all = gather_metric_rows(ids, rows_t, group)
    return chis, ids_all, rows_all
