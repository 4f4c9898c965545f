"""Multi-process path on CPU (gloo, world_size 2): sharding + the metric-row all-gather (the path's only collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from packppi_amd.parallel import METRIC_KEYS, gather_metric_rows, shard_complexes


def test_shard_complexes_balanced_and_complete():
    lens = [300, 270, 330, 299, 310, 280, 305, 1500, 64]
    for world in (1, 2, 4, 8):
        shards = shard_complexes(lens, world)
        assert sorted(i for s in shards for i in s) == list(range(len(lens)))
        loads = [sum(lens[i] for i in s) for s in shards]
        assert max(loads) - min(loads) <= max(lens)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, lens, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_complexes(lens, world)[rank]
    ids = torch.tensor(mine, dtype=torch.int64)
    rows = torch.stack([torch.full((len(METRIC_KEYS),), float(i)) + torch.arange(len(METRIC_KEYS)) * 0.5
                        for i in mine]) if mine else torch.zeros(0, len(METRIC_KEYS))
    ids_all, rows_all = gather_metric_rows(ids, rows)
    q.put((rank, ids_all.tolist(), rows_all.tolist()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lens", [[300, 270, 330, 299, 310], [64]])
def test_gather_metric_rows_gloo_world2(lens):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, lens, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for _, ids, rows in got:
        assert ids == list(range(len(lens)))
        for i, row in zip(ids, rows):
            assert row == [float(i) + 0.5 * k for k in range(len(METRIC_KEYS))]
