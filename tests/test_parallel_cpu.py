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


class _FakeModel:
    """CPU stand-in for TDiffusionModule on the sharded path: 'sampling' and 'atom14' are fixed functions of the rows, the
    metric code is the product's own (TDiffusionModule.analyze_samples / compute_rmsd, unbound), so the packing, unpacking,
    shard assignment, the packed metric rows and the gather can be checked without a GPU."""
    device = torch.device("cpu")
    NUM_CHI_ANGLES, eps = 4, 1e-6

    def __init__(self):
        from types import SimpleNamespace
        self.hparams = SimpleNamespace(sample_cfg=SimpleNamespace(violation_tolerance_factor=12., clash_overlap_tolerance=0.5,
                                                                  lamda=1., num_steps=2))
        self.packed_shapes = []

    def sample_from(self, batch, x0, sde_noise=None):
        assert batch.num_proteins == 1 and x0.shape == (1, batch.max_size, 4)
        self.packed_shapes.append((int(batch.max_size), batch.get("seg_offsets_host") or [0, int(batch.max_size)]))
        return (0.5 * x0 + batch.residue_type[..., None].float() * 0.01) * batch.SC_D_mask

    def get_atom14_coords(self, batch, chi):
        return batch.X + 0.05 * chi.sum(-1)[..., None, None] * batch.atom_mask[..., None]

    def _geometry_context(self, batch):
        from types import SimpleNamespace
        return SimpleNamespace(atom14=lambda chi: self.get_atom14_coords(batch, chi))

    def compute_rmsd(self, *a):
        from packppi_amd.module import TDiffusionModule
        return TDiffusionModule.compute_rmsd(self, *a)

    def analyze_samples(self, batch, chi):
        from packppi_amd.module import TDiffusionModule
        return TDiffusionModule.analyze_samples(self, batch, chi)


def _sharded_worker(rank, world, port, q):
    from packppi_amd import synth
    from packppi_amd.featurize import protein_to_batch
    from packppi_amd.parallel import sample_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = [40, 20, 57, 33, 64]                      # the 20-residue complex (K = 20) must go alone
    cs = [protein_to_batch(synth.make_complex(n, 50 + n)) for n in lens]
    cs[2].residue_mask[0, 17] = 0.0                  # a residue masked out mid-chain (missing backbone atom) stays in place
    g = torch.Generator().manual_seed(1)
    init = {i: torch.rand(1, n, 4, generator=g) for i, n in enumerate(lens)}
    model = _FakeModel()
    chis, ids, rows = sample_sharded(model, cs, init_chi=init, max_rows=100)
    expect = {i: (0.5 * init[i] + cs[i].residue_type[..., None].float() * 0.01) * cs[i].SC_D_mask for i in chis}
    ok = all(torch.equal(chis[i], expect[i]) for i in chis)
    # every gathered row (packed groups: parallel.packed_metric_rows) equals the per-complex analyze_samples of that complex
    from packppi_amd.parallel import metrics_to_row
    full = {i: (0.5 * init[i] + cs[i].residue_type[..., None].float() * 0.01) * cs[i].SC_D_mask for i in range(len(lens))}
    want = torch.stack([metrics_to_row(model.analyze_samples(cs[i], full[i])) for i in range(len(lens))])
    ok = ok and bool(torch.allclose(rows, want, rtol=1e-5, atol=1e-7)) and bool((want[:, -1] > 0).all())
    q.put((rank, sorted(chis), model.packed_shapes, ids.tolist(), ok))
    dist.barrier()
    dist.destroy_process_group()


def test_sample_sharded_packs_and_gathers_gloo_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lens = [40, 20, 57, 33, 64]
    owned = []
    for rank, mine, shapes, ids, ok in got:
        assert ok and mine == shard_complexes(lens, world)[rank]
        owned += mine
        assert ids == list(range(5))                                                # every rank sees every row
        for total, offs in shapes:                                                  # packed groups: no padding rows
            seg = [b - a for a, b in zip(offs[:-1], offs[1:])]
            assert total == sum(seg) and (min(seg) >= 32 or len(seg) == 1) and (total <= 100 or len(seg) == 1)
    assert sorted(owned) == list(range(5))


def test_gather_arguments_are_what_a_device_backend_needs(monkeypatch):
    """The "nccl" (RCCL) branch cannot run here; what it is handed can be checked: every all_gather gets a list of `world`
    buffers with the shape, dtype and device of the contiguous input tensor -- same size on every rank, padded to the longest
    shard -- and the result is assembled from the true counts."""
    from packppi_amd import parallel
    calls = []

    class FakeDist:
        """two ranks; the other rank holds one complex more, with ids shifted by 100"""
        @staticmethod
        def is_available():
            return True

        @staticmethod
        def is_initialized():
            return True

        @staticmethod
        def get_world_size(group=None):
            return 2

        @staticmethod
        def all_gather(outs, t, group=None):
            assert len(outs) == 2 and t.is_contiguous()
            for o in outs:
                assert o.shape == t.shape and o.dtype == t.dtype and o.device == t.device and o.is_contiguous()
            calls.append((tuple(t.shape), t.dtype))
            outs[0].copy_(t)
            if t.dtype == torch.int64 and t.numel() == 1:          # the counts
                outs[1].copy_(t + 1)
            elif t.dtype == torch.int64:                          # [cap] ids: rank 1 = rank 0's ids + 100, and one more
                other = torch.zeros_like(t)
                other[:3] = t[:3] + 100
                other[3] = (1 << 40) + 7                           # survives: ids are not squeezed through float32
                outs[1].copy_(other)
            else:                                                 # [cap, W] rows: rank 1 = rank 0's rows, and one more
                other = torch.zeros_like(t)
                other[:3] = t[:3]
                other[3] = 7.0
                outs[1].copy_(other)

    monkeypatch.setattr(parallel, "dist", FakeDist)
    W = len(METRIC_KEYS)
    ids = torch.tensor([4, 2, 9])
    rows = torch.arange(3 * W, dtype=torch.float32).reshape(3, W) + 1
    ids_all, rows_all = gather_metric_rows(ids, rows)
    assert calls == [((1,), torch.int64), ((4,), torch.int64), ((4, W), torch.float32)]      # padded to the longest shard (3 + 1)
    assert ids_all.tolist() == [2, 4, 9, 102, 104, 109, (1 << 40) + 7] and rows_all.shape == (7, W) and ids_all.dtype == torch.int64
    assert torch.equal(rows_all[0], rows[1]) and torch.equal(rows_all[3], rows[1]) and bool((rows_all[6] == 7).all())


def test_bench_refuses_a_rank_count_mismatch():
    """bench.py --gpus N under a launcher with a different WORLD_SIZE exits non-zero before touching any GPU."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=3" in (r.stderr + r.stdout)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "0"], env=env, capture_output=True, text=True,
                       timeout=120)
    assert r.returncode != 0
