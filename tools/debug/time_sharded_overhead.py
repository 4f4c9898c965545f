"""What parallel.sample_sharded adds to the bare sampling of one GPU's share of configs[4] (32 complexes): packing, the metric rows,
unpacking, the gather.  Rehearses rank 0 of an 8-rank job on one GPU.   python tools/debug/time_sharded_overhead.py [reps]"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
import torch
import bench
from packppi_amd.batch import pack, unpack
from packppi_amd.lib import Context
from packppi_amd.module import TDiffusionModule
from packppi_amd.parallel import sample_sharded, packed_metric_rows
from packppi_amd.weights import make_random_state_dict

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
prot = bench.c5_proteins(__import__("packppi_amd.parallel", fromlist=["x"]).shard_complexes(__import__("packppi_amd.synth", fromlist=["x"]).c5_lengths(256), 8)[0], 8)
model = TDiffusionModule(make_random_state_dict(20251003), device=dev)
model.schedule = torch.linspace(1, 0, 101)
lens, cx = bench.c5_share(0, 8, dev, prot)
inits = {i: v.to(dev) for i, v in bench.c5_inits(cx, 1000).items()}
x0 = torch.cat([inits[i][:, : c.true_residues()] for i, c in cx.items()], 1)


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


t_bare, _ = timed(lambda: bench.sample_sharded_local(model, list(cx.values()), x0), reps)
t_full, _ = timed(lambda: sample_sharded(model, cx, init_chi=inits, lengths=lens, rank=0, world=8), reps)
pb = pack(list(cx.values()))
t_pack, _ = timed(lambda: pack(list(cx.values())), reps)
chi = Context(model._plan, pb).sample(x0, model.schedule)
t_rows, _ = timed(lambda: packed_metric_rows(model, pb, chi, [int(c["max_size"]) for c in cx.values()]), reps)
t_unpack, _ = timed(lambda: unpack(pb, chi), reps)
print(f"bare sampling of the packed share {t_bare:.2f} ms | sample_sharded {t_full:.2f} ms | of which pack {t_pack:.2f}, metric rows {t_rows:.2f}, unpack {t_unpack:.2f}")
