"""The N > 1 path on CPU: two gloo ranks shard the rays in blocks, trace their
shard (with the CPU oracle standing in for the GPU), all-reduce the tally, and
must reproduce the one-process tally exactly (integer sums)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from turtle_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_PER_RANK, N_MEDIA, N_BINS, LMAX = 600, 2, 64, 16384.0


def trace_block(rank):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import terrains as T
    from oracle import ffi as O
    geo = T.c1_oracle()
    lat, lon, az, el = sharding.rank_rays(N_PER_RANK, rank, T.C1_Y, T.C1_X, seed=99)
    pos, _ = geo.position(lat, lon, 500.0)
    return geo.trace(pos, O.ecef_from_horizontal(lat, lon, az, el))


def local_tally(t):
    h, g, steps_at, size = sharding.tally_layout(N_MEDIA, N_BINS)
    v = torch.zeros(size, dtype=torch.int64)
    hits, hist = sharding.tally_reference(t["index"], t["length"], N_MEDIA, N_BINS, LMAX)
    v[h], v[g], v[steps_at] = torch.as_tensor(hits), torch.as_tensor(hist), int(t["n_steps"].sum())
    return v


def worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    v = local_tally(trace_block(rank))
    sharding.all_reduce_tally(v, world)
    t = torch.tensor([1.0 + rank])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # bench.py's max-over-ranks timing
    if rank == 0:
        np.save(out, np.concatenate([v.numpy(), [int(t.item())]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one(tmp_path):
    out = os.path.join(tmp_path, "tally.npy")
    mp.spawn(worker, args=(2, 29500 + os.getpid() % 2000, out), nprocs=2, join=True)
    got = np.load(out)
    whole = sum(local_tally(trace_block(r)) for r in range(2)).numpy()
    assert np.array_equal(got[:-1], whole) and got[-1] == 2
    h, g, steps_at, _ = sharding.tally_layout(N_MEDIA, N_BINS)
    assert whole[h].sum() == 2 * N_PER_RANK == whole[g].sum() and whole[steps_at] > 0


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 1000, 10 ** 8):
        for world in (1, 2, 3, 8):
            b = [sharding.shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


def test_rank_blocks_are_distinct_and_reproducible():
    a = sharding.rank_rays(100, 0, (45, 46), (3, 4))
    b = sharding.rank_rays(100, 1, (45, 46), (3, 4))
    a2 = sharding.rank_rays(100, 0, (45, 46), (3, 4))
    assert not np.array_equal(a[0], b[0]) and np.array_equal(a[0], a2[0])
