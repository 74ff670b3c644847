"""Multi-GPU model: one process per GPU, rays block-partitioned, terrain
replicated, no collective on the data path.  The only exchange
is the sum of the uint64 tally vector (hit counts, path-length histogram, step
count) after tracing: one all-reduce of ~8 KB per pass over RCCL ("nccl" in
torch.distributed); the same code runs over gloo in the CPU tests.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n_total: int, rank: int, world: int):
    """Contiguous block of the global ray array owned by `rank`:
    [floor(rank*n/world), floor((rank+1)*n/world))."""
    return (n_total * rank) // world, (n_total * (rank + 1)) // world


def rank_rays(n_per_rank, rank, lat_range, lon_range, seed=0x5EED2026, **kw):
    """Block `rank` of the global ray array: rank r draws the r-th jumped
    Philox stream, so the global array is the same whatever the world size
    that later shards it in blocks of n_per_rank."""
    from . import synth
    return synth.uniform_rays(n_per_rank, lat_range, lon_range, seed=seed, jump=rank, **kw)


def tally_layout(n_media: int, n_bins: int):
    """Slices of the int64 tally vector: hits[n_media + 1] (final medium -1 ..
    n_media-1), histogram[n_bins + 1] (last = overflow), then total steps."""
    h = slice(0, n_media + 1)
    g = slice(n_media + 1, n_media + 1 + n_bins + 1)
    return h, g, n_media + 1 + n_bins + 1, n_media + n_bins + 3


def all_reduce_tally(tally, world: int):
    """Sum the tally over ranks in place (no-op for one rank)."""
    if world > 1:
        import torch.distributed as dist
        if tally.is_cuda and dist.get_backend() != "nccl":
            host = tally.cpu()          # rehearsal over gloo: reduce on the host
            dist.all_reduce(host)
            tally.copy_(host)
        else:
            dist.all_reduce(tally)
    return tally


def tally_reference(index, length, n_media, n_bins, length_max):
    """numpy statement of turtle_amd_tally_n, for tests."""
    hits = np.bincount(np.asarray(index)[:, 0] + 1, minlength=n_media + 1)[: n_media + 1]
    t = np.asarray(length) * (n_bins / length_max)
    b = np.full(t.shape, n_bins, dtype=np.int64)
    ok = (t >= 0) & (t < n_bins)
    b[ok] = t[ok].astype(np.int64)
    return hits.astype(np.int64), np.bincount(b, minlength=n_bins + 1).astype(np.int64)
