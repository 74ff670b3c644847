"""C3 and C5 at their BASELINE sizes (missing #5): 10 M rays through the 4 x 4
mosaic of 3601^2 tiles behind a stack [ref src/turtle/stack.c:300-361], and 10 M scattering
rays x 256 steps over the 10 x 10 mosaic of GeoTIFF tiles [the exit detection of ref
tests/test-turtle.c:872-883 at scale: rays leave through the rim].  The reference's outputs are
not stored at these sizes: a 100 000-ray (C5: 50 000-ray) sample goes through the CPU
restatement on the host cores -- what bench.py does in-run -- and the whole batch through the
size-independent properties of a trace (determinism, halves == whole, tally of tallies)."""
import os

import numpy as np
import pytest

import turtle_amd as TA
from turtle_amd import sharding

pytestmark = pytest.mark.gpu

N = 10_000_000
SEED = 0x5EED2026


@pytest.fixture(autouse=True)
def one_stream():
    """torch's copies and the library's launches on ONE stream, as bench.py has them"""
    import torch
    stream = torch.cuda.Stream(device=0)
    with torch.cuda.stream(stream):
        TA.set_stream(stream)
        yield
        torch.cuda.synchronize()
    TA.set_stream(None)


def _terrain(tiles, fmt):
    import bench
    env = {"world": 1, "rank": 0}
    return bench.Terrain(TA, tiles, True, env, 0, fmt=fmt), bench


def _rays(terrain, n):
    import torch
    lat, lon, az, el = sharding.rank_rays(n, 0, terrain.lat_range, terrain.lon_range)
    dev = torch.device("cuda", 0)
    t_lat, t_lon, t_az, t_el = (torch.as_tensor(v, device=dev) for v in (lat, lon, az, el))
    pos, di = terrain.stepper.position(t_lat, t_lon, 500.0)
    assert int((di != 0).sum()) == 0
    return pos, TA.ecef_from_horizontal(t_lat, t_lon, t_az, t_el)


def test_c3_at_full_size():
    import torch
    terrain, bench = _terrain((45, 3, 4, 4), "hgt")
    try:
        st = terrain.stepper
        pos0, d = _rays(terrain, N)
        dev = pos0.device

        def trace(lo, hi):
            pos = pos0[lo:hi].clone()
            index = torch.empty((hi - lo, 2), dtype=torch.int32, device=dev)
            length = torch.empty(hi - lo, dtype=torch.float64, device=dev)
            nsteps = torch.empty(hi - lo, dtype=torch.int32, device=dev)
            st.trace_into(pos, d[lo:hi].contiguous(), index, length, nsteps)
            return dict(position=pos, index=index, length=length, n_steps=nsteps)

        t = trace(0, N)
        s = st.trace_stats()
        assert s["rays"] == N and s["steps"] == int(t["n_steps"].sum(dtype=torch.int64)) and s["capped"] == 0
        assert st.rounds == 1                       # every tile resident: no paging round
        # a sample against the CPU restatement, all host cores: no allowance on medium or length
        m = 100_000
        ref = terrain.oracle().trace(pos0[:m].cpu().numpy(), d[:m].cpu().numpy(), local_range=0.0,
                                     threads=bench.host_cores())
        c = bench.parity_counts(t["index"][:m].cpu().numpy(), t["length"][:m].cpu().numpy(), ref["index"],
                                ref["length"], t["n_steps"][:m].cpu().numpy(), ref["n_steps"])
        print("C3, 10 M rays, first 100 000 against the CPU restatement:", c)
        assert c["medium_mismatch"] == 0 and c["beyond_1e-6"] == 0
        assert c["step_count_mismatch"] <= 10 and c["max_step_count_difference"] <= 1
        # media: the ground (0) or out of the mosaic (-1); every ray took a step
        media = torch.unique(t["index"][:, 0]).tolist()
        assert set(media) <= {-1, 0} and int(t["n_steps"].min()) >= 1
        # deterministic, and two halves == the whole (what sharding over GPUs relies on)
        t2 = trace(0, N)
        for k in ("index", "length", "n_steps", "position"):
            assert torch.equal(t[k], t2[k]), k
        h = N // 2
        a, b = trace(0, h), trace(h, N)
        for k in ("index", "length", "n_steps", "position"):
            assert torch.equal(torch.cat([a[k], b[k]]), t[k]), k
        hits, hist = TA.tally(t["index"], t["length"], 2, 1024, 65536.0)
        h1, g1 = TA.tally(a["index"], a["length"], 2, 1024, 65536.0)
        h2, g2 = TA.tally(b["index"], b["length"], 2, 1024, 65536.0, h1, g1)
        assert torch.equal(hits, h2) and torch.equal(hist, g2) and int(hits.sum()) == N
    finally:
        terrain.close()


def test_c4_at_full_size_with_the_ray_pool():
    """C4's share of one GPU (12.5 M rays over one 3601^2 tile): the size at which the lined pass's
    waves exchange rays through LDS (device.hip, RayPool: on for one map from 6 M rays).  A 100 000-ray
    sample against the CPU restatement, the device's totals, and the whole batch bit for bit against
    two halves -- each of which is ALSO above the threshold -- and against a quarter, which is below
    it and runs the kernel without the pool: the pool changes no bit at this size either."""
    import bench
    import torch
    n = 12_500_000
    env = {"world": 1, "rank": 0}
    terrain = bench.Terrain(TA, (45, 3, 1, 1), False, env, 0)
    try:
        st = terrain.stepper
        pos0, d = _rays(terrain, n)
        dev = pos0.device

        def trace(lo, hi):
            pos = pos0[lo:hi].clone()
            index = torch.empty((hi - lo, 2), dtype=torch.int32, device=dev)
            length = torch.empty(hi - lo, dtype=torch.float64, device=dev)
            nsteps = torch.empty(hi - lo, dtype=torch.int32, device=dev)
            st.trace_into(pos, d[lo:hi].contiguous(), index, length, nsteps)
            return dict(position=pos, index=index, length=length, n_steps=nsteps)

        t = trace(0, n)
        s = st.trace_stats()
        assert s["rays"] == n and s["steps"] == int(t["n_steps"].sum(dtype=torch.int64)) and s["capped"] == 0
        m = 100_000
        ref = terrain.oracle().trace(pos0[:m].cpu().numpy(), d[:m].cpu().numpy(), local_range=0.0,
                                     threads=bench.host_cores())
        c = bench.parity_counts(t["index"][:m].cpu().numpy(), t["length"][:m].cpu().numpy(), ref["index"],
                                ref["length"], t["n_steps"][:m].cpu().numpy(), ref["n_steps"])
        print("C4, 12.5 M rays (pool on), first 100 000 against the CPU restatement:", c)
        assert c["medium_mismatch"] == 0 and c["beyond_1e-6"] == 0
        assert c["step_count_mismatch"] <= 10 and c["max_step_count_difference"] <= 1
        h, q = n // 2, n // 4
        a, b, small = trace(0, h), trace(h, n), trace(0, q)       # 6.25 M each: pooled; 3.1 M: not
        for k in ("index", "length", "n_steps", "position"):
            assert torch.equal(torch.cat([a[k], b[k]]), t[k]), k
            assert torch.equal(small[k], t[k][:q]), k
    finally:
        terrain.close()


def test_c5_at_full_size():
    import torch
    terrain, bench = _terrain((40, 0, 10, 10), "tif")
    try:
        st = terrain.stepper
        pos0, _ = _rays(terrain, N)
        K = 256
        w = st.scatter(pos0.clone(), SEED, K)
        s = st.trace_stats()
        assert s["rays"] == N and s["steps"] == int(w["steps"].sum(dtype=torch.int64))
        assert int((w["steps"] < K).sum()) > 1000       # rays did leave through the rim
        assert int(w["steps"].max()) == K
        # a sample, step for step through the CPU restatement (directions: the library's Philox)
        m = 50_000
        ref_pos = pos0[:m].cpu().numpy().copy()
        geo = terrain.oracle()
        total = np.zeros(m)
        o = geo.step(ref_pos)
        alive = o["index"][:, 0] >= 0
        taken = np.zeros(m, dtype=np.int64)
        for k in range(K):
            dk = TA.isotropic(m, SEED, k, device=False)
            o = geo.step(ref_pos, dk)
            ref_pos = np.where(alive[:, None], o["position"], ref_pos)
            total += np.where(alive, o["step"], 0.0)
            taken += alive
            alive &= o["index"][:, 0] >= 0
        ref_index = np.where(alive[:, None], o["index"], -1)
        c = bench.parity_counts(w["index"][:m].cpu().numpy(), w["length"][:m].cpu().numpy(), ref_index, total,
                                w["steps"][:m].cpu().numpy(), taken)
        print("C5, 10 M rays x 256 steps, first 50 000 against the CPU restatement:", c)
        assert c["medium_mismatch"] == 0 and c["beyond_1e-6"] == 0 and c["step_count_mismatch"] == 0
        # deterministic; two halves (the second keyed from ray N/2 on) == the whole
        w2 = st.scatter(pos0.clone(), SEED, K)
        for k in ("index", "length", "steps", "position"):
            assert torch.equal(w[k], w2[k]), k
        h = N // 2
        a = st.scatter(pos0[:h].clone(), SEED, K)
        b = st.scatter(pos0[h:].clone(), SEED, K, first_ray=h)
        for k in ("index", "length", "steps", "position"):
            assert torch.equal(torch.cat([a[k], b[k]]), w[k]), k
        hits, hist = TA.tally(w["index"], w["length"], 2, 1024, 65536.0)
        h1, g1 = TA.tally(a["index"], a["length"], 2, 1024, 65536.0)
        h2, g2 = TA.tally(b["index"], b["length"], 2, 1024, 65536.0, h1, g1)
        assert torch.equal(hits, h2) and torch.equal(hist, g2) and int(hits.sum()) == N
    finally:
        terrain.close()
