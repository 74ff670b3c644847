"""Paged tile stacks on the GPU: a stack with fewer resident tiles than
the batch touches gives the results of a stack with every tile in memory.

The reference loads a tile when a query needs it and evicts the least recently
used one beyond stack_size [ref stack.c:399-450]; here the batch runs in rounds
(rays / points that meet a non-resident tile are listed, the host pages the
tiles in, the list runs again)."""
import os

import numpy as np
import pytest

import turtle_amd as TA
from turtle_amd import synth

pytestmark = pytest.mark.gpu

N_TILE = 1201
TILES = [(la, lo) for la in range(40, 45) for lo in range(5, 10) if (la, lo) != (42, 7)]
BUDGET = 16   # the smallest a stack goes (turtle_amd.h): 24 tiles do not fit


@pytest.fixture(scope="module")
def mosaic_dir(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("paged") / "grid")
    for la, lo in TILES:
        synth.write_hgt(d, la, lo, N_TILE)
    return d


def points(n, seed):
    rng = np.random.default_rng(seed)
    lat = rng.uniform(39.9, 45.1, n)     # a margin outside the directory too
    lon = rng.uniform(4.9, 10.1, n)
    # and points ON the seams between tiles, where the neighbours' boxes decide
    lat[: n // 20] = np.round(lat[: n // 20])
    lon[n // 20: n // 10] = np.round(lon[n // 20: n // 10])
    return lat, lon


def test_elevation_and_gradient_paged(mosaic_dir):
    full, paged = TA.Stack(mosaic_dir, 0), TA.Stack(mosaic_dir, BUDGET)
    full.load()
    assert full.resident == 24 and paged.resident == 0
    lat, lon = points(40000, 1)
    z0, in0 = full.elevation(lat, lon)
    z1, in1 = paged.elevation(lat, lon)
    assert np.array_equal(in0, in1) and np.array_equal(z0, z1)
    assert 0 < paged.resident <= BUDGET      # 24 tiles were wanted, 16 at most stay
    g0, g1 = full.gradient(lat, lon), paged.gradient(lat, lon)
    for a, b in zip(g0, g1):
        assert np.array_equal(a, b)
    # a second, local batch: the tiles it needs come back in
    sel = (lat > 40.2) & (lat < 40.8) & (lon > 5.2) & (lon < 5.8)
    z2, in2 = paged.elevation(lat[sel], lon[sel])
    assert np.array_equal(z2, z0[sel]) and paged.resident <= BUDGET
    # scalar drop-in calls page too
    paged.clear()
    assert paged.resident == 0
    zz, ii = paged.elevation_scalar(44.5, 9.5)
    assert ii == 1 and zz == full.elevation_scalar(44.5, 9.5)[0] and paged.resident >= 1
    full.destroy()
    paged.destroy()


@pytest.mark.parametrize("math", ["strict", "fast"])
def test_trace_paged(mosaic_dir, math):
    TA.set_math(math)
    try:
        full, paged = TA.Stack(mosaic_dir, 0), TA.Stack(mosaic_dir, BUDGET)
        full.load()
        sf, sp = TA.Stepper(), TA.Stepper()
        sf.add_stack(full, 0.0)
        sp.add_stack(paged, 0.0)
        rng = np.random.default_rng(3)
        n = 6000
        lat, lon = rng.uniform(40.1, 44.9, n), rng.uniform(5.1, 9.9, n)
        az, el = rng.uniform(0, 360, n), rng.uniform(-12.0, 2.0, n)   # long, low: tile to tile
        p0, d0 = sf.position(lat, lon, 400.0)
        p1, d1 = sp.position(lat, lon, 400.0)
        assert np.array_equal(d0, d1) and np.array_equal(p0, p1)
        keep = d0 == 0                         # not above the missing tile
        p0 = p0[keep]
        d = TA.ecef_from_horizontal(lat, lon, az, el)[keep]
        t0 = sf.trace(p0.copy(), d)
        t1 = sp.trace(p0.copy(), d)
        s1 = sp.trace_stats()
        assert paged.resident <= BUDGET
        assert np.array_equal(t0["index"], t1["index"])
        assert s1["rays"] == p0.shape[0] and s1["steps"] == int(t1["n_steps"].sum())
        if math == "strict":
            # a ray that waited for a tile carries on with the same arithmetic
            for k in ("length", "n_steps", "position"):
                assert np.array_equal(t0[k], t1[k]), k
        else:
            # in the second phase it carries on along a new line: 1e-9 m level
            assert np.array_equal(t0["n_steps"], t1["n_steps"])
            rel = np.abs(t0["length"] - t1["length"]) / np.maximum(t0["length"], 1e-300)
            assert rel.max() < 1e-9 and np.abs(t0["position"] - t1["position"]).max() < 1e-5
        # the batch did go from tile to tile
        lat1, lon1, _ = TA.ecef_to_geodetic(t1["position"])
        moved = (np.floor(lat1) != np.floor(lat[keep])) | (np.floor(lon1) != np.floor(lon[keep]))
        assert moved.sum() > 100
        # resumed traces (second medium) page as well
        r0 = sf.trace(t0["position"].copy(), d, resume_index=t0["index"])
        r1 = sp.trace(t1["position"].copy(), d, resume_index=t1["index"])
        assert np.array_equal(r0["index"], r1["index"])
        assert np.abs(r0["length"] - r1["length"]).max() < 1e-5
        for o in (sf, sp, full, paged):
            o.destroy()
    finally:
        TA.set_math("fast")


def test_single_steps_paged(mosaic_dir):
    """a scattering walk through turtle_stepper_step_n, sample handed back each step"""
    full, paged = TA.Stack(mosaic_dir, 0), TA.Stack(mosaic_dir, BUDGET)
    full.load()
    sf, sp = TA.Stepper(), TA.Stepper()
    sf.add_stack(full, 0.0)
    sp.add_stack(paged, 0.0)
    rng = np.random.default_rng(9)
    n = 20000
    lat, lon = rng.uniform(40.05, 44.95, n), rng.uniform(5.05, 9.95, n)
    p, di = sf.position(lat, lon, 150.0)
    keep = di == 0
    p = p[keep]
    a, b = sf.step(p.copy(), None), sp.step(p.copy(), None)   # positions move in place
    for k in ("index", "altitude", "elevation"):
        assert np.array_equal(a[k], b[k]), k
    for gen in range(12):
        d = TA.isotropic(p.shape[0], 11, gen, 0, device=False)
        a = sf.step(a["position"], d, resume=a)
        b = sp.step(b["position"], d, resume=b)
        for k in ("index", "position", "step", "altitude", "elevation"):
            assert np.array_equal(a[k], b[k]), (gen, k)
        assert paged.resident <= BUDGET
    for o in (sf, sp, full, paged):
        o.destroy()
