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
BUDGET = 16   # 24 tiles do not fit


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


@pytest.mark.parametrize("budget", [BUDGET, 2])
@pytest.mark.parametrize("math", ["strict", "fast"])
def test_trace_paged(mosaic_dir, math, budget):
    TA.set_math(math)
    try:
        full, paged = TA.Stack(mosaic_dir, 0), TA.Stack(mosaic_dir, budget)
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
        assert paged.resident <= budget
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


def test_reference_stack_test_tile_counts(tmp_path):
    """The reference's own test of its stack [ref tests/test-turtle.c:628-690]: four
    tiles of zeros, stack_size 3, scalar queries; the number of tiles in memory after
    each one, clear and load."""
    d = str(tmp_path / "four")
    os.makedirs(d)
    flat = np.zeros((1201, 1201), dtype=np.int16)
    for la, lo in ((45, 2), (45, 3), (46, 2), (46, 3)):
        with open(os.path.join(d, synth.hgt_name(la, lo, 1201)), "wb") as f:
            f.write(synth.hgt_bytes(flat))
    s = TA.Stack(d, 3)
    assert s.resident == 0
    for (la, lo), count in (((45.5, 3.5), 1), ((45.0, 3.5), 1), ((46.5, 3.5), 2), ((45.0, 3.5), 2)):
        z, inside = s.elevation_scalar(la, lo)
        assert (z, inside, s.resident) == (0.0, 1, count), (la, lo)
    z, inside = s.elevation_scalar(45.5, 4.5)
    assert inside == 0 and s.resident == 2
    for (la, lo), count in (((45.5, 2.5), 3), ((46.5, 2.5), 3)):
        z, inside = s.elevation_scalar(la, lo)
        assert (z, inside, s.resident) == (0.0, 1, count), (la, lo)
    s.clear()
    assert s.resident == 0
    s.elevation_scalar(45.5, 2.5)
    assert s.resident == 1
    s.load()
    assert s.resident == 3
    s.clear()
    s.load()
    s.load()
    assert s.resident == 3
    s.destroy()
    s = TA.Stack(d, 0)
    s.load()
    assert s.resident == 4
    s.destroy()


@pytest.mark.parametrize("size", [1, 3, 8])
@pytest.mark.parametrize("math", ["strict", "fast"])
def test_trace_paged_against_the_oracle(mosaic_dir, math, size):
    """Traces through a stack that keeps `size` tiles (the reference's test uses 3)
    against the CPU restatement with every tile in memory -- directly, not through the
    all-resident GPU stack: rays that go from tile to tile, rays whose bisection
    straddles a seam, rays over the hole; and the stack is back within its size when
    the call returns."""
    import terrains as T
    from oracle import ffi as O
    TA.set_math(math)
    try:
        geo = T.mosaic_oracle(TILES, N_TILE, 40, 5, 5, 5)
        paged = TA.Stack(mosaic_dir, size)
        sp = TA.Stepper()
        sp.add_stack(paged, 0.0)
        rng = np.random.default_rng(17)
        n = 3000
        lat, lon = rng.uniform(40.1, 44.9, n), rng.uniform(5.1, 9.9, n)
        # a third of the rays start within 30 m of a seam, heading across it
        k = n // 3
        lat[:k] = np.round(lat[:k]) + rng.uniform(-3e-4, 3e-4, k)
        az, el = rng.uniform(0, 360, n), rng.uniform(-12.0, 2.0, n)
        p, di = geo.position(lat, lon, 400.0)
        keep = di == 0
        p, d = p[keep], O.ecef_from_horizontal(lat, lon, az, el)[keep]
        ref = geo.trace(p, d, threads=8)
        pg, dg = sp.position(lat, lon, 400.0)
        assert np.array_equal(dg == 0, keep) and np.abs(pg[keep] - p).max() < 1e-8   # OCML vs glibc sin/cos
        assert paged.resident <= size
        t = sp.trace(p.copy(), d)
        assert paged.resident <= size
        flipped = t["index"][:, 0] != ref["index"][:, 0]
        rel = np.abs(t["length"] - ref["length"]) / np.maximum(ref["length"], 1e-300)
        assert flipped.sum() == 0 and (rel[~flipped] <= 1e-6).all(), (int(flipped.sum()), rel[~flipped].max())
        if math == "strict":
            assert (np.abs(t["n_steps"] - ref["n_steps"])[~flipped] <= 1).all()
        # single steps page too, and agree with the oracle's
        o = geo.step(p, d)
        g = sp.step(p.copy(), d)
        same = g["index"][:, 0] == o["index"][:, 0]
        assert (~same).sum() == 0
        assert np.abs(g["step"][same] - o["step"][same]).max() <= 1e-6 * max(1.0, o["step"][same].max())
        assert paged.resident <= size
        sp.destroy()
        paged.destroy()
    finally:
        TA.set_math("fast")


def test_client_on_a_locked_stack(tmp_path):
    """SURVEY a10 [ref client.c:99-188, tests/test-turtle.c:697-775]: a locked stack of
    size 1, its client gives the stack's answers (lock and unlock called in pairs
    around every tile change), the PATH_ERROR text for a missing tile; a stepper over
    the locked stack creates its own client and traces as over an unlocked one."""
    import ctypes as C
    from turtle_amd import binding as Bn
    d = str(tmp_path / "four")
    os.makedirs(d)
    for la, lo in ((45, 2), (45, 3), (46, 2), (46, 3)):
        synth.write_hgt(d, la, lo, N_TILE)
    LOCKER = C.CFUNCTYPE(C.c_int)
    calls = []
    lock = LOCKER(lambda: (calls.append("lock"), 0)[1])
    unlock = LOCKER(lambda: (calls.append("unlock"), 0)[1])
    L = TA.lib()
    locked, plain = TA.Stack(d, 1, lock, unlock), TA.Stack(d, 0)
    plain.load()
    h = C.c_void_p()
    Bn._check(L.turtle_client_create(C.byref(h), locked.h))

    def client_elevation(la, lo, want_inside=True):
        z, inside = C.c_double(-1.0), C.c_int(-1)
        rc = L.turtle_client_elevation(h, C.c_double(la), C.c_double(lo), C.byref(z),
                                       C.byref(inside) if want_inside else None)
        Bn._check(rc)
        return z.value, inside.value

    for la, lo in ((45.5, 3.5), (45.0, 3.5), (46.5, 3.5), (45.0, 3.5), (45.5, 2.5), (46.5, 2.5),
                   (45.73, 2.11), (46.0, 3.0)):
        before = len(calls)
        assert client_elevation(la, lo) == plain.elevation_scalar(la, lo), (la, lo)
        new = calls[before:]
        assert new.count("lock") == new.count("unlock") and locked.resident <= 1
    assert "lock" in calls
    assert client_elevation(45.5, 4.5)[1] == 0                       # outside: no error with `inside`
    Bn._check(L.turtle_client_clear(h))
    assert client_elevation(45.5, 3.5) == plain.elevation_scalar(45.5, 3.5)
    with pytest.raises(TA.TurtleError) as e:                          # [ref test-turtle.c:764-771]
        client_elevation(45.5, 4.5, want_inside=False)
    assert e.value.name == "PATH_ERROR"
    assert "turtle_client_elevation" in str(e.value) and f"missing elevation data in `{d}'" in str(e.value)
    Bn._check(L.turtle_client_destroy(C.byref(h)))
    # the stepper's own client: traces over the locked stack == over the unlocked one
    sl, su = TA.Stepper(), TA.Stepper()
    sl.add_stack(locked, 0.0)
    su.add_stack(plain, 0.0)
    lat, lon, az, el = synth.uniform_rays(4000, (45.0, 47.0), (2.0, 4.0), seed=3)
    p, di = su.position(lat, lon, 300.0)
    pl, dl = sl.position(lat, lon, 300.0)
    assert np.array_equal(p, pl) and np.array_equal(di, dl) and (di == 0).all()
    dirs = TA.ecef_from_horizontal(lat, lon, az, el)
    TA.set_math("strict")
    try:
        before = len(calls)
        a, b = su.trace(p.copy(), dirs), sl.trace(p.copy(), dirs)
        new = calls[before:]
    finally:
        TA.set_math("fast")
    for key in ("index", "length", "n_steps", "position"):
        assert np.array_equal(a[key], b[key]), key
    assert new.count("lock") == new.count("unlock") >= 1 and locked.resident <= 1
    for o in (sl, su, locked, plain):
        o.destroy()


def test_paged_trace_on_device_arrays_without_outputs(mosaic_dir):
    """A paged trace keeps each waiting ray's path length and step count in arrays of
    its own when the caller asks for neither -- also when the rays are device arrays
    (nothing staged, the library's arena not yet sized) and the batch is large."""
    import torch
    full, paged = TA.Stack(mosaic_dir, 0), TA.Stack(mosaic_dir, 4)
    full.load()
    sf, sp = TA.Stepper(), TA.Stepper()
    sf.add_stack(full, 0.0)
    sp.add_stack(paged, 0.0)
    n = 600_000
    rng = np.random.default_rng(5)
    lat, lon = rng.uniform(40.1, 41.9, n), rng.uniform(5.1, 6.9, n)
    az, el = rng.uniform(0, 360, n), rng.uniform(-20.0, -5.0, n)
    p, di = sf.position(lat, lon, 300.0)
    d = TA.ecef_from_horizontal(lat, lon, az, el)
    a = sf.trace(p.copy(), d, want=())
    tp, td = torch.as_tensor(p, device="cuda"), torch.as_tensor(d, device="cuda")
    b = sp.trace(tp, td, want=())
    TA.synchronize()
    assert b["length"] is None and b["n_steps"] is None
    assert np.array_equal(a["index"], b["index"].cpu().numpy())
    assert np.abs(a["position"] - b["position"].cpu().numpy()).max() < 1e-5
    assert paged.resident <= 4
    for o in (sf, sp, full, paged):
        o.destroy()


def test_threads_share_a_map_and_a_locked_stack(tmp_path):
    """The reference's threading pattern [ref examples/example-pthread.c:66-125]: a
    stepper (and a client) per thread over a shared map / a shared stack with lock
    callbacks whose tiles come and go (stack_size 2 over 4 files).  Each thread's
    device context (stream, scratch arena, tables) is its own: the answers are those
    of one thread doing the same calls, bit for bit."""
    import ctypes as C
    import threading
    from turtle_amd import binding as Bn
    d = str(tmp_path / "four")
    os.makedirs(d)
    for la, lo in ((45, 2), (45, 3), (46, 2), (46, 3)):
        synth.write_hgt(d, la, lo, N_TILE)
    tile = TA.Map.load(os.path.join(d, synth.hgt_name(45, 3, N_TILE)))
    mutex = threading.Lock()
    LOCKER = C.CFUNCTYPE(C.c_int)
    lock = LOCKER(lambda: (mutex.acquire(), 0)[1])
    unlock = LOCKER(lambda: (mutex.release(), 0)[1])
    shared = TA.Stack(d, 2, lock, unlock)
    L = TA.lib()
    n_threads, n = 4, 3000

    def work(seed, out):
        try:
            rng = np.random.default_rng(seed)
            # (a) a stepper of its own over the shared map: batches and scalar steps
            sm = TA.Stepper()
            sm.add_map(tile, 0.0)
            lat, lon = rng.uniform(45.1, 45.9, n), rng.uniform(3.1, 3.9, n)
            az, el = rng.uniform(0, 360, n), rng.uniform(-10.0, -1.0, n)
            p, _ = sm.position(lat, lon, 300.0)
            dirs = TA.ecef_from_horizontal(lat, lon, az, el)
            out["map_trace"] = sm.trace(p.copy(), dirs)
            q, steps = p[0].copy(), []
            for _ in range(40):
                r = sm.step_scalar(q, dirs[0])
                q = r["position"]
                steps.append((r["step"], r["altitude"], int(r["index"][0])))
            out["map_steps"] = steps
            # (b) a client and a stepper of its own over the shared, locked stack
            h = C.c_void_p()
            Bn._check(L.turtle_client_create(C.byref(h), shared.h))
            zs = []
            for la, lo in zip(rng.uniform(45.0, 47.0, 60), rng.uniform(2.0, 4.0, 60)):
                z, inside = C.c_double(), C.c_int()
                Bn._check(L.turtle_client_elevation(h, C.c_double(la), C.c_double(lo), C.byref(z),
                                                    C.byref(inside)))
                zs.append((z.value, inside.value))
            out["client"] = zs
            Bn._check(L.turtle_client_destroy(C.byref(h)))
            # strict arithmetic (this thread's own setting): a ray that waited for a tile
            # carries on with the same bits whatever the order the tiles came in
            TA.set_math("strict")
            ss = TA.Stepper()
            ss.add_stack(shared, 0.0)
            lat, lon = rng.uniform(45.1, 46.9, n), rng.uniform(2.1, 3.9, n)
            p, di = ss.position(lat, lon, 300.0)
            out["stack_di"] = di
            out["stack_trace"] = ss.trace(p.copy(), TA.ecef_from_horizontal(lat, lon, az, el))
            sm.destroy()
            ss.destroy()
        except BaseException as e:      # noqa: BLE001 -- reported by the main thread
            out["error"] = repr(e)
        finally:
            L.turtle_amd_thread_release()

    alone = [dict() for _ in range(n_threads)]
    for i in range(n_threads):
        t = threading.Thread(target=work, args=(100 + i, alone[i]))   # one at a time
        t.start()
        t.join()
    together = [dict() for _ in range(n_threads)]
    threads = [threading.Thread(target=work, args=(100 + i, together[i])) for i in range(n_threads)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    # the same traces through a stack with every tile in memory (strict arithmetic:
    # bit for bit what a paged stack gives)
    full = TA.Stack(d, 0)
    full.load()
    sfull = TA.Stepper()
    sfull.add_stack(full, 0.0)
    TA.set_math("strict")
    try:
        for i, (a, b) in enumerate(zip(alone, together)):
            rng = np.random.default_rng(100 + i)
            rng.uniform(45.1, 45.9, n), rng.uniform(3.1, 3.9, n)
            az, el = rng.uniform(0, 360, n), rng.uniform(-10.0, -1.0, n)
            rng.uniform(45.0, 47.0, 60), rng.uniform(2.0, 4.0, 60)
            lat, lon = rng.uniform(45.1, 46.9, n), rng.uniform(2.1, 3.9, n)
            p, di = sfull.position(lat, lon, 300.0)
            ref = sfull.trace(p.copy(), TA.ecef_from_horizontal(lat, lon, az, el))
            for tag, got in (("alone", a), ("together", b)):
                if "error" in got:
                    continue
                for k in ("index", "length", "n_steps", "position"):
                    bad = np.flatnonzero(np.any(np.atleast_2d(ref[k].T != got["stack_trace"][k].T), axis=0))
                    assert bad.size == 0, (tag, i, k, bad.size, bad[:6], ref["index"][bad[:6]].tolist(),
                                           got["stack_trace"]["index"][bad[:6]].tolist(),
                                           ref["n_steps"][bad[:6]].tolist(),
                                           got["stack_trace"]["n_steps"][bad[:6]].tolist())
    finally:
        TA.set_math("fast")
        sfull.destroy()
        full.destroy()
    for a, b in zip(alone, together):
        assert "error" not in a and "error" not in b, (a.get("error"), b.get("error"))
        assert a["map_steps"] == b["map_steps"] and a["client"] == b["client"]
        assert np.array_equal(a["stack_di"], b["stack_di"])
        for key in ("map_trace", "stack_trace"):
            for k in ("index", "length", "n_steps", "position"):
                bad = np.flatnonzero(np.any(np.atleast_2d(a[key][k].T != b[key][k].T), axis=0))
                assert bad.size == 0, (key, k, bad.size, bad[:8], a[key]["index"][bad[:8]],
                                       b[key]["index"][bad[:8]], a[key]["n_steps"][bad[:8]],
                                       b[key]["n_steps"][bad[:8]], a["stack_di"][bad[:8]])
    assert shared.resident <= 2 and not mutex.locked()
    shared.destroy()
    tile.destroy()


def test_a_tile_that_comes_back_is_not_read_again_unless_its_file_changed(tmp_path):
    """Round 4: the page-locked buffers tiles are read into keep the nodes of the tile they last
    held, and a tile that comes back while they are there is copied from the buffer instead of
    read from its file (a batch over a stack smaller than its ground goes to and fro between two
    sets of tiles).  Same answers as a stack that holds everything -- host nodes (turtle_map_node
    of the scalar path) and HBM copy alike -- and a file that was rewritten meanwhile is READ."""
    import time
    d = str(tmp_path / "grid")
    for la, lo in ((45, 3), (45, 4), (46, 3)):
        synth.write_hgt(d, la, lo, N_TILE)
    full, small = TA.Stack(d, 0), TA.Stack(d, 1)
    full.load()
    rng = np.random.default_rng(3)
    boxes = {"a": (45.0, 3.0), "b": (45.0, 4.0), "c": (46.0, 3.0)}
    pts = {k: (rng.uniform(la + 0.01, la + 0.99, 2000), rng.uniform(lo + 0.01, lo + 0.99, 2000))
           for k, (la, lo) in boxes.items()}
    # to and fro: every tile is dropped and comes back several times
    for k in "abcabacbca":
        z0, in0 = full.elevation(*pts[k])
        z1, in1 = small.elevation(*pts[k])
        assert in0.all() and np.array_equal(in0, in1) and np.array_equal(z0, z1), k
        assert small.resident == 1
        # the host's copy of a tile that came from a buffer is the file's too (scalar path)
        TA.set_scalar("host")
        try:
            zs, _ = small.elevation_scalar(float(pts[k][0][0]), float(pts[k][1][0]))
        finally:
            TA.set_scalar("device")
        assert zs == z0[0]
    # tile a's file is rewritten while tile b is the one in memory: a is read again
    z_b, _ = small.elevation(*pts["b"])
    nodes = synth.srtm_like_nodes(45, 3, N_TILE)
    time.sleep(0.02)
    with open(os.path.join(d, synth.hgt_name(45, 3, N_TILE)), "wb") as f:
        f.write(synth.hgt_bytes((nodes + 100).astype(nodes.dtype)))
    z_old, _ = full.elevation(*pts["a"])
    z_new, inside = small.elevation(*pts["a"])
    assert inside.all() and np.allclose(z_new, z_old + 100.0, atol=1e-9)
    full.destroy()
    small.destroy()
