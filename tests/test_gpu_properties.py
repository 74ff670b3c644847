"""Parity at the BASELINE sizes (C2: 1 M rays through the 3601^2 tile), where
the reference's outputs are not stored: the CPU restatement on all host cores
is the checker, plus size-independent properties of a trace."""
import os

import numpy as np
import pytest

import turtle_amd as TA
from oracle import ffi as O
from turtle_amd import sharding, synth

import amd_build as B
import terrains as T

pytestmark = pytest.mark.gpu

N = 1_000_000
HAND_OVER_MOVED = 0     # (logged: none of creep_probe.py's 2 x 40 000 rays changes its step count with the hand-over)


@pytest.fixture(scope="module")
def c2(tmp_path_factory):
    tmp = tmp_path_factory.mktemp("c2")
    tile = B.hgt_tile(tmp)
    st = TA.Stepper()
    st.add_map(tile, 0.0)
    lat, lon, az, el = sharding.rank_rays(N, 0, (45.0, 46.0), (3.0, 4.0))
    pos, di = st.position(lat, lon, 500.0)
    assert (di == 0).all()
    d = TA.ecef_from_horizontal(lat, lon, az, el)
    yield dict(stepper=st, pos=pos, dir=d, map=tile)
    st.destroy()
    tile.destroy()


@pytest.mark.parametrize("mode", ["fast", "strict"])
def test_c2_full_size_against_cpu_oracle(c2, mode):
    import os
    nodes, geo = T.hgt_oracle()
    ref = geo.trace(c2["pos"], c2["dir"], threads=max(1, min(16, os.cpu_count() or 1)))
    TA.set_math(mode)
    try:
        t = c2["stepper"].trace(c2["pos"].copy(), c2["dir"])
    finally:
        TA.set_math("fast")
    rel = np.abs(t["length"] - ref["length"]) / np.maximum(ref["length"], 1e-300)
    flipped = t["index"][:, 0] != ref["index"][:, 0]
    # the bar: identical medium and 1e-6 on the path length.  A ray that skims
    # the surface for its last centimetre-steps (clearance ~1e-9 m) may take its
    # last decision either way -- another medium, or the crossing found one
    # minimum step earlier or later: allow 1e-5 of the rays, and report them.
    grazing = np.flatnonzero(flipped | (rel > 1e-6))
    ok = np.ones(N, dtype=bool)
    ok[grazing] = False
    dsteps = np.abs(t["n_steps"] - ref["n_steps"])
    print(f"[{mode}] 1M rays: {grazing.size} grazing rays ({int(flipped.sum())} with a "
          f"different medium, {int((~flipped & (rel > 1e-6)).sum())} beyond 1e-6 in length, "
          f"worst |dL| = {np.abs(t['length'] - ref['length'])[grazing].max() if grazing.size else 0.:.2e} m); "
          f"the others: max |dL|/L = {rel[ok].max():.2e}, median {np.median(rel[ok]):.1e}; "
          f"rays with a different step count: {int((dsteps != 0).sum())}, "
          f"steps GPU {int(t['n_steps'].sum())} CPU {int(ref['n_steps'].sum())}")
    assert grazing.size == 0
    assert rel[ok].max() <= 1e-6
    # a grazing ray ends within a few minimum steps (1e-2 m) of the reference's end
    assert np.abs(t["length"] - ref["length"])[grazing].max(initial=0.) < 0.1
    # (logged: 4 rays of the million in fast arithmetic, 1 in strict, take a step more or less)
    assert (dsteps[ok] <= 1).all() and (dsteps != 0).sum() <= 10


def test_c2_properties(c2):
    st, pos0, d = c2["stepper"], c2["pos"], c2["dir"]
    t = st.trace(pos0.copy(), d)
    s = st.trace_stats()
    # the device's own totals agree with the per-ray outputs
    assert s["rays"] == N and s["steps"] == int(t["n_steps"].sum()) and s["capped"] == 0
    assert 1.0 < s["samples"] / s["steps"] < 1.2
    # a ray is a straight line: final = origin + direction * path length
    err = np.abs(t["position"] - (pos0 + d * t["length"][:, None])).max()
    assert err < 1e-9 * max(1000, int(t["n_steps"].max())) , err  # 1 ulp of 6.4e6 m per step
    # every ray ended by changing medium: hit the ground (0) or left the tile (-1)
    assert set(np.unique(t["index"][:, 0])) <= {-1, 0}
    assert (t["n_steps"] >= 1).all() and (t["length"] > 0).all()
    # deterministic: the lane/wave a ray ran on does not matter
    t2 = st.trace(pos0.copy(), d)
    for k in ("index", "length", "n_steps", "position"):
        assert np.array_equal(t[k], t2[k]), k
    # sharding invariance: two halves == the whole (what multi-GPU relies on)
    h = N // 2
    a = st.trace(pos0[:h].copy(), d[:h])
    b = st.trace(pos0[h:].copy(), d[h:])
    assert np.array_equal(np.concatenate([a["length"], b["length"]]), t["length"])
    assert np.array_equal(np.concatenate([a["index"], b["index"]]), t["index"])
    # re-sampling the final position of a hit: the stored medium is the one the
    # reference's cache would hold; the boundary is within 1e-8 m along the ray
    # a hit ends ON the surface: |altitude - ground| there is below the 1e-8 m
    # bisection width (times the slope of the ray), whichever side it fell
    hit = np.flatnonzero(t["index"][:, 0] == 0)[:20000]
    o = st.step(t["position"][hit].copy(), None)
    ground = np.where(o["index"][:, 0] == 0, o["elevation"][:, 1], o["elevation"][:, 0])
    assert np.abs(o["altitude"] - ground).max() < 1e-7
    # tally of the whole == sum of the tallies of the halves (integer exact)
    hits, hist = TA.tally(t["index"], t["length"], 2, 1024, 65536.0)
    h1, g1 = TA.tally(a["index"], a["length"], 2, 1024, 65536.0)
    h2, g2 = TA.tally(b["index"], b["length"], 2, 1024, 65536.0, h1, g1)
    assert np.array_equal(hits, h2) and np.array_equal(hist, g2) and hits.sum() == N


def test_creep_loop_changes_no_bit(tmp_path):
    """The lean steps of the lined pass take the samples the general
    iteration would take, with the same functions on the same values: a batch with
    thousands-of-steps rays gives the same bits with the lean loop off
    (TURTLE_AMD_CREEP_LANES=0 and TURTLE_AMD_DENSE_GO=0), engaging in sparse waves only,
    at its defaults, and at other thresholds -- through one map and through a regular
    stack."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    results = {}
    for lanes, go in (("0", "0"), ("8", "0"), ("8", "24"), ("64", "0"), ("8", "48"), ("0", "8")):
        out = os.path.join(tmp_path, f"lanes{lanes}_{go}.npz")
        work = os.path.join(tmp_path, f"work{lanes}_{go}")
        # (each lane keeps its ray here: with the block's pool a lean wave steps whatever these say)
        env = dict(os.environ, TURTLE_AMD_CREEP_LANES=lanes, TURTLE_AMD_DENSE_GO=go, TURTLE_AMD_POOL="0")
        subprocess.run([sys.executable, os.path.join(here, "creep_probe.py"), out, work],
                       check=True, env=env, timeout=300)
        results[(lanes, go)] = dict(np.load(out))
    base = results[("0", "0")]
    assert base["map_n_steps"].max() > 2000 and base["stack_n_steps"].max() > 2000
    for which, r in results.items():
        for key, ref in base.items():
            assert np.array_equal(r[key], ref), (which, key)


def test_ray_pool_changes_no_bit(tmp_path):
    """Round 4: the four waves of a block of the lined pass exchange rays through a pool in LDS, so
    that lean steps run with every lane going and general iterations for a full wave of rays that
    need one (device.hip, RayPool).  Which lane or wave takes a step of a ray changes no bit of it:
    the batch of test_creep_loop_changes_no_bit -- rays of thousands of steps, one map and a regular
    stack -- with the pool off (and the lean loop off: the closed form's and the line's general
    iterations only), with the pool on at its defaults, and on with other thresholds of the loop
    that runs once the pass is down to its last rays."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    results = {}
    for tag, env in (("off", dict(TURTLE_AMD_POOL="0", TURTLE_AMD_CREEP_LANES="0", TURTLE_AMD_DENSE_GO="0")),
                     ("off, lean loop at its defaults", dict(TURTLE_AMD_POOL="0")),
                     ("on", dict(TURTLE_AMD_POOL="1")),
                     ("on, dense", dict(TURTLE_AMD_POOL="1", TURTLE_AMD_CREEP_LANES="64")),
                     ("on, no lean loop at the end", dict(TURTLE_AMD_POOL="1", TURTLE_AMD_CREEP_LANES="0",
                                                         TURTLE_AMD_DENSE_GO="0"))):
        out = os.path.join(tmp_path, f"pool{len(results)}.npz")
        subprocess.run([sys.executable, os.path.join(here, "creep_probe.py"), out,
                        os.path.join(tmp_path, f"work{len(results)}")],
                       check=True, env=dict(os.environ, **env), timeout=300)
        results[tag] = dict(np.load(out))
    base = results["off"]
    assert base["map_n_steps"].max() > 2000 and base["stack_n_steps"].max() > 2000
    for which, r in results.items():
        for key, ref in base.items():
            assert np.array_equal(r[key], ref), (which, key)


def test_two_batches_in_flight_give_the_same_bits(c2):
    """One stepper is one stream of calls (as one turtle_stepper is one thread's in the
    reference); two of them over the same map, each on a stream of its own, take batches in
    turn -- the few long rays a trace ends with then step beside the bulk of the next batch
    (bench.py's `in_flight`).  Same bits as one after the other, and as the fixture's stepper."""
    import torch
    dev = torch.device("cuda", 0)
    pos0 = torch.as_tensor(c2["pos"], device=dev)
    d = torch.as_tensor(c2["dir"], device=dev)
    ref = c2["stepper"].trace(c2["pos"].copy(), c2["dir"])
    steppers, streams = [], []
    try:
        for k in range(2):
            if k == 0:
                st = TA.Stepper()
                st.add_map(c2["map"], 0.0)
            else:
                st = c2["stepper"].clone()      # turtle_amd_stepper_clone
            steppers.append(st)
            streams.append(torch.cuda.Stream(device=dev))
        torch.cuda.synchronize()
        TA.set_in_flight(2)                   # the hint (a kernel's share of a CU): no result depends on it
        assert TA.get_in_flight() == 2
        outs = [None, None]
        bufs = [pos0.clone(), pos0.clone()]
        for k in range(6):                    # batches 0 .. 5, two in flight at any time
            w = k % 2
            torch.cuda.set_stream(streams[w])
            TA.set_stream(streams[w])
            bufs[w].copy_(pos0)
            outs[w] = steppers[w].trace(bufs[w], d)
        torch.cuda.synchronize()
        for out in outs:
            for key in ("index", "length", "n_steps", "position"):
                assert np.array_equal(out[key].cpu().numpy(), ref[key]), key
    finally:
        TA.set_in_flight(1)
        torch.cuda.set_stream(torch.cuda.default_stream(dev))
        TA.set_stream(None)
        for st in steppers:
            st.destroy()


def test_sorted_hand_over_changes_no_bit(tmp_path):
    """Phase A lists the rays it hands over from both ends of the list -- the ones heading for the
    ground at the back, which the lined pass reads last.  Where a ray is listed
    changes when it is traced, not what comes out: the same bits with the list unsorted
    (TURTLE_AMD_SORT_LONG=0), at the default, with every ray at the back and with a figure that
    splits the batch in the middle.  Round 4: between the passes the front of the list is ORDERED,
    the shallowest rays first (a radix sort on a one-byte key: TURTLE_AMD_SORT_KEY) -- on, off,
    and on with the two-ended list off or all at the back: the same bits again."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    results = {}
    # ... and (round 4) the trace may take its rays in the order of WHERE THEY START -- a raster of
    # cells over the terrain, a radix sort, the passes working in arrays of the library's own and
    # writing each ray's results to its place in the caller's (TURTLE_AMD_SPATIAL): on and off
    for figure, key, spatial in (("0", "0", "0"), ("120", "0", "0"), ("1000000", "0", "0"), ("400", "0", "0"),
                                 ("120", "1", "0"), ("0", "1", "0"), ("1000000", "1", "0"), ("400", "1", "0"),
                                 ("120", "1", "1"), ("120", "0", "1"), ("0", "0", "1")):
        out = os.path.join(tmp_path, f"sort{figure}_{key}_{spatial}.npz")
        env = dict(os.environ, TURTLE_AMD_SORT_LONG=figure, TURTLE_AMD_SORT_KEY=key, TURTLE_AMD_SPATIAL=spatial)
        subprocess.run([sys.executable, os.path.join(here, "creep_probe.py"), out,
                        os.path.join(tmp_path, f"work{figure}_{key}_{spatial}")], check=True, env=env, timeout=300)
        results[(figure, key, spatial)] = dict(np.load(out))
    base = results[("0", "0", "0")]
    assert base["map_n_steps"].max() > 2000 and base["stack_n_steps"].max() > 2000
    for figure, r in results.items():
        for key, ref in base.items():
            assert np.array_equal(r[key], ref), (figure, key)


def test_hand_over_step_moves_no_result(tmp_path):
    """A ray goes on its line at a fixed step count (TURTLE_AMD_PARK: 32 by default for one map and
    for a stack).  That count decides which arithmetic takes which sample -- not
    what comes out: the same batch (shallow rays, thousands of steps) with the hand-over at step
    16, 32 (the default), 512 and never (0: one phase, the closed form throughout) ends in the same
    media, and its path lengths agree to 1e-7 (the bound asked of parity is 1e-6)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    results = {}
    for park in ("0", "16", "32", "512"):
        out = os.path.join(tmp_path, f"park{park}.npz")
        env = dict(os.environ, TURTLE_AMD_PARK=park)
        subprocess.run([sys.executable, os.path.join(here, "creep_probe.py"), out,
                        os.path.join(tmp_path, f"work{park}")], check=True, env=env, timeout=300)
        results[park] = dict(np.load(out))
    base = results["0"]
    assert base["map_n_steps"].max() > 2000 and base["stack_n_steps"].max() > 2000
    for park, r in results.items():
        for tag in ("map", "stack"):
            assert np.array_equal(r[f"{tag}_index"], base[f"{tag}_index"]), (park, tag)
            rel = np.abs(r[f"{tag}_length"] - base[f"{tag}_length"]) / np.maximum(base[f"{tag}_length"], 1.0)
            assert rel.max() < 1e-7, (park, tag, rel.max())
            # a grazing ray may take a step more or less; the others take the same number
            moved = int((r[f"{tag}_n_steps"] != base[f"{tag}_n_steps"]).sum())
            print(f"hand-over at {park}, {tag}: {moved} of {base[f'{tag}_n_steps'].size} rays take another step count")
            assert moved <= HAND_OVER_MOVED, (park, tag, moved)


def test_odd_batch_sizes_through_every_optional_path():
    """The trace's optional machinery -- the block's ray pool, the ordered hand-over, the rays in
    the order of where they start -- is switched on by batch size; here ALL of it is forced on for
    batches of 1 to 100 003 rays (odd sizes: lists and the sorts' rooms that end on no boundary, waves
    with one ray), one map and a 2 x 2 stack, against the same batches with all of it off: the same
    bits (scripts/exp_odd_sizes.py; up to 700 001 rays there)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "scripts", "exp_odd_sizes.py")
    run = subprocess.run([sys.executable, script, "1", "65", "257", "4097", "100003"], capture_output=True,
                         text=True, timeout=600)
    print(run.stdout[-1500:])
    assert run.returncode == 0, run.stderr[-1500:]
    assert run.stdout.count("same bits") == 5
