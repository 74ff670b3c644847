"""The CPU restatement (oracle/) against the REAL reference, live: where the
reference's build is present (oracle/_ref/, made by oracle/Makefile in the build
container; it also travels to the GPU box) the same rays are stepped by both and
must agree bit for bit -- medium, data index, step count, path length -- with the
exact transform (range 0) and with the reference's local approximation (range 1).
Skipped where the build is absent; the committed golden vectors (test_oracle_golden)
pin the restatement there."""
import numpy as np
import pytest

from oracle import ffi as O
from oracle import ref_ffi as R
from turtle_amd import synth

import terrains as T

pytestmark = pytest.mark.skipif(not R.driver_available(),
                                reason="oracle/_ref (the compiled reference) is not present")


@pytest.mark.parametrize("local_range", [0.0, 1.0])
def test_traces_equal_bit_for_bit(tmp_path, local_range):
    n_nodes = 1201
    path = synth.write_hgt(str(tmp_path), 45, 3, n_nodes)
    _, geo = T.hgt_oracle(45, 3, n_nodes)
    lat, lon, az, el = synth.uniform_rays(6000, (45.0, 46.0), (3.0, 4.0), seed=21)
    pos, _ = geo.position(lat, lon, 500.0)
    d = O.ecef_from_horizontal(lat, lon, az, el)
    theirs = R.trace_map(path, pos, d, local_range=local_range, threads=4)
    ours = geo.trace(pos, d, local_range=local_range, threads=4)
    assert theirs["total_steps"] == ours["total_steps"] > 500_000
    assert np.array_equal(theirs["index"], ours["index"])
    assert np.array_equal(theirs["n_steps"], ours["n_steps"])
    assert np.array_equal(theirs["length"], ours["length"])          # bit for bit
    assert np.array_equal(theirs["position"], ours["position"])
    # the cap and a ray that starts outside
    theirs = R.trace_map(path, pos[:64], d[:64], local_range=local_range, max_steps=9)
    ours = geo.trace(pos[:64], d[:64], local_range=local_range, max_steps=9)
    assert np.array_equal(theirs["n_steps"], ours["n_steps"]) and (ours["n_steps"] <= 9).all()
    far = pos[:4] * 3.0
    theirs, ours = R.trace_map(path, far, d[:4]), geo.trace(far, d[:4], local_range=1.0)
    assert (theirs["index"][:, 0] == -1).all() and np.array_equal(theirs["index"], ours["index"])


def test_walks_equal_bit_for_bit_and_the_client_memo_quirk(tmp_path):
    """C5's shape in small, live against the reference: scattering walks (a new direction at every
    step) over a mosaic that TOUCHES LONGITUDE 0, from 20 km up so that thousands of rays leave it.
    Through a stack without lock / unlock (the stepper looks the stack up itself) the restatement
    and the reference agree bit for bit on every ray -- medium, step count, path length.

    Through a LOCKED stack the reference's stepper makes a client, whose memo of "no data at this
    integer (latitude, longitude)" truncates toward zero [ref client.c:117-124, :157-160]: a failed
    lookup at longitude -0.3 makes the client answer "no data" for +0.3 as well, and a ray that
    leaves through the rim at longitude 0 is then located up to a degree too early.  Found in round 4
    by checking ALL of C5's 10 M rays against the reference (docs/lab_notebook_r4.md); the restatement
    and the kernels do not reproduce it (it depends on the order in which one thread's client met the
    rays).  Pinned here so that nobody mistakes it for a parity gap: every ray that differs under a
    locked stack is one that left the mosaic, and the unlocked reference sides with the restatement."""
    import philox_ref as P
    n, K, N = 6000, 48, 1201
    tiles = [(45, 0), (45, 1)]
    d = str(tmp_path / "tiles")
    for la, lo in tiles:
        synth.write_hgt(d, la, lo, N)
    geo = T.mosaic_oracle(tiles, N, 45, 0, 1, 2)
    lat, lon, _, _ = synth.uniform_rays(n, (45.0, 46.0), (0.0, 2.0), seed=5)
    pos, _ = geo.position(lat, lon, np.full(n, 20000.0))
    dirs = np.stack([P.isotropic(n, 7, k) for k in range(K)])
    ref_pos, total, taken = pos.copy(), np.zeros(n), np.zeros(n, dtype=np.int64)
    o = geo.step(ref_pos)
    alive = o["index"][:, 0] >= 0
    for k in range(K):
        o = geo.step(ref_pos, dirs[k])
        ref_pos = np.where(alive[:, None], o["position"], ref_pos)
        total += np.where(alive, o["step"], 0.0)
        taken += alive
        alive &= o["index"][:, 0] >= 0
    medium = np.where(alive, o["index"][:, 0], -1)
    assert 500 < (taken < K).sum() < n                 # rays did leave, not all of them
    free = R.stack_run(d, pos, dirs, walk_steps=K, local_range=0.0, locked=False)
    assert np.array_equal(free["index"][:, 0], medium) and np.array_equal(free["n_steps"], taken)
    assert np.array_equal(free["length"], total)        # bit for bit, every ray
    locked = R.stack_run(d, pos, dirs, walk_steps=K, local_range=0.0, threads=1, locked=True)
    differs = locked["length"] != total
    assert 0 < differs.sum() < n                        # the quirk is there ...
    # ... and only ever takes steps away: a ray is located leaving the mosaic too early, or -- the
    # memo outlives the ray that set it -- the NEXT ray of that client starts "outside" and takes none
    assert (locked["n_steps"] <= taken).all()
    assert (locked["n_steps"][differs] < taken[differs]).any() and (locked["n_steps"][differs] == 0).any()
