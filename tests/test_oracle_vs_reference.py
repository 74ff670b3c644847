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
