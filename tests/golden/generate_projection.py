#!/usr/bin/env python3
"""G9 projection.npz: turtle_projection_project/unproject for every Lambert
variant and UTM (zone / extended / south), and stepper outputs + traces on a
UTM-projected map with and without a geoid, from the real reference (run in
the build container; see generate.py for the conventions)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_ffi as R  # noqa: E402
from turtle_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
NAMES = ["Lambert I", "Lambert II", "Lambert IIe", "Lambert III", "Lambert IV", "Lambert 93",
         "UTM 31N", "UTM 3.5N", "UTM 19S", "UTM -70.25S"]
# the projected test map of tests/test-turtle.c:67-95, without the PNG round trip
UTM_X, UTM_Y, UTM_Z = (495000.0, 497000.0), (5066000.0, 5068000.0), (0.0, 1000.0)


def utm_nodes():
    i = np.arange(201, dtype=np.float64)
    return 300.0 + 250.0 * np.sin(i / 17.0)[None, :] * np.cos(i / 23.0)[:, None] + i[None, :]


def main():
    rng = np.random.Generator(np.random.Philox(909))
    out = {}
    n = 512
    for k, name in enumerate(NAMES):
        pr = R.RefProjection(name)
        if name.startswith("Lambert"):
            lat = rng.uniform(41.0, 51.5, n)
            lon = rng.uniform(-5.5, 10.0, n)
        elif name.endswith("N"):
            lat = rng.uniform(0.0, 84.0, n)
            lon = rng.uniform(0.0, 6.0, n)
        else:
            lat = rng.uniform(-80.0, 0.0, n)
            lon = rng.uniform(-73.5, -67.0, n)
        x, y = pr.project(lat, lon)
        la2, lo2 = pr.unproject(x, y)
        out[f"p{k}_lat"], out[f"p{k}_lon"] = lat, lon
        out[f"p{k}_x"], out[f"p{k}_y"] = x, y
        out[f"p{k}_ulat"], out[f"p{k}_ulon"] = la2, lo2
        pr.destroy()

    nodes = utm_nodes()
    m = R.RefMap.create(nodes, UTM_X, UTM_Y, UTM_Z, "UTM 31N")
    gn = np.linspace(-30.0, 30.0, 361)[None, :] * np.ones((181, 1))
    geoid = R.RefMap.create(gn, (0.0, 360.0), (-90.0, 90.0), (-40.0, 40.0))
    centre = R.RefProjection("UTM 31N")
    clat, clon = centre.unproject([496000.0], [5067000.0])
    centre.destroy()
    for gname, g in (("nogeoid", None), ("geoid", geoid)):
        st = R.RefStepper()
        if g is not None:
            st.geoid_set(g)
        st.range_set(0.0)
        st.add_flat(-5.0)        # a flat sea below, then the projected map on top of it
        st.add_map(m, 0.0)
        la = clat[0] + rng.uniform(-0.007, 0.007, 400)
        lo = clon[0] + rng.uniform(-0.010, 0.010, 400)
        az = rng.uniform(0, 360, 400)
        el = rng.uniform(-20, -2, 400)
        pos = np.empty((400, 3))
        di = np.empty(400, dtype=np.int32)
        for k in range(400):
            rc, p, d = st.position(la[k], lo[k], 150.0, 0)
            assert rc == 0
            pos[k], di[k] = p, d
        direction = R.ecef_from_horizontal(la, lo, az, el)
        rows = []
        for k in range(0, 400, 4):
            o = st.step(pos[k], None)
            rows.append([o["latitude"], o["longitude"], o["altitude"], *o["elevation"], o["step"],
                         *o["index"]])
        t = st.trace(pos, direction)
        st.destroy()
        out.update({f"{gname}_lat": la, f"{gname}_lon": lo, f"{gname}_pos": pos,
                    f"{gname}_di": di, f"{gname}_dir": direction, f"{gname}_rows": np.array(rows),
                    **{f"{gname}_t_{k}": v for k, v in t.items()}})
    m.destroy()
    geoid.destroy()
    np.savez_compressed(os.path.join(OUT, "projection.npz"), names=np.array(NAMES),
                        nodes=nodes, geoid_nodes=gn, **out)
    print("projection.npz: data index counts", np.bincount(out["nogeoid_di"] + 1),
          "final media", np.bincount(out["nogeoid_t_index"][:, 0] + 1))
    print(R.errors())


if __name__ == "__main__":
    main()
