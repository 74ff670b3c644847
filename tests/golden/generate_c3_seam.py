#!/usr/bin/env python3
"""G11 c3_seam.npz: rays across the seams of a 2x2 mosaic of FULL-SIZE tiles (3601^2 nodes, the
tile of BASELINE's C3), traced by the REAL reference through a turtle_stack.  Build container only:

    make -C oracle ref && python tests/golden/generate_c3_seam.py

The four tiles are regenerated from turtle_amd/synth.py at test time (SHA-256 of the south-west one
stored); the fixture holds the ray origins and directions and the reference's index, path length
and step count -- 1500 rays that start within 0.01 degree of a seam, heading any way at -6 .. -0.5
degrees of elevation (they cross it, run along it, or leave through the mosaic's rim), and 500
anywhere on the mosaic."""
from __future__ import annotations

import hashlib
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import ref_ffi as R  # noqa: E402
from turtle_amd import synth  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
TILES = [(45, 3), (45, 4), (46, 3), (46, 4)]


def main():
    if not R.available():
        sys.exit("oracle/_ref/libturtle_ref.so missing: run `make -C oracle ref`")
    tmp = tempfile.mkdtemp(prefix="turtle_c3seam_")
    try:
        for la, lo in TILES:
            synth.write_hgt(tmp, la, lo)
        stack = R.RefStack(tmp, 0)
        stack.load()
        rng = np.random.Generator(np.random.Philox(1111))
        n_seam, n_any = 1500, 500
        # near the meridian seam (lon = 4), the parallel seam (lat = 46), and the crossing point
        lat = np.concatenate([rng.uniform(45.05, 46.95, n_seam // 2), 46.0 + rng.uniform(-0.01, 0.01, n_seam // 2),
                              rng.uniform(45.05, 46.95, n_any)])
        lon = np.concatenate([4.0 + rng.uniform(-0.01, 0.01, n_seam // 2), rng.uniform(3.05, 4.95, n_seam // 2),
                              rng.uniform(3.05, 4.95, n_any)])
        lat[:16] = 46.0 + rng.uniform(-0.002, 0.002, 16)   # all four tiles within reach
        n = lat.size
        az = rng.uniform(0.0, 360.0, n)
        el = rng.uniform(-6.0, -0.5, n)
        st = R.RefStepper()
        st.add_stack(stack, 0.0)
        st.range_set(0.0)
        pos = np.empty((n, 3))
        for k in range(n):
            rc, p, di = st.position(lat[k], lon[k], 300.0, 0)
            assert rc == 0 and di >= 0
            pos[k] = p
        direction = R.ecef_from_horizontal(lat, lon, az, el)
        t = st.trace(pos, direction)
        st.destroy()
        stack.destroy()
        sw = synth.srtm_like_nodes(45, 3)
        path = os.path.join(OUT, "c3_seam.npz")
        np.savez_compressed(path, tiles=np.array(TILES), lat=lat, lon=lon, position=pos, direction=direction,
                            nodes_sha=np.array(hashlib.sha256(np.ascontiguousarray(sw).tobytes()).hexdigest()),
                            **{f"t_{k}": v for k, v in t.items()})
        print(f"c3_seam.npz: {os.path.getsize(path) / 1024:.1f} KiB; media", np.unique(t["index"][:, 0], return_counts=True),
              "steps max", int(t["n_steps"].max()), "mean", float(t["n_steps"].mean()))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
