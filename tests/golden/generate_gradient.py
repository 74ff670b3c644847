#!/usr/bin/env python3
"""G8 gradient.npz: turtle_map_gradient / turtle_stack_gradient of the real
reference (run in the build container; see generate.py for the conventions)."""
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
C1_X, C1_Y, C1_Z = (3.0, 4.0), (45.0, 46.0), (0.0, 2000.0)


def main():
    rng = np.random.Generator(np.random.Philox(808))
    nodes = synth.c1_gradient_nodes()
    nodes = 0.8 * nodes + 250.0 + 200.0 * np.sin(np.arange(256) / 9.0)[:, None]  # vary along y too
    m = R.RefMap.create(nodes, C1_X, C1_Y, C1_Z)
    n = 4096
    x = rng.uniform(2.98, 4.02, n)
    y = rng.uniform(44.98, 46.02, n)
    d = 1.0 / 255
    special = [(3.0, 45.0), (4.0, 46.0), (3.0 + 0.3 * d, 45.0 + 0.3 * d),  # first half-cells
               (3.5, 45.0 + 0.2 * d), (3.5, 45.0 + 0.7 * d), (3.0 + 0.2 * d, 45.5),
               (4.0 - 0.2 * d, 45.5), (3.5, 46.0 - 0.2 * d), (4.0 - 0.7 * d, 46.0 - 0.7 * d),
               (3.0 + 1.5 * d, 45.0 + 1.5 * d), (np.nan, 45.5), (3.5, 47.0)]
    for k, (a, b) in enumerate(special):
        x[k], y[k] = a, b
    gx, gy, inside = m.gradient(x, y)
    m.destroy()
    tmp = tempfile.mkdtemp(prefix="turtle_grad_")
    try:
        d2 = os.path.join(tmp, "mosaic")
        for la, lo in ((45, 3), (45, 4), (46, 3)):
            synth.write_hgt(d2, la, lo, 1201)
        stack = R.RefStack(d2, 0)
        stack.load()
        lat = rng.uniform(44.95, 47.05, 2048)
        lon = rng.uniform(2.95, 5.05, 2048)
        lat[:4] = [45.5, 46.0, 46.5, 45.0 + 0.3 / 1200]
        lon[:4] = [3.5, 4.0, 4.5, 3.5]
        glat, glon, sin = stack.gradient(lat, lon)
        stack.destroy()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    np.savez_compressed(os.path.join(OUT, "gradient.npz"), nodes=nodes, x=x, y=y, gx=gx, gy=gy,
                        inside=inside, lat=lat, lon=lon, glat=glat, glon=glon, sinside=sin)
    print("gradient.npz", int(inside.sum()), "inside;", int(sin.sum()), "stack inside")
    R.errors()


if __name__ == "__main__":
    main()
