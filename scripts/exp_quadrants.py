"""Experiment: fast vs strict deviations from the oracle for grazing rays at several places."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import turtle_amd as TA
from oracle import ffi as O
import amd_build as B, terrains as T
places = {"north-east": ((3.0, 4.0), (45.0, 46.0)), "south-west": ((-71.0, -70.0), (-34.0, -33.0)),
          "north-80": ((15.0, 17.0), (79.5, 80.0)), "equator-dateline": ((178.9, 179.85), (-0.5, 0.5))}
for where, (x, y) in places.items():
    nodes = T.c1_nodes()
    geo = O.OracleGeometry(grids=[O.default_grid(nodes, x, y, T.C1_Z)], layers=[[(O.MAP, 0, 0.0)]])
    m = TA.Map.create(nodes, x, y, T.C1_Z)
    st = B.c1_stepper(m)
    rng = np.random.default_rng(5)
    n = 1500
    lat = rng.uniform(y[0] + 0.2 * (y[1] - y[0]), y[1] - 0.2 * (y[1] - y[0]), n)
    lon = rng.uniform(x[0] + 0.2 * (x[1] - x[0]), x[1] - 0.2 * (x[1] - x[0]), n)
    az = rng.uniform(0.0, 360.0, n); el = rng.uniform(-3.0, 1.0, n)
    pos0, di = geo.position(lat, lon, 30.0)
    d = O.ecef_from_horizontal(lat, lon, az, el)
    ref = geo.trace(pos0, d, threads=4)
    for mode in ("strict", "fast"):
        TA.set_math(mode)
        t = st.trace(pos0.copy(), d)
        same = t["index"][:, 0] == ref["index"][:, 0]
        rel = np.abs(t["length"] - ref["length"]) / np.maximum(ref["length"], 1e-300)
        w = int(np.argmax(np.where(same, rel, 0)))
        print(f"{where:18s} {mode:6s} long rays {int((ref['n_steps'] > 512).sum())}, flipped {int((~same).sum())}, rel>1e-9: {int((rel[same] > 1e-9).sum())} >1e-8: {int((rel[same] > 1e-8).sum())} >1e-7: {int((rel[same] > 1e-7).sum())} >1e-6: {int((rel[same] > 1e-6).sum())}"
              f" worst ray {w}: dL {t['length'][w] - ref['length'][w]:+.2e} m of {ref['length'][w]:.1f}, steps {t['n_steps'][w]} vs {ref['n_steps'][w]}")
    TA.set_math("fast")
