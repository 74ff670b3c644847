"""Experiment: |altitude - ground| at the end point of rays that hit, against their step count."""
import os, sys, tempfile, pathlib
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import turtle_amd as TA
from turtle_amd import sharding
import amd_build as B
N = 1000000
tile = B.hgt_tile(pathlib.Path(tempfile.mkdtemp()))
st = TA.Stepper(); st.add_map(tile, 0.0)
lat, lon, az, el = sharding.rank_rays(N, 0, (45.0, 46.0), (3.0, 4.0))
pos, di = st.position(lat, lon, 500.0)
d = TA.ecef_from_horizontal(lat, lon, az, el)
t = st.trace(pos.copy(), d)
hit = np.flatnonzero(t["index"][:, 0] == 0)
o = st.step(t["position"][hit].copy(), None)
ground = np.where(o["index"][:, 0] == 0, o["elevation"][:, 1], o["elevation"][:, 0])
dev = np.abs(o["altitude"] - ground)
n = t["n_steps"][hit]
for lo, hi in ((0, 512), (512, 1024), (1024, 4096), (4096, 100000)):
    sel = (n >= lo) & (n < hi)
    if sel.any():
        print(f"steps [{lo},{hi}): {int(sel.sum())} rays, max dev {dev[sel].max():.2e}, 99.9% {np.percentile(dev[sel], 99.9):.2e}, median {np.median(dev[sel]):.1e}")
w = np.argsort(-dev)[:5]
# straight-line check: position vs origin + d * length
err = np.abs(t["position"] - (pos + d * t["length"][:, None])).max(axis=1)
print("worst:", [(int(hit[i]), int(n[i]), float(dev[i]), float(err[hit[i]])) for i in w])
print("straight-line error: max %.2e" % err.max())
