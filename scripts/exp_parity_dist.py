"""Experiment: distribution of the GPU-vs-oracle path-length differences on the C2 batch."""
import os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import turtle_amd as TA
from turtle_amd import sharding
import amd_build as B, terrains as T
import tempfile, pathlib
N = int(os.environ.get("RAYS", "1000000"))
tile = B.hgt_tile(pathlib.Path(tempfile.mkdtemp()))
st = TA.Stepper(); st.add_map(tile, 0.0)
lat, lon, az, el = sharding.rank_rays(N, 0, (45.0, 46.0), (3.0, 4.0))
pos, di = st.position(lat, lon, 500.0)
d = TA.ecef_from_horizontal(lat, lon, az, el)
nodes, geo = T.hgt_oracle()
ref = geo.trace(pos, d, threads=16)
for mode in ("fast", "strict"):
    TA.set_math(mode)
    t = st.trace(pos.copy(), d)
    same = t["index"][:, 0] == ref["index"][:, 0]
    rel = np.abs(t["length"] - ref["length"]) / np.maximum(ref["length"], 1e-300)
    print(mode, "different medium:", int((~same).sum()), " rel>1e-9:", int((rel[same] > 1e-9).sum()), ">1e-8:", int((rel[same] > 1e-8).sum()),
          ">1e-7:", int((rel[same] > 1e-7).sum()), ">1e-6:", int((rel[same] > 1e-6).sum()), "median %.1e" % np.median(rel[same]))
    o = np.argsort(-np.where(same, rel, 0))[:5]
    for i in o:
        print("   ray", i, "rel %.2e" % rel[i], "L gpu %.6f ref %.6f" % (t["length"][i], ref["length"][i]), "steps", t["n_steps"][i], ref["n_steps"][i], "medium", t["index"][i, 0])
