import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import turtle_amd as TA
from turtle_amd import sharding, synth
n = 1_000_000
tmp = tempfile.mkdtemp()
tile = TA.Map.load(synth.write_hgt(tmp, 45, 3))
st = TA.Stepper(); st.add_map(tile, 0.0)
lat, lon, az, el = sharding.rank_rays(n, 0, (45.0, 46.0), (3.0, 4.0))
pos, _ = st.position(lat, lon, 500.0)
d = TA.ecef_from_horizontal(lat, lon, az, el)
t = st.trace(pos.copy(), d)
s = t["n_steps"]
print("rays", n, "steps", s.sum(), "max", s.max())
for T in (32, 128, 512, 1024, 2048, 3000, 4096, 6000, 8000, 10000):
    sel = s > T
    print(f"> {T:5d} steps: {sel.sum():7d} rays ({100.0 * sel.mean():.3f} %), their steps beyond it {int((s[sel] - T).sum()):10d}")

# How well does a ray's state after its first 32 steps predict that it will be a long one?
t32 = st.trace(pos.copy(), d, max_steps=32)
alive = t32["n_steps"] >= 32
o = st.step(t32["position"].copy(), None)
ground = np.where(o["index"][:, 0] == 0, o["elevation"][:, 1], o["elevation"][:, 0])
clear = np.abs(o["altitude"] - ground)
clear[~alive] = np.inf
order = np.argsort(clear)
for T in (512, 1024, 2048):
    longs = s > T
    for frac in (0.01, 0.02, 0.05, 0.1, 0.2, 0.3, 0.5):
        head = order[: int(frac * n)]
        print(f"rays beyond {T} steps: {100.0 * longs[head].sum() / max(1, longs.sum()):5.1f} % are among the {100 * frac:4.1f} % of rays "
              f"closest to the ground after 32 steps (clearance below {clear[head[-1]]:.2f} m)")

# ... and the rate at which it came down: clearance c0 = 500 m at the start, c32 after 32 steps,
# shrinking by (c32/c0)^(1/32) a step: steps left ~ 32 ln(c32 / resolution) / ln(c0 / c32)
c32 = np.where(alive, clear, 1.0)
with np.errstate(divide="ignore", invalid="ignore"):
    left = 32.0 * np.log(np.maximum(c32, 0.02) / 0.01) / np.log(500.0 / c32)
left[~alive] = 0.0
left[c32 >= 500.0] = 0.0           # going up
order = np.argsort(-left)
for T in (512, 1024, 2048):
    longs = s > T
    for frac in (0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.2):
        head = order[: int(frac * n)]
        print(f"rays beyond {T} steps: {100.0 * longs[head].sum() / max(1, longs.sum()):5.1f} % are among the {100 * frac:4.1f} % of rays "
              f"with most steps predicted (more than {left[head[-1]]:.0f})")
