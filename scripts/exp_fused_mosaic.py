#!/usr/bin/env python3
"""What a seamless mosaic is worth as ONE grid: C3's 4 x 4 tiles (10 M rays) through the stack they
come as, and through one 14 401 x 14 401 GeoTIFF of the same nodes loaded as a map (the upper bound of
fusing a resident, regular, seamless stack into one grid at upload)."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turtle_amd as TA
from turtle_amd import sharding, synth

n = int(os.environ.get("RAYS", "10000000"))
tmp = tempfile.mkdtemp(prefix="turtle_fused_")
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); TA.set_stream(stream)
dev = torch.device("cuda", 0)
N = synth.HGT_N - 1
for la in range(45, 49):
    for lo in range(3, 7):
        synth.write_hgt(os.path.join(tmp, "tiles"), la, lo)
# the same nodes as one grid (srtm_like_nodes counts its nodes globally: the seams agree)
j = (3 * N + np.arange(4 * N + 1, dtype=np.int64)).astype(np.float64)
i = (45 * N + np.arange(4 * N + 1, dtype=np.int64)).astype(np.float64)
z = np.rint(500.0 + 400.0 * np.sin(0.01 * j)[None, :] * np.cos(0.013 * i)[:, None]).astype(np.int16)
os.makedirs(os.path.join(tmp, "one"))
with open(os.path.join(tmp, "one", "mosaic.tif"), "wb") as f:
    f.write(synth.geotiff_bytes(z, 3.0, 49.0, 1.0 / N, 1.0 / N))
del z
lat, lon, az, el = sharding.rank_rays(n, 0, (45.0, 49.0), (3.0, 7.0))
t = [torch.as_tensor(v, device=dev) for v in (lat, lon, az, el)]
d = TA.ecef_from_horizontal(*t)
results = {}
for name in ("stack", "one grid"):
    st = TA.Stepper()
    if name == "stack":
        terrain = TA.Stack(os.path.join(tmp, "tiles"), 0); terrain.load(); st.add_stack(terrain, 0.0)
    else:
        terrain = TA.Map.load(os.path.join(tmp, "one", "mosaic.tif")); st.add_map(terrain, 0.0)
    pos0, di = st.position(t[0], t[1], 500.0)
    ts = []
    for _ in range(4):
        p = pos0.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); out = st.trace(p, d); b.record(stream); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    results[name] = out
    print(f"{name:9s} {min(ts):7.3f} ms (median {np.median(ts):.3f}); steps {int(out['n_steps'].sum())}, hits {int((out['index'][:, 0] == 0).sum())}", flush=True)
    st.destroy(); terrain.destroy()
a, b = results["stack"], results["one grid"]
same = (a["index"][:, 0] == b["index"][:, 0])
rel = (a["length"] - b["length"]).abs() / a["length"].clamp_min(1e-300)
print(f"media equal for {int(same.sum())} of {n}; worst |dL|/L among them {float(rel[same].max()):.2e}; "
      f"step counts differ on {int((a['n_steps'] != b['n_steps']).sum())}")
