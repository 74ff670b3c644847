#!/usr/bin/env python3
"""Diagnosis: C5's first 500 000 rays, GPU walk against the reference's walk and the restatement's."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import turtle_amd as TA
from turtle_amd import sharding
from oracle import ref_ffi as R
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=0); torch.cuda.set_stream(stream); TA.set_stream(stream)
env = {"world": 1, "rank": 0, "dev": dev, "backend": "none"}
tiles, use_stack, n, text = bench.WORKLOADS["c5"]
n = int(os.environ.get("RAYS", "500000"))
terrain = bench.Terrain(TA, tiles, use_stack, env, 0, fmt="tif")
lat, lon, az, el = sharding.rank_rays(n, 0, terrain.lat_range, terrain.lon_range)
t = [torch.as_tensor(v, device=dev) for v in (lat, lon, az, el)]
pos0, di = terrain.stepper.position(t[0], t[1], 500.0)
K, SEED = 256, bench.SEED
for mode in ("fast", "strict"):
    TA.set_math(mode)
    w = terrain.stepper.scatter(pos0.clone(), SEED, K)
    torch.cuda.synchronize()
    if mode == "fast":
        dirs = np.stack([TA.isotropic(n, SEED, k, device=False) for k in range(K)])
        a = R.stack_run(terrain.hgt_files(), pos0.cpu().numpy(), dirs, walk_steps=K, local_range=0.0, threads=bench.host_cores())
    gl, gi, gs = (w[k].cpu().numpy() for k in ("length", "index", "steps"))
    rel = np.abs(gl - a["length"]) / np.maximum(a["length"], 1e-300)
    bad = np.flatnonzero(rel > 1e-6)
    print(f"[{mode}] beyond 1e-6: {bad.size}; media differ: {(gi[:,0] != a['index'][:,0]).sum()}; steps differ: {(gs != a['n_steps']).sum()}")
    for r in bad[:12]:
        print(f"   ray {r}: GPU L {gl[r]:.9f} steps {gs[r]} medium {gi[r,0]} | ref L {a['length'][r]:.9f} steps {a['n_steps'][r]} medium {a['index'][r,0]}  rel {rel[r]:.2e}")
TA.set_math("fast")
# the restatement on the rays that differ
geo = terrain.oracle()
sel = bad[:2000]
ref_pos = pos0.cpu().numpy()[sel].copy(); total = np.zeros(sel.size)
o = geo.step(ref_pos); alive = o["index"][:, 0] >= 0
for k in range(K):
    o = geo.step(ref_pos, dirs[k][sel])
    ref_pos = np.where(alive[:, None], o["position"], ref_pos)
    total += np.where(alive, o["step"], 0.0)
    alive &= o["index"][:, 0] >= 0
print("restatement vs reference on those rays: equal lengths", int((total == a["length"][sel]).sum()), "of", sel.size,
      "; restatement vs GPU(strict) equal:", int((np.abs(total - gl[sel]) <= 1e-6 * total).sum()))
terrain.close()
