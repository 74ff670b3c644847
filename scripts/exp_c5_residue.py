#!/usr/bin/env python3
"""The handful of C5's 10 M walks that differ from the reference once its client is out of the
comparison (scripts/full_parity.py): which rays, where they end, in both arithmetics."""
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
terrain = bench.Terrain(TA, tiles, use_stack, env, 0, fmt="tif")
lat, lon, az, el = sharding.rank_rays(n, 0, terrain.lat_range, terrain.lon_range)
t = [torch.as_tensor(v, device=dev) for v in (lat, lon, az, el)]
pos0, di = terrain.stepper.position(t[0], t[1], 500.0)
K, SEED = 256, bench.SEED
walks = {}
for mode in ("fast", "strict"):
    TA.set_math(mode)
    w = terrain.stepper.scatter(pos0.clone(), SEED, K)
    torch.cuda.synchronize()
    walks[mode] = {k: w[k].cpu().numpy() for k in ("length", "index", "steps", "position")}
TA.set_math("fast")
path = terrain.hgt_files()
chunk, found = 1_000_000, []
for lo in range(0, n, chunk):
    hi = min(n, lo + chunk)
    dirs = np.stack([TA.isotropic(hi - lo, SEED, k, first_ray=lo, device=False) for k in range(K)])
    p = pos0[lo:hi].cpu().numpy()
    a = R.stack_run(path, p, dirs, walk_steps=K, local_range=0.0, threads=bench.host_cores())
    g = walks["fast"]
    rel = np.abs(g["length"][lo:hi] - a["length"]) / np.maximum(a["length"], 1e-300)
    sel = np.flatnonzero((g["index"][lo:hi, 0] != a["index"][:, 0]) | (rel > 1e-6))
    if sel.size:
        b = R.stack_run(path, p[sel], np.ascontiguousarray(dirs[:, sel]), walk_steps=K, local_range=0.0, locked=False)
        for j, r in enumerate(sel):
            for mode in ("fast", "strict"):
                g = walks[mode]
                rr = lo + r
                rel2 = abs(g["length"][rr] - b["length"][j]) / max(b["length"][j], 1e-300)
                if (g["index"][rr, 0] != b["index"][j, 0]) or (rel2 > 1e-6):
                    geo = TA.ecef_to_geodetic(g["position"][rr:rr + 1])
                    found.append((mode, rr))
                    print(f"[{mode}] ray {rr}: GPU L {g['length'][rr]:.6f} steps {g['steps'][rr]} medium {g['index'][rr, 0]} ends at "
                          f"lat {geo[0][0]:.9f} lon {geo[1][0]:.9f} alt {geo[2][0]:.3f} | reference (no client) L {b['length'][j]:.6f} "
                          f"steps {b['n_steps'][j]} medium {b['index'][j, 0]}  rel {rel2:.2e}", flush=True)
    print(f"  rays {lo}..{hi} done", flush=True)
print("found:", found)
terrain.close()
