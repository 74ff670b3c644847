#!/usr/bin/env python3
"""EVERY ray of the BASELINE configurations against the CPU, not a sample: C2 (1 M rays), C4 (12.5 M,
the ray pool on) and C3 (10 M through the 4 x 4 stack) against the CPU restatement on all host cores;
C5 (10 M rays x 256 scattering steps over the 10 x 10 GeoTIFF mosaic) against the REFERENCE itself
(oracle/_ref: one locked stack shared by the threads, a client each, exact transform), where that
build travelled with the repo.  Minutes of host time: run by hand (profiles/r04_full_parity.txt), the
test suite and bench.py check samples of the same."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402
import turtle_amd as TA                        # noqa: E402
from turtle_amd import sharding                # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=0)
torch.cuda.set_stream(stream)
TA.set_stream(stream)
env = {"world": 1, "rank": 0, "dev": dev, "backend": "none"}
cores = bench.host_cores()
which = sys.argv[1:] or ["c2", "c4", "c3", "c5"]


def rays(terrain, n):
    lat, lon, az, el = sharding.rank_rays(n, 0, terrain.lat_range, terrain.lon_range)
    t = [torch.as_tensor(v, device=dev) for v in (lat, lon, az, el)]
    pos, di = terrain.stepper.position(t[0], t[1], 500.0)
    assert int((di != 0).sum()) == 0
    return pos, TA.ecef_from_horizontal(*t)


for name in which:
    tiles, use_stack, n, text = bench.WORKLOADS[name]
    t0 = time.time()
    terrain = bench.Terrain(TA, tiles, use_stack, env, 0, fmt="tif" if name == "c5" else "hgt")
    pos0, d = rays(terrain, n)
    if name != "c5":
        out = terrain.stepper.trace(pos0.clone(), d)
        torch.cuda.synchronize()
        t1 = time.time()
        ref = terrain.oracle().trace(pos0.cpu().numpy(), d.cpu().numpy(), local_range=0.0, threads=cores)
        c = bench.parity_counts(out["index"].cpu().numpy(), out["length"].cpu().numpy(), ref["index"], ref["length"],
                                out["n_steps"].cpu().numpy(), ref["n_steps"])
        c.pop("checker")
        print(f"{name}: ALL {n} rays against the CPU restatement ({cores} threads, {time.time() - t1:.0f} s): {c}; "
              f"steps GPU {int(out['n_steps'].sum())} CPU {int(ref['n_steps'].sum())}", flush=True)
    else:
        from oracle import ref_ffi as R
        if not R.driver_available():
            print("c5: the reference's build (oracle/_ref) is not here: skipped")
            terrain.close()
            continue
        K, SEED = 256, bench.SEED
        w = terrain.stepper.scatter(pos0.clone(), SEED, K)
        torch.cuda.synchronize()
        path = terrain.hgt_files()
        chunk, bad_medium, beyond, bad_steps, worst, secs, total = 500_000, 0, 0, 0, 0.0, 0.0, 0
        # the reference's CLIENT (a locked stack) mislocates rays that leave through the rim at
        # longitude 0 [ref client.c:117-124: its memo truncates toward zero]: the rays that differ
        # from it go through the reference WITHOUT a client (an unlocked stack, one thread) as well
        again, again_medium, again_beyond, again_steps, again_worst = 0, 0, 0, 0, 0.0
        t1 = time.time()
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            dirs = np.stack([TA.isotropic(hi - lo, SEED, k, first_ray=lo, device=False) for k in range(K)])
            a = R.stack_run(path, pos0[lo:hi].cpu().numpy(), dirs, walk_steps=K, local_range=0.0, threads=cores)
            secs += a["seconds"]
            total += a["total_steps"]
            gi, gl, gs = (w[k][lo:hi].cpu().numpy() for k in ("index", "length", "steps"))
            flipped = gi[:, 0] != a["index"][:, 0]
            rel = np.abs(gl - a["length"]) / np.maximum(np.abs(a["length"]), 1e-300)
            bad_medium += int(flipped.sum())
            beyond += int((~flipped & (rel > 1e-6)).sum())
            bad_steps += int((gs != a["n_steps"]).sum())
            worst = max(worst, float(rel[~flipped].max(initial=0.0)))
            sel = np.flatnonzero(flipped | (rel > 1e-6) | (gs != a["n_steps"]))
            if sel.size:
                b = R.stack_run(path, pos0[lo:hi].cpu().numpy()[sel], np.ascontiguousarray(dirs[:, sel]),
                                walk_steps=K, local_range=0.0, locked=False)
                f2 = gi[sel, 0] != b["index"][:, 0]
                r2 = np.abs(gl[sel] - b["length"]) / np.maximum(np.abs(b["length"]), 1e-300)
                again += sel.size
                again_medium += int(f2.sum())
                again_beyond += int((~f2 & (r2 > 1e-6)).sum())
                again_steps += int((gs[sel] != b["n_steps"]).sum())
                again_worst = max(again_worst, float(r2[~f2].max(initial=0.0)))
            print(f"  c5 rays {lo}..{hi}: so far {bad_medium} with another medium, {beyond} beyond 1e-6, "
                  f"{bad_steps} with another step count than the reference's clients give; of those {again} rays "
                  f"through the reference without a client: {again_medium} / {again_beyond} / {again_steps} "
                  f"({time.time() - t1:.0f} s)", flush=True)
        print(f"c5: ALL {n} rays x {K} steps against the reference itself ({cores} threads, exact transform, "
              f"{secs:.0f} s of stepping, {total} steps): medium_mismatch {bad_medium}, beyond_1e-6 {beyond}, "
              f"step_count_mismatch {bad_steps}, max_rel_path_length {worst:.3g}; steps GPU {int(w['steps'].sum())}.\n"
              f"    Those {again} rays (all of them rays that left the mosaic) against the reference WITHOUT a client "
              f"(an unlocked stack, one thread): medium_mismatch {again_medium}, beyond_1e-6 {again_beyond}, "
              f"step_count_mismatch {again_steps}, max_rel_path_length {again_worst:.3g} -- the reference's client "
              f"memoises 'no data' per integer (latitude, longitude) by truncation toward zero [ref client.c:117-124], "
              f"so a ray leaving through the rim at longitude 0 is located up to a degree early (lab notebook r4).",
              flush=True)
    terrain.close()
    print(f"  ({name}: {time.time() - t0:.0f} s in all)", flush=True)
