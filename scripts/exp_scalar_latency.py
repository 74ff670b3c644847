#!/usr/bin/env python3
"""Latency of the scalar drop-in calls (each is a kernel launch over one ray)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import turtle_amd as TA          # noqa: E402
import amd_build as B            # noqa: E402
import terrains as T             # noqa: E402

m = B.c1_map()
st = B.c1_stepper(m)
lat, lon, az, el = TA.synth.uniform_rays(4, T.C1_Y, T.C1_X, seed=3)
pos, _ = st.position(lat, lon, 300.0)
d = TA.ecef_from_horizontal(lat, lon, az, el)
p = pos[0].copy()
for _ in range(200):
    st.step_scalar(p, d[0])
n = 3000
t0 = time.perf_counter()
q = p.copy()
for _ in range(n):
    q = st.step_scalar(q, d[0])["position"]
t1 = time.perf_counter()
for _ in range(n):
    m.elevation_scalar(3.5, 45.5)
t2 = time.perf_counter()
for _ in range(n):
    TA.binding.scalar_ecef_to_geodetic(p)
t3 = time.perf_counter()
print(f"turtle_stepper_step {1e6 * (t1 - t0) / n:.1f} us, turtle_map_elevation {1e6 * (t2 - t1) / n:.1f} us, "
      f"turtle_ecef_to_geodetic {1e6 * (t3 - t2) / n:.1f} us per call (ctypes overhead included)")
