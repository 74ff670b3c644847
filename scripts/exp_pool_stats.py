#!/usr/bin/env python3
"""What the waves of a pooled lined pass do (diagnostic build: scripts/build_variant.sh stats
-DTRACE_POOL_STATS, then TURTLE_AMD_LIBRARY=variants/stats.so).  usage: exp_pool_stats.py [c2|c4|c3] [rays]"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turtle_amd as TA                      # noqa: E402
from turtle_amd import binding, sharding, synth   # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else {"c2": 1_000_000, "c4": 12_500_000, "c3": 10_000_000}[wl]
tmp = tempfile.mkdtemp()
st = TA.Stepper()
if wl == "c3":
    for la in range(45, 49):
        for lo in range(3, 7):
            synth.write_hgt(tmp, la, lo)
    terrain = TA.Stack(tmp, 0)
    terrain.load()
    st.add_stack(terrain, 0.0)
    box = ((45.0, 49.0), (3.0, 7.0))
else:
    terrain = TA.Map.load(synth.write_hgt(tmp, 45, 3))
    st.add_map(terrain, 0.0)
    box = ((45.0, 46.0), (3.0, 4.0))
lat, lon, az, el = sharding.rank_rays(n, 0, *box)
pos, _ = st.position(lat, lon, 500.0)
d = TA.ecef_from_horizontal(lat, lon, az, el)
c = np.zeros(32, dtype=np.uint64)
st.trace(pos.copy(), d)
binding.lib().tamd_dev_pool_stats(c.ctypes.data_as(C.c_void_p), 1)
st.trace(pos.copy(), d)
binding.lib().tamd_dev_pool_stats(c.ctypes.data_as(C.c_void_p), 0)
c = c.astype(float)
z = lambda a, b: a / max(1.0, b)
print(f"{wl}, {n} rays, TURTLE_AMD_POOL={os.environ.get('TURTLE_AMD_POOL', '1')}")
print(f"looks at the pool {c[0]:.0f}: lean {c[1]:.0f}, service {c[2]:.0f}, as is {c[3]:.0f}; "
      f"{z(c[4], c[0]):.0f} ticks each, of which {z(c[16], c[0]):.0f} waiting for the lock; rays out {c[13]:.0f}, in {c[14]:.0f}")
print(f"  a wave then holds {z(c[17], c[0]):.1f} ready + {z(c[18], c[0]):.1f} waiting rays; the pool {z(c[19], c[0]):.1f} ready + {z(c[20], c[0]):.1f} waiting")
print(f"lean groups {c[5]:.0f}: {z(c[7], c[5]):.1f} lanes going at the start, {z(c[6], c[5]):.1f} lane-steps each "
      f"(of {32 * 64}); {c[6]:.0f} lean steps in all; {z(c[11], c[5]):.0f} ticks a group")
print(f"general iterations {c[8]:.0f}: {z(c[9], c[8]):.1f} lanes with a ray, {z(c[10], c[8]):.1f} closed forms each; {z(c[12], c[8]):.0f} ticks each")
print(f"ticks (sum over waves): pool {c[4]:.3g}, lean {c[11]:.3g}, general {c[12]:.3g}")
