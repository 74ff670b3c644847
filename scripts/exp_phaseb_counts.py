#!/usr/bin/env python3
"""What the waves of the lined pass do (diagnostic build: scripts/exp_phaseb_build.py, then run with
TURTLE_AMD_LIBRARY=scratch/prof3/libturtle_amd.so).  C2's terrain; rays on the command line."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import turtle_amd as TA                      # noqa: E402
from turtle_amd import binding, sharding, synth   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
tmp = tempfile.mkdtemp()
tile = TA.Map.load(synth.write_hgt(tmp, 45, 3))
st = TA.Stepper()
st.add_map(tile, 0.0)
lat, lon, az, el = sharding.rank_rays(n, 0, (45.0, 46.0), (3.0, 4.0))
pos, _ = st.position(lat, lon, 500.0)
d = TA.ecef_from_horizontal(lat, lon, az, el)
# SUBSET=lo:hi:k -- only the first k rays whose step count is in [lo, hi] (a wave of medium rays alone)
if os.environ.get("SUBSET"):
    lo, hi, k = (int(x) for x in os.environ["SUBSET"].split(":"))
    steps = st.trace(pos.copy(), d)["n_steps"]
    ids = np.nonzero((steps >= lo) & (steps <= hi))[0][:k]
    pos, d = np.ascontiguousarray(pos[ids]), np.ascontiguousarray(d[ids])
    print(f"subset: {ids.size} rays of {lo}..{hi} steps")
c = np.zeros(32, dtype=np.uint64)
st.trace(pos.copy(), d)
binding.lib().tamd_dev_cnt_read(c.ctypes.data_as(C.c_void_p), 1)
st.trace(pos.copy(), d)
binding.lib().tamd_dev_cnt_read(c.ctypes.data_as(C.c_void_p), 0)
c = c.astype(float)
print(f"general iterations (waves) {c[0]:.0f} ({c[11]:.0f} of them with more than creep_lanes live lanes), samples in them {c[1]:.0f} "
      f"= {c[1] / max(1, c[0]):.1f} lanes each; closed forms among them {c[8]:.0f}")
print(f"lean groups: sparse {c[2]:.0f} with {c[4]:.0f} steps ({c[4] / max(1, c[2]):.2f} per group); "
      f"busy {c[3]:.0f} with {c[5]:.0f} steps ({c[5] / max(1, c[3]):.2f} per group); busy entries {c[6]:.0f}, backed off {c[7]:.0f}")
print(f"cycles (sum over waves): lean loops {c[9]:.3g}, general iterations {c[10]:.3g} = {c[10] / max(1, c[0]):.0f} each")
print(f"first step of a lean group, live lanes {c[12]:.0f}: not stepping {c[13]:.0f}, no line {c[14]:.0f}, rim/range/cap {c[15]:.0f}, "
      f"another cell {c[16]:.0f}; of those that evaluate: line does not serve {c[17]:.0f}, another medium {c[18]:.0f}")
print(f"lanes of the general iterations: not yet on a line {c[20]:.0f}, bisecting {c[21]:.0f}, starting {c[22]:.0f}; "
      f"closed forms taken at once {c[19]:.0f}, lane-iterations spent waiting for one {c[23]:.0f}")
sp = np.zeros((4096, 4), dtype=np.uint64)
binding.lib().tamd_dev_span_read(sp.ctypes.data_as(C.c_void_p))
used = sp[:, 1] > 0
# (each XCD has its own counter: only differences within a wave mean anything; the waves of a
# persistent launch start together)
end = (sp[used, 1] - sp[used, 0]).astype(float)
dry = np.where(sp[used, 2] > 0, sp[used, 2].astype(float) - sp[used, 0].astype(float), np.nan)
unit = end.max() / 100.0
print(f"{used.sum()} waves; the longest lives {end.max():.3g} ticks of s_memtime; in hundredths of that:")
print("  waves find the queue dry at: " + " ".join(f"{np.nanpercentile(dry, q) / unit:.0f}" for q in (1, 25, 50, 75, 99)) + "  (percentiles 1 25 50 75 99)")
print("  waves end at:                " + " ".join(f"{np.percentile(end, q) / unit:.0f}" for q in (1, 10, 25, 50, 75, 90, 99)) + "  (percentiles 1 10 25 50 75 90 99)")
