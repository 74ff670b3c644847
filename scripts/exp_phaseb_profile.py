#!/usr/bin/env python3
"""Where the waves of phase B spend their cycles (diagnostic build of the library:
scratch/prof2, s_memtime stamps around the sections of k_trace's loop).  C2, 1 M rays."""
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
for _ in range(3):
    t = st.trace(pos.copy(), d)
buf = np.zeros((4096, 8), dtype=np.uint64)
assert binding.lib().tamd_dev_prof_read(buf.ctypes.data_as(C.c_void_p)) == 0
f = buf.astype(float)
busy = np.flatnonzero(buf[:, 6] > 0)
order = busy[np.argsort(-f[busy, 0])]
print(f"{busy.size} waves of phase B took samples (cycles of s_memtime)")
print(" wave      total   refill    creep+   sample     book     tail  | iters  relays   per iter: sample book tail refill")
for w in list(order[:10]) + list(order[len(order) // 2: len(order) // 2 + 3]):
    it = max(1.0, f[w, 6])
    print(f"{w:5d} {f[w,0]:10.0f} {f[w,1]:8.0f} {f[w,2]:9.0f} {f[w,3]:8.0f} {f[w,4]:8.0f} {f[w,5]:8.0f}  | {int(it):5d} "
          f"{int(buf[w,7] & np.uint64(0xffffffff)):6d}   {f[w,3]/it:8.0f} {f[w,4]/it:5.0f} {f[w,5]/it:5.0f} {f[w,1]/it:5.0f}")
sec = np.zeros((4096, 16), dtype=np.uint64)
binding.lib().tamd_dev_sec_read(sec.ctypes.data_as(C.c_void_p))
sf = sec.astype(float)
hist = np.zeros((4096, 8, 16), dtype=np.uint32)
binding.lib().tamd_dev_hist_read(hist.ctypes.data_as(C.c_void_p))
names = ["prep", "line eval", "classify", "serves", "closed form", "rest", "book", "tail+refill+creep"]
print("sections of the general iteration: total cycles, entries, histogram (count at 2^b cycles)")
for w in order[:4]:
    print(f"  wave slot {w}:")
    for k in range(8):
        print(f"    {names[k]:18s} {sf[w, k]:9.0f} cycles in {int(sec[w, 8 + k]):5d} = {sf[w, k] / max(1.0, sf[w, 8 + k]):5.0f}: "
              + " ".join(f"2^{b + 8}:{int(hist[w, k, b])}" for b in range(16) if hist[w, k, b]))
