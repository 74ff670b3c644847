#!/usr/bin/env python3
"""What the waves of the lined pass spend their time on (diagnostic build of the
library with -DTAMD_PROFILE: scripts/exp_lined_profile.sh).  C2, 1 M rays."""
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
rc = binding.lib().tamd_dev_prof_read(buf.ctypes.data_as(C.c_void_p))
assert rc == 0
tot = buf[:, 0].astype(float)
busy = np.flatnonzero((buf[:, 3] >> np.uint64(32)) > 0)
print(f"{busy.size} waves took samples; clock ticks are s_memtime (100 MHz on gfx950? see ratio below)")
order = busy[np.argsort(-tot[busy])]
print("wave   total   creep    slow  | iters groups entries slows fetch_iters  lanes/iter  last_count")
for w in list(order[:12]) + list(order[len(order) // 2: len(order) // 2 + 3]):
    it, gr = int(buf[w, 3] >> np.uint64(32)), int(buf[w, 3] & np.uint64(0xffffffff))
    sl, fe = int(buf[w, 4] >> np.uint64(32)), int(buf[w, 4] & np.uint64(0xffffffff))
    print(f"{w:5d} {tot[w]:8.0f} {float(buf[w, 1]):8.0f} {float(buf[w, 2]):8.0f} | {it:6d} {gr:6d} {int(buf[w, 7]):6d} "
          f"{sl:5d} {fe:6d}  {float(buf[w, 5]) / max(it, 1):8.1f}  {int(buf[w, 6])}")
w = order[0]
it, gr = int(buf[w, 3] >> np.uint64(32)), int(buf[w, 3] & np.uint64(0xffffffff))
gen = tot[w] - float(buf[w, 1])
print(f"critical wave: general part {gen:.0f} ticks over {it} iterations = {gen / max(it, 1):.1f} ticks each "
      f"(of which slow passes {float(buf[w, 2]):.0f}); creep {float(buf[w, 1]):.0f} ticks over {gr} groups of 4 = "
      f"{float(buf[w, 1]) / max(gr, 1) / 4:.1f} ticks per step")
