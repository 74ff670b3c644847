"""Experiment: rounds of a paged trace (TAMD_DEBUG_PAGING=1 prints them)."""
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turtle_amd as TA
from turtle_amd import synth
d = os.path.join(tempfile.mkdtemp(), "grid")
tiles = [(la, lo) for la in range(40, 45) for lo in range(5, 10) if (la, lo) != (42, 7)]
for la, lo in tiles:
    synth.write_hgt(d, la, lo, 1201)
TA.set_math(os.environ.get("MATH", "strict"))
paged = TA.Stack(d, 16)
sp = TA.Stepper(); sp.add_stack(paged, 0.0)
rng = np.random.default_rng(3)
n = int(os.environ.get("RAYS", "6000"))
lat, lon = rng.uniform(40.1, 44.9, n), rng.uniform(5.1, 9.9, n)
az, el = rng.uniform(0, 360, n), rng.uniform(-12.0, 2.0, n)
p1, d1 = sp.position(lat, lon, 400.0)
print("position done, resident", paged.resident)
keep = d1 == 0
dd = TA.ecef_from_horizontal(lat, lon, az, el)[keep]
t1 = sp.trace(p1[keep].copy(), dd)
print("trace done, resident", paged.resident, "steps", int(t1["n_steps"].sum()))
r1 = sp.trace(t1["position"].copy(), dd, resume_index=t1["index"])
print("resumed trace done, resident", paged.resident, "steps", int(r1["n_steps"].sum()))
