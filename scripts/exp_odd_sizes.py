#!/usr/bin/env python3
"""Odd batch sizes through every optional path of the trace forced ON (the ray pool, the ordered
hand-over, the rays in spatial order) against the same batch with all of them OFF: bit for bit.
Run as two child processes per size (the knobs are read once)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %r)
import turtle_amd as TA
from turtle_amd import synth
n, out, work = int(sys.argv[1]), sys.argv[2], sys.argv[3]
res = {}
for tag in ("map", "stack"):
    st = TA.Stepper()
    if tag == "map":
        terrain = TA.Map.load(synth.write_hgt(os.path.join(work, "map"), 45, 3, 1201)); st.add_map(terrain, 0.0)
    else:
        for la, lo in ((45, 3), (45, 4), (46, 3), (46, 4)):
            synth.write_hgt(os.path.join(work, "stack"), la, lo, 1201)
        terrain = TA.Stack(os.path.join(work, "stack"), 0); st.add_stack(terrain, 0.0)
    box = ((45.0, 46.0), (3.0, 4.0)) if tag == "map" else ((45.0, 47.0), (3.0, 5.0))
    lat, lon, az, el = synth.uniform_rays(n, box[0], box[1], seed=n, el_range=(-6.0, -0.3))
    pos, _ = st.position(lat, lon, 400.0)
    d = TA.ecef_from_horizontal(lat, lon, az, el)
    t = st.trace(pos.copy(), d)
    for k in ("position", "index", "length", "n_steps"):
        res[tag + "_" + k] = np.asarray(t[k])
    st.destroy(); terrain.destroy()
np.savez(out, **res)
''' % ROOT
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "child.py"), "w").write(CHILD)
ok = True
sizes = [int(a) for a in sys.argv[1:]] or [1, 2, 63, 64, 65, 255, 256, 257, 1000, 4097, 100003, 700001]
for n in sizes:
    got = {}
    for tag, env in (("off", dict(TURTLE_AMD_POOL="0", TURTLE_AMD_SORT_KEY="0", TURTLE_AMD_SPATIAL="0")),
                     ("on", dict(TURTLE_AMD_POOL="1", TURTLE_AMD_SORT_KEY="1", TURTLE_AMD_SPATIAL="1"))):
        out = os.path.join(tmp, f"{n}_{tag}.npz")
        subprocess.run([sys.executable, os.path.join(tmp, "child.py"), str(n), out, os.path.join(tmp, "work")],
                       check=True, env=dict(os.environ, **env), timeout=300)
        got[tag] = dict(np.load(out))
    same = all(np.array_equal(got["on"][k], got["off"][k]) for k in got["off"])
    ok &= same
    print(f"n = {n:7d}: {'same bits' if same else 'DIFFERENT'}; steps {int(got['off']['map_n_steps'].sum())} / {int(got['off']['stack_n_steps'].sum())}", flush=True)
sys.exit(0 if ok else 1)
