"""Experiment: deviation of the fast trace from the reference's C1 golden traces."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import turtle_amd as TA
import amd_build as B
g = np.load(os.path.join("tests", "golden", "c1_traces.npz"))
for mode in ("strict", "fast"):
    TA.set_math(mode)
    m = B.c1_map()
    st = B.c1_stepper(m)
    t = st.trace(g["position"].copy(), g["direction"])
    dev = np.abs(t["position"] - g["r0_position"]).max(axis=1)
    dl = np.abs(t["length"] - g["r0_length"])
    o = np.argsort(-dev)[:4]
    print(mode, "range", os.environ.get("TURTLE_AMD_LINE_RANGE"), "max pos dev %.3e" % dev.max(), "max dL %.3e" % dl.max(),
          "worst rays", o, "dev", dev[o], "steps", t["n_steps"][o], "ref steps", g["r0_n_steps"][o], "median dev %.2e" % np.median(dev))
