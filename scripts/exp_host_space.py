"""Experiment: the C2 batch through the HOST-space API (numpy arrays in, numpy arrays
out: H2D + kernels + D2H inside the call) beside the DEVICE-space call."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turtle_amd as TA
from turtle_amd import sharding, synth

n = int(os.environ.get("RAYS", "1000000"))
tmp = tempfile.mkdtemp(prefix="turtle_host_")
synth.write_hgt(tmp, 45, 3)
terrain = TA.Map.load(os.path.join(tmp, synth.hgt_name(45, 3)))
st = TA.Stepper(); st.add_map(terrain, 0.0)
lat, lon, az, el = sharding.rank_rays(n, 0, (45., 46.), (3., 4.))
pos0, _ = st.position(lat, lon, 500.0)          # numpy in -> numpy out
d = TA.ecef_from_horizontal(lat, lon, az, el)
for rep in range(4):
    p = pos0.copy()
    t0 = time.perf_counter()
    out = st.trace(p, d)
    dt = time.perf_counter() - t0
    steps = int(out["n_steps"].sum())
    print(f"HOST space, {n} rays: {1e3 * dt:7.2f} ms per call, {steps / dt:.3g} ray-steps/s (92 B per ray over PCIe)")
dev = torch.device("cuda", 0)
tp, td = torch.as_tensor(pos0, device=dev), torch.as_tensor(d, device=dev)
for rep in range(3):
    p = tp.clone(); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = st.trace(p, td); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"DEVICE space: {1e3 * dt:7.2f} ms per call")
