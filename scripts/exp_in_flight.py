"""Two batches in flight: the same C2 traces, K of them, on one stepper and stream, and alternating
between two steppers on two streams (the tail of one batch beside the bulk of the next).
usage: python3 scripts/exp_in_flight.py"""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turtle_amd as TA
from turtle_amd import sharding, synth

n = int(os.environ.get("RAYS", "1000000"))
K = int(os.environ.get("K", "20"))
tmp = tempfile.mkdtemp(prefix="turtle_flight_")
synth.write_hgt(tmp, 45, 3)
terrain = TA.Map.load(os.path.join(tmp, synth.hgt_name(45, 3)))
dev = torch.device("cuda", 0)
lat, lon, az, el = sharding.rank_rays(n, 0, (45., 46.), (3., 4.))
t = [torch.as_tensor(v, device=dev) for v in (lat, lon, az, el)]
streams = [torch.cuda.Stream() for _ in range(3)]
steppers = []
for _ in range(3):
    st = TA.Stepper(); st.add_map(terrain, 0.0); steppers.append(st)
torch.cuda.set_stream(streams[0]); TA.set_stream(streams[0])
pos0, _ = steppers[0].position(t[0], t[1], 500.0)
d = TA.ecef_from_horizontal(*t)
torch.cuda.synchronize()
bufs = [pos0.clone() for _ in range(3)]

def run(width):
    outs = [None] * width
    for w in range(width):      # warm-up: scratch of each stepper
        torch.cuda.set_stream(streams[w]); TA.set_stream(streams[w])
        bufs[w].copy_(pos0); outs[w] = steppers[w].trace(bufs[w], d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        w = k % width
        torch.cuda.set_stream(streams[w]); TA.set_stream(streams[w])
        bufs[w].copy_(pos0)
        outs[w] = steppers[w].trace(bufs[w], d)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = int(outs[0]["n_steps"].sum().item())
    return dt / K * 1e3, steps * K / dt, outs[0]

base = None
for width in (1, 2, 3, 1, 2):
    ms, rate, out = run(width)
    if base is None: base = {k: out[k].clone() for k in ("index", "length", "n_steps")}
    same = all(torch.equal(out[k], base[k]) for k in base)
    print(f"{width} in flight: {ms:.3f} ms a trace, {rate:.4g} ray-steps/s, same results: {same}", flush=True)
