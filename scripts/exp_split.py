"""One batch split over streams: the same C2 rays traced as ONE call, and as P parts on P steppers
and streams issued together and waited for together (no overlap between consecutive batches): the
latency of a batch, not the throughput of a sequence of them.
usage: python3 scripts/exp_split.py"""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turtle_amd as TA
from turtle_amd import sharding, synth

n = int(os.environ.get("RAYS", "1000000"))
K = int(os.environ.get("K", "10"))
tmp = tempfile.mkdtemp(prefix="turtle_split_")
synth.write_hgt(tmp, 45, 3)
terrain = TA.Map.load(os.path.join(tmp, synth.hgt_name(45, 3)))
dev = torch.device("cuda", 0)
lat, lon, az, el = sharding.rank_rays(n, 0, (45., 46.), (3., 4.))
t = [torch.as_tensor(v, device=dev) for v in (lat, lon, az, el)]
P_MAX = 6
streams = [torch.cuda.Stream() for _ in range(P_MAX)]
first = TA.Stepper(); first.add_map(terrain, 0.0)
steppers = [first] + [first.clone() for _ in range(P_MAX - 1)]
torch.cuda.set_stream(streams[0]); TA.set_stream(streams[0])
pos0, _ = first.position(t[0], t[1], 500.0)
d = TA.ecef_from_horizontal(*t)
torch.cuda.synchronize()

def run(parts, hint):
    bounds = [n * i // parts for i in range(parts + 1)]
    p0 = [pos0[bounds[i]:bounds[i + 1]].contiguous() for i in range(parts)]
    dd = [d[bounds[i]:bounds[i + 1]].contiguous() for i in range(parts)]
    bufs = [x.clone() for x in p0]
    outs = [None] * parts
    TA.set_in_flight(hint)
    ts = []
    for k in range(K + 2):
        for i in range(parts):
            bufs[i].copy_(p0[i])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(parts):
            torch.cuda.set_stream(streams[i]); TA.set_stream(streams[i])
            outs[i] = steppers[i].trace(bufs[i], dd[i])
        torch.cuda.synchronize()
        if k >= 2: ts.append(time.perf_counter() - t0)
    TA.set_in_flight(1)
    torch.cuda.set_stream(streams[0]); TA.set_stream(streams[0])
    steps = sum(int(o["n_steps"].sum().item()) for o in outs)
    length = torch.cat([o["length"] for o in outs])
    return 1e3 * float(np.median(ts)), 1e3 * min(ts), steps, length

base = None
for parts, hint in ((1, 1), (2, 1), (2, 2), (3, 1), (3, 3), (4, 4), (6, 6), (1, 1)):
    med, best, steps, length = run(parts, hint)
    if base is None: base = length.clone()
    print(f"{parts} part(s), hint {hint}: median {med:.3f} ms, best {best:.3f} ms a batch, {steps / (med * 1e-3):.4g} ray-steps/s, "
          f"same path lengths: {torch.equal(length, base)}", flush=True)
