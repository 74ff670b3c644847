"""Experiment: how fast do the longest rays of the C2 batch advance when traced alone?

Traces the C2 batch once, picks the K longest rays, and times launches that hold
only those (1, 8, 64, 512 rays): the time of such a launch is the serial chain
of its longest ray, i.e. the floor of the tail of a full launch.
"""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turtle_amd as TA
from turtle_amd import sharding, synth

n = int(os.environ.get("RAYS", "1000000"))
tmp = tempfile.mkdtemp(prefix="turtle_longest_")
synth.write_hgt(tmp, 45, 3)
terrain = TA.Map.load(os.path.join(tmp, synth.hgt_name(45, 3)))
st = TA.Stepper(); st.add_map(terrain, 0.0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); TA.set_stream(stream)
lat, lon, az, el = sharding.rank_rays(n, 0, (45., 46.), (3., 4.))
dev = torch.device("cuda", 0)
t = [torch.as_tensor(v, device=dev) for v in (lat, lon, az, el)]
pos0, _ = st.position(t[0], t[1], 500.0)
d = TA.ecef_from_horizontal(*t)
out = st.trace(pos0.clone(), d)
steps = out["n_steps"].cpu().numpy()
length = out["length"].cpu().numpy()
order = np.argsort(-steps)
print("longest rays: steps", steps[order[:8]], "mean step m", (length[order[:8]] / steps[order[:8]]).round(3))
q = np.percentile(steps, [50, 90, 99, 99.9, 99.99])
print("steps percentiles 50/90/99/99.9/99.99:", q)

def timed(ids, reps=3):
    ids_t = torch.as_tensor(ids, device=dev)
    p0, dd = pos0[ids_t].contiguous(), d[ids_t].contiguous()
    best = 1e9
    for _ in range(reps):
        p = p0.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); o = st.trace(p, dd); b.record(stream); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best, int(o["n_steps"].max()), int(o["n_steps"].sum())

for k in (1, 8, 64, 512, 4096):
    ms, longest, total = timed(order[:k])
    print(f"top {k:5d} rays alone: {ms:7.3f} ms, longest {longest} steps -> {1e3 * ms / longest:.3f} us per step of the longest; total steps {total}")
# a ray in the middle of the distribution, alone
mid = order[len(order) // 2: len(order) // 2 + 1]
ms, longest, total = timed(mid)
print(f"a median ray alone: {ms:.3f} ms, {longest} steps -> {1e3 * ms / longest:.3f} us per step")
# rays of 1 500 - 2 500 steps (the ones phase B waits for), alone and a few to a wave
medium = order[(steps[order] <= 2500) & (steps[order] >= 1500)]
for k in (1, 8, 64, 1024):
    if medium.size >= k:
        ms, longest, total = timed(medium[:k])
        print(f"{k:5d} medium rays alone: {ms:7.3f} ms, longest {longest} steps -> {1e3 * ms / longest:.3f} us per step of the longest")
