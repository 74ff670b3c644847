"""What the ORDER of the rays in a batch is worth: the same C2 rays (1 M through one tile) traced in
their drawn order, sorted by elevation angle, by their step count (either way), and in waves of
like step counts shuffled among themselves.
usage: python3 scripts/exp_order.py"""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turtle_amd as TA
from turtle_amd import sharding, synth

n = int(os.environ.get("RAYS", "1000000"))
tmp = tempfile.mkdtemp(prefix="turtle_order_")
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

def timed(ids, reps=6):
    if ids is None:
        p0, dd = pos0, d
    else:
        ids_t = torch.as_tensor(np.ascontiguousarray(ids), device=dev)
        p0, dd = pos0[ids_t].contiguous(), d[ids_t].contiguous()
    ts = []
    for _ in range(reps):
        p = p0.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); st.trace(p, dd); b.record(stream); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return min(ts), float(np.median(ts))

rng = np.random.default_rng(1)
by_steps = np.argsort(-steps, kind="stable")
blocks = by_steps.reshape(-1, 64).copy()          # waves of like rays ...
rng.shuffle(blocks, axis=0)                       # ... in any order
orders = {
    "as drawn": None,
    "by elevation angle, steepest first": np.argsort(el, kind="stable"),
    "by elevation angle, shallowest first": np.argsort(-el, kind="stable"),
    "by step count, longest first": by_steps,
    "by step count, shortest first": by_steps[::-1],
    "like step counts together, blocks of 64 shuffled": blocks.reshape(-1),
}
for name, ids in orders.items():
    best, med = timed(ids)
    print(f"{name:52s} best {best:.3f} ms  median {med:.3f} ms", flush=True)
