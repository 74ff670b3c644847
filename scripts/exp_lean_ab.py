"""Quick A/B probe of the lined pass: the longest C2 ray alone (a lone wave's step), 64 of the
longest / 64 medium rays in one wave (a busy wave's step), and the whole 1 M-ray trace.
usage: TURTLE_AMD_LIBRARY=<build> python3 scripts/exp_lean_ab.py"""
import os, sys, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turtle_amd as TA
from turtle_amd import sharding, synth

n = int(os.environ.get("RAYS", "1000000"))
tmp = tempfile.mkdtemp(prefix="turtle_lean_")
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
order = np.argsort(-steps, kind="stable")

def timed(ids, reps=5):
    if ids is None:
        p0, dd = pos0, d
    else:
        ids_t = torch.as_tensor(ids, device=dev)
        p0, dd = pos0[ids_t].contiguous(), d[ids_t].contiguous()
    best = 1e9
    for _ in range(reps):
        p = p0.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream); st.trace(p, dd); b.record(stream); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best

medium = order[(steps[order] <= 2500) & (steps[order] >= 1500)]
only = os.environ.get("ONLY")
if only:
    sel = dict(top1=order[:1], top64=order[:64], med8=medium[:8], med64=medium[:64], med1024=medium[:1024])[only]
    print(only, f"{timed(sel, reps=3):.3f} ms", st.trace_stats())
    sys.exit(0)
res = dict(top1=timed(order[:1]), top8=timed(order[:8]), top64=timed(order[:64]), med8=timed(medium[:8]),
           med64=timed(medium[:64]), med1024=timed(medium[:1024]), all=timed(None))
print(os.path.basename(os.environ.get("TURTLE_AMD_LIBRARY", "in-tree")), " ".join(f"{k} {v:.3f}" for k, v in res.items()), "ms")
