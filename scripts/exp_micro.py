#!/usr/bin/env python3
"""The path's bandwidth-shaped batch kernels alone (rows a5-a7, a11 of SURVEY 8): points/s and the
algorithmic GB/s of the ECEF transforms, the bilinear lookup and stepper_position over 50 M points.
bench.py reports the same under `micro`."""
import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
import turtle_amd as TA
from turtle_amd import synth, sharding
import tempfile
dev = torch.device("cuda", 0)
stream = torch.cuda.Stream(device=0); torch.cuda.set_stream(stream); TA.set_stream(stream)
n = 50_000_000
tmp = tempfile.mkdtemp()
tile = TA.Map.load(synth.write_hgt(tmp, 45, 3))
st = TA.Stepper(); st.add_map(tile, 0.0)
g = torch.Generator(device=dev); g.manual_seed(1)
lat = 45.05 + 0.9 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
lon = 3.05 + 0.9 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
h = 1000.0 * torch.rand(n, dtype=torch.float64, device=dev, generator=g)
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
ecef = TA.ecef_from_geodetic(lat, lon, h)
for name, f, bytes_ in (
    ("ecef_from_geodetic_n", lambda: TA.ecef_from_geodetic(lat, lon, h), 48),
    ("ecef_to_geodetic_n strict", lambda: TA.ecef_to_geodetic(ecef), 48),
    ("map.elevation_n", lambda: tile.elevation(lon, lat), 16 + 8 + 4 + 8),
    ("stepper.position_n", lambda: st.position(lat, lon, 500.0), 24 + 24 + 4 + 8),
):
    ms = timeit(f)
    print(f"{name:28s} {ms:7.3f} ms  {n / ms * 1e3:.3g} points/s  {n * bytes_ / ms / 1e6:.0f} GB/s")
TA.set_math("fast")
