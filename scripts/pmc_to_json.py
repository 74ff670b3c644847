#!/usr/bin/env python3
"""Turn a pmc_summary.txt (scripts/pmc_summary.py) into the profiles/*_pmc.json that
bench.py reads for `roofline.traffic`: per trace_n call, all k_trace kernels summed.
usage: pmc_to_json.py <pmc_summary.txt> <workload> <rays_per_gpu> <math> > out.json"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
h = hashlib.sha256()
for name in ("device.hip", "internal.h"):      # as bench.py's device_source_hash()
    with open(os.path.join(ROOT, "turtle_amd", "csrc", name), "rb") as f:
        h.update(f.read())

path, workload, rays, math = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
kernels, cur = {}, None
for line in open(path):
    if line.startswith("== "):
        cur = kernels.setdefault(line[3:].strip(), {})
    elif cur is not None and "mean/dispatch" in line:
        name, rest = line.split("mean/dispatch")
        cur[name.strip()] = float(rest.split()[0])
fetch = sum(k.get("FETCH_SIZE", 0.0) for k in kernels.values())
write = sum(k.get("WRITE_SIZE", 0.0) for k in kernels.values())
json.dump({
    "workload": workload, "rays_per_gpu": rays, "math": math,
    "source_hash": h.hexdigest()[:16],
    "how": "rocprofv3 --pmc, separate passes (scripts/profile_round.sh); per trace_n call = phase A + phase B + "
           "k_cross kernels; FETCH_SIZE/WRITE_SIZE are KB of L2<->fabric requests (Infinity-Cache hits "
           "included); NOT multiplied by the guide's x2 wide-stream correction, which is uncalibrated "
           "for 4-byte gathers",
    "fetch_kb_per_launch": fetch, "write_kb_per_launch": write,
    "traffic_bytes_per_launch": 1024.0 * (fetch + write),
    "kernels": kernels,
}, sys.stdout, indent=1)
