#!/usr/bin/env python3
"""Per-dispatch durations of the trace kernels from a rocprofv3 --kernel-trace CSV:
the passes of the LAST trace of the run, in launch order, and their sum.

usage: trace_timeline.py <dir with *_kernel_trace.csv> [kernel-name regex]"""
import csv
import glob
import os
import re
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_trace|k_cross"
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
if not files:
    sys.exit("no kernel trace under " + root)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sel = [r for r in rows if re.search(want, r[2])]
if not sel:
    sys.exit("no dispatch of " + want)
# group dispatches separated by less than 200 us from each other: one trace call
groups, cur = [], [sel[0]]
for r in sel[1:]:
    if r[0] - cur[-1][1] < 200_000:
        cur.append(r)
    else:
        groups.append(cur)
        cur = [r]
groups.append(cur)


def short(name):
    m = re.search(r"k_\w+<[^>]*>", name)
    return m.group(0) if m else name[:60]


spans = [(g[-1][1] - g[0][0]) / 1e3 for g in groups]
print(f"{len(groups)} calls; span of each (us): median {sorted(spans)[len(spans) // 2]:.1f}, "
      f"min {min(spans):.1f}, max {max(spans):.1f}")
g = groups[-1]
t0 = g[0][0]
for a, b, name in g:
    print(f"  +{(a - t0) / 1e3:9.1f} us  {(b - a) / 1e3:9.1f} us  {short(name)}")
print(f"  last call: kernels {sum(b - a for a, b, _ in g) / 1e3:.1f} us, span {(g[-1][1] - t0) / 1e3:.1f} us")
