#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel, mean counter value per dispatch.
PMC_TAIL=<fraction>: only the last such fraction of each kernel's dispatches (a
workload that changes as it runs: the scattering walk's later generations)."""
import csv
import glob
import os
import re
import sys
from collections import defaultdict

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else "k_trace|k_cross"  # a regex
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        m = re.search(want, name)
        if not m:
            continue
        short = name[m.start():].split("(")[0] + " grid=" + row["Grid_Size"]
        acc[short][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
for kern, ctr in acc.items():
    print(f"== {kern}")
    for c in sorted(ctr):
        v = [x for _, x in sorted(ctr[c])]
        tail = float(os.environ.get("PMC_TAIL", "1"))
        v = v[int(len(v) * (1 - tail)):]
        print(f"  {c:28s} mean/dispatch {sum(v) / len(v):.6g}   (n={len(v)})")
