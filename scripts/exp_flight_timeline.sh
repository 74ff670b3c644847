#!/bin/bash
# The dispatches of a bench run with batches in flight, in launch order (-> profiles/r03_c2_in_flight_timeline.txt)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/flight_tl; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- \
    python3 bench.py --workload ${1:-c2} --also none --no-cpu --in-flight ${2:-3} --steps 9 --warmup 2 > $out/log.txt 2>&1
echo "exit $?"
python3 - $out <<'PY'
import csv, glob, os, re, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "t", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if re.search("k_trace|k_cross", r["Kernel_Name"]):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
# the timed region: the last 9 passes of 3 kernels each
rows = rows[-27:]
t0 = rows[0][0]
print("# start us   end us   duration us   queue   kernel      (a pass = closed-form pass, lined pass, k_cross; three batches in flight)")
busy_end = t0
for a, b, name, q in rows:
    m = re.search(r"k_\w+<[^>]*>", name)
    print(f"{(a - t0) / 1e3:9.1f} {(b - t0) / 1e3:9.1f} {(b - a) / 1e3:9.1f}   q{q}   {m.group(0) if m else name[:40]}")
span = (max(r[1] for r in rows) - t0) / 1e3
print(f"# 9 passes in {span:.1f} us = {span / 9:.1f} us a pass; the kernels' own durations add up to {sum(b - a for a, b, _, _ in rows) / 1e3:.1f} us")
PY
