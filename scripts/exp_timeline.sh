#!/bin/bash
# Start / end of the trace kernels of the last bench step (rocprofv3 kernel trace).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/timeline
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- \
    python3 bench.py --steps 2 --warmup 1 --no-cpu --rays ${RAYS:-1000000} > $out/log.txt 2>&1
f=$(find $out/t -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_trace" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-3:] if len(rows) >= 3 else rows
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    name = r["Kernel_Name"]
    tag = name[name.find("k_trace"):][:32]
    print(f"{tag:34s} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s} start {(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  end {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} us")
PY
rm -rf $out/t
