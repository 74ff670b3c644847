#!/bin/bash
# Phase split of the one-stack trace (C3) from the rocprofv3 kernel trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/c3ph; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- \
    python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu ${EXTRA} > $out/log.txt 2>&1
f=$(find $out/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg_ms {float(r['AverageNs'])/1e6:9.3f}")
PY
rm -rf $out/t
