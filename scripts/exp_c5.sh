#!/bin/bash
# Kernel breakdown of the scattering workload (C5) at a given size.
# RAYS=<n>; EXTRA="<more bench.py arguments>"; TAG=<output name>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/${TAG:-c5}; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- \
    python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu --rays ${RAYS:-2000000} $EXTRA > $out/log.txt 2>&1
f=$(find $out/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(__import__("os").environ.get("TOP", "10"))]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.2f}")
PY
cut -c1-300 $out/log.txt | tail -1
rm -rf $out/t
