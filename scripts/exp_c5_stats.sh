#!/bin/bash
# Kernel breakdown of the scattering workload (rocprofv3 kernel stats), optional env settings
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/c5stats; mkdir -p $out; rm -rf $out/t
( export ${1:-X=1}; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- \
    python3 bench.py --workload c5 --steps 1 --warmup 1 --no-cpu --rays ${RAYS:-0} > $out/log.txt 2>&1 )
echo "exit $?"
f=$(find $out/t -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.1f}")
PY
grep '^{' $out/log.txt | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('value %.4g' % d['value'], 'ms/gen', d['kernel']['ms_per_generation'], 'frac', d['roofline']['frac'])"
rm -rf $out/t
