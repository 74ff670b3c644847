#!/bin/bash
# Per-phase kernel durations (rocprofv3 kernel trace) for a few park thresholds.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/phases
mkdir -p $out
for park in ${@:-512}; do
  rm -rf $out/t
  TURTLE_AMD_PARK=$park timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/t -- \
      python3 bench.py --steps 5 --warmup 1 --no-cpu --rays ${RAYS:-1000000} > $out/log_$park.txt 2>&1
  f=$(find $out/t -name "*kernel_stats.csv" | head -1)
  echo "park $park rays ${RAYS:-1000000}"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'k_trace' in r['Name']: print('   ', r['Name'][30:75], 'calls', r['Calls'], 'avg_us', float(r['AverageNs'])/1000)
"
done
rm -rf $out/t
