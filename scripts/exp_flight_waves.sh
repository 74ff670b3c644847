#!/bin/bash
# Batches in flight x waves a SIMD of the trace kernels (TURTLE_AMD_TRACE_WAVES), C2 and C4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for wl in ${WLS:-c2 c4}; do
  steps=12; [ $wl != c2 ] && steps=6
  echo "#### $wl"
  for f in 1 2 3 4; do
    for w in 0 1 2; do
      [ $f = 1 ] && [ $w != 0 ] && continue
      echo -n "in flight $f, waves ${w} (0: as many as fit): "
      ( [ $w != 0 ] && export TURTLE_AMD_TRACE_WAVES=$w; timeout -k 10 300 python3 bench.py --steps $steps --warmup 2 --no-cpu --in-flight $f --workload $wl --also none 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step %.3f  steps/s %.4g  (alone %.3f ms)' % (d['ms_per_step'], d['value'], d['kernel']['ms']))
" )
    done
  done
done
