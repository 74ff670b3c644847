#!/bin/bash
# bench.py with batches in flight, 10 / 20 / 30 passes (C2), then C4 and C3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo -n "$1 steps $2 in flight $3: "; timeout -k 10 300 python3 bench.py --steps $2 --warmup 2 --no-cpu --in-flight $3 --workload $1 --also none 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step %.3f  steps/s %.4g  alone %.3f ms' % (d['ms_per_step'], d['value'], d['kernel']['ms']))
"; }
for rep in 1 2; do for k in 10 20 30; do run c2 $k 3; done; done
run c2 10 2; run c2 10 4; run c4 6 3; run c3 6 3
