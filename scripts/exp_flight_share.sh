cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo -n "$1 steps $2 share $3: "; TURTLE_AMD_IN_FLIGHT_SHARE=$3 timeout -k 10 300 python3 bench.py --steps $2 --warmup 2 --no-cpu --in-flight 3 --workload $1 --also none 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step %.3f  steps/s %.4g' % (d['ms_per_step'], d['value']))
"; }
for s in 0 1 2 0 1 2; do run c3 6 $s; done
for s in 0 1 2; do run c4 6 $s; run c2 10 $s; run c2 20 $s; done
