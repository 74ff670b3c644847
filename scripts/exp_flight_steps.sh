cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo -n "$1 steps $2 in flight $3 waves ${4:-fit}: "; ( [ -n "$4" ] && export TURTLE_AMD_TRACE_WAVES=$4; timeout -k 10 300 python3 bench.py --steps $2 --warmup 2 --no-cpu --in-flight $3 --workload $1 --also none 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step %.3f  steps/s %.4g' % (d['ms_per_step'], d['value']))
" ); }
for rep in 1 2; do
for k in 10 30; do
  run c2 $k 2; run c2 $k 3; run c2 $k 3 2; run c2 $k 4 2
done
done
run c3 6 2; run c3 6 3; run c3 6 3 2
