cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { echo -n "$1 $2: "; ( [ -n "$2" ] && export $2; timeout -k 10 300 python3 bench.py --steps ${K:-20} --warmup 2 --no-cpu --in-flight 3 --workload $1 --also none 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('ms/step %.3f  steps/s %.4g' % (d['ms_per_step'], d['value']))
" ); }
for rep in 1 2; do
for s in "" TURTLE_AMD_CREEP_LANES=4 TURTLE_AMD_CREEP_LANES=16 TURTLE_AMD_CREEP_LANES=32 TURTLE_AMD_DENSE_GO=16 TURTLE_AMD_DENSE_GO=32 TURTLE_AMD_PARK=24 TURTLE_AMD_PARK=48 TURTLE_AMD_DRAIN=16 TURTLE_AMD_SORT_LONG=90 TURTLE_AMD_SORT_LONG=160; do run c2 "$s"; done
done
K=6; for s in "" TURTLE_AMD_CREEP_LANES=8 TURTLE_AMD_CREEP_LANES=64 TURTLE_AMD_DENSE_GO=16 TURTLE_AMD_PARK=24; do run c4 "$s"; done
