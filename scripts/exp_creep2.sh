#!/bin/bash
# creep loop with the cell fetch inside: lanes threshold sweep, C2 at 1 M rays
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "trace or long or oracle or properties" 2>&1 | tail -2
for c in 8 4 12 16 24 32 64; do run creep$c TURTLE_AMD_CREEP_LANES=$c; done
