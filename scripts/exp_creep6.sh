#!/bin/bash
# Creep loop engaged by the share of lanes stepping on their lines: C2 at 1 M (and 4 M) rays.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
for c in 0 90 75 50 30 15 0; do run share$c TURTLE_AMD_CREEP_SHARE=$c; done
for c in 0 75 30; do RAYS=4000000 run share$c TURTLE_AMD_CREEP_SHARE=$c; done
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "trace or long or oracle or properties or degenerate" 2>&1 | tail -2
