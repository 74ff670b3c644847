#!/bin/bash
# Live-lane threshold of the creep loop, now that lanes that can go on do: C2 at 1 M (and 4 M) rays.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
for c in 8 12 16 24 32 48 64 8; do run creep$c TURTLE_AMD_CREEP_LANES=$c; done
for c in 8 16 32 64; do RAYS=4000000 run creep$c TURTLE_AMD_CREEP_LANES=$c; done
