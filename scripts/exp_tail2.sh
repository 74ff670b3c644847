#!/bin/bash
# Phase B grid width with the hand-over at drain in place (blocks = rays / 256 / DIV).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
for d in 1 6 8 12 16; do run div$d TURTLE_AMD_TAIL_DIV=$d; done
for d in 1 8 16; do RAYS=4000000 run div$d TURTLE_AMD_TAIL_DIV=$d; done
