#!/bin/bash
# Phase B grid width (blocks = rays / 256 / DIV, at most what fits) and park threshold.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
for d in 16 8 4 2 1; do run div$d TURTLE_AMD_TAIL_DIV=$d; done
for p in 128 256 384 768 1024; do run park$p TURTLE_AMD_PARK=$p; done
for p in 256 1024; do RAYS=4000000 run park$p TURTLE_AMD_PARK=$p; done
