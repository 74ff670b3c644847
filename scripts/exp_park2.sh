#!/bin/bash
# A third phase after the hand-over at drain: thresholds 768..3072, C2 at 1 M rays.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
run two TURTLE_AMD_PARK2=0
for p2 in 768 1024 1536 2048 3072; do run b$p2 TURTLE_AMD_PARK2=$p2; done
run two TURTLE_AMD_PARK2=0
