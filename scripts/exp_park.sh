#!/bin/bash
# Thresholds of the three-phase fast trace: kernel time on C2 (1 M rays).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
run two TURTLE_AMD_PARK2=0
for p2 in 544 576 640 768; do run a512-b$p2 TURTLE_AMD_PARK2=$p2; done
for cfg in "256 288" "256 320" "256 384" "128 160" "128 256" "384 448"; do set -- $cfg; run a$1-b$2 TURTLE_AMD_PARK=$1 TURTLE_AMD_PARK2=$2; done
