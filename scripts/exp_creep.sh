#!/bin/bash
# Creep loop: kernel time against the number of live lanes at which it engages.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
for c in 4 8 12 16 24 32 64; do run two-creep$c TURTLE_AMD_PARK2=0 TURTLE_AMD_CREEP_LANES=$c; done
for c in 16 32; do run three-creep$c TURTLE_AMD_CREEP_LANES=$c; done
