#!/bin/bash
# Creep loop only when every live lane is on its line (rays before their 512th step get the wave to themselves).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
for c in 0 1 0 1; do run skip$c TURTLE_AMD_CREEP_SKIP=$c; done
for c in 1 1; do run skip${c}_lanes16 TURTLE_AMD_CREEP_SKIP=$c TURTLE_AMD_CREEP_LANES=16; done
for c in 0 1; do RAYS=4000000 run skip$c TURTLE_AMD_CREEP_SKIP=$c; done
