#!/bin/bash
# The hand-over / line threshold re-measured with the faster creep loop: C2 at 1 M and 4 M rays.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
for p in 512 384 256 192 128 64 512; do run park$p TURTLE_AMD_PARK=$p; done
for p in 512 256 128; do RAYS=4000000 run park$p TURTLE_AMD_PARK=$p; done
