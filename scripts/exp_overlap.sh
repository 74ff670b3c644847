#!/bin/bash
# Sensitivities of the overlapped two-phase trace.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 2 --warmup 1 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
run base
run lanes64 TURTLE_AMD_LIVE_LANES=64
run lanes1 TURTLE_AMD_LIVE_LANES=1
run park2048 TURTLE_AMD_PARK=2048
run park128 TURTLE_AMD_PARK=128
run off TURTLE_AMD_OVERLAP=0
