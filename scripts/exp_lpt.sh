#!/bin/bash
# What would "longest rays first" buy phase B?  Rays ordered by their true step count
# (from a trace made beforehand), phase B at 3 waves per SIMD and at 1.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift; extra=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 6 --warmup 2 --no-cpu --rays ${RAYS:-1000000} $extra 2>/dev/null | python -c "$show" "$name"
}
run plain "" X=1
run plain_div16 "" TURTLE_AMD_TAIL_DIV=16
run longest_first "--sort-steps 1" X=1
run longest_first_div8 "--sort-steps 1" TURTLE_AMD_TAIL_DIV=8
run longest_first_div16 "--sort-steps 1" TURTLE_AMD_TAIL_DIV=16
run shortest_first "--sort-steps -1" X=1
run shortest_first_div16 "--sort-steps -1" TURTLE_AMD_TAIL_DIV=16
