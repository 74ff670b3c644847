#!/bin/bash
# Is the line kernel bound by gather latency?  Kernel time against waves per SIMD,
# and with the rays ordered by origin (fewer distinct cells in flight).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --rays ${RAYS:-4000000} $EXTRA 2>/dev/null | python -c "$show" "$name"
}
run waves2-park0 TURTLE_AMD_PARK=0
run waves1-park0 TURTLE_AMD_PARK=0 TURTLE_AMD_TRACE_WAVES=1
EXTRA="--sort 64" run waves2-park0-sorted TURTLE_AMD_PARK=0
EXTRA="--sort 64" run waves1-park0-sorted TURTLE_AMD_PARK=0 TURTLE_AMD_TRACE_WAVES=1
