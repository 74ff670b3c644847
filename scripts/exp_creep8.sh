#!/bin/bash
# Groups of creep steps between two general passes for lanes that are bisecting or not yet on their line.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
for c in 1 2 3 4 6 8 1; do run groups$c TURTLE_AMD_CREEP_GROUPS=$c; done
for c in 1 3 6; do RAYS=4000000 run groups$c TURTLE_AMD_CREEP_GROUPS=$c; done
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q -k "trace or long or oracle or properties or degenerate" 2>&1 | tail -2
