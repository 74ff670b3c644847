#!/bin/bash
# Creep loop with the next cell fetched per group, and a minimum number of groups per entry.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/creep7
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/creep7/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/creep7/pytest.log
timeout -k 10 200 python3 scripts/exp_longest.py 2>&1 | grep "alone" | head -4
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
for c in 1 2 4 8 1; do run groups$c TURTLE_AMD_CREEP_GROUPS=$c; done
for c in 1 4; do RAYS=4000000 run groups$c TURTLE_AMD_CREEP_GROUPS=$c; done
