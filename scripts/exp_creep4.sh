#!/bin/bash
# The one-map creep loop: the GPU suite, the longest rays alone, C2 at 1 M / 4 M rays.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/creep4
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/creep4/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/creep4/pytest.log
timeout -k 10 200 python3 scripts/exp_longest.py 2>&1 | grep "alone"
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 8 --warmup 2 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
run c2 X=1; run c2 X=1
RAYS=4000000 run c2 X=1
