#!/bin/bash
# GPU suite, then C3 (and C2 for reference).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c3q
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/c3q/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/c3q/pytest.log
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift; w=$1; shift; r=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload $w --rays $r --steps 4 --warmup 1 --no-cpu 2>/dev/null | python -c "$show" "$name"
}
run c3 c3 10000000 X=1
run c3 c3 10000000 X=1
run c3_1M c3 1000000 X=1
run c2 c2 1000000 X=1
