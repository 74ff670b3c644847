#!/bin/bash
# GPU suite, then C2 at 1 M and 4 M rays and C3 (kernel time, steps/s).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/quick
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/quick/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/quick/pytest.log
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],3),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
for rep in 1 2; do
timeout -k 10 200 python3 bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python3 -c "$show" c2
done
timeout -k 10 200 python3 bench.py --steps 4 --warmup 1 --no-cpu --rays 4000000 2>/dev/null | python3 -c "$show" c2
[ -n "$SKIP_C3" ] || timeout -k 10 300 python3 bench.py --workload c3 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python3 -c "$show" c3
