#!/bin/bash
# One GPU-box visit: parity tests, then the bench in both arithmetic modes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -15 gpurun_out/pytest_gpu.log
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2), "value", round(d["value"]/1e9,2))'
for mode in fast strict; do
  for r in 1000000; do
    TURTLE_AMD_MATH=$mode timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --rays $r 2>/dev/null | python -c "$show" $mode
  done
done
timeout -k 10 600 python bench.py --workload c3 --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "$show" c3-fast

