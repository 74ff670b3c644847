#!/bin/bash
# Trace kernels held to 168 registers (3 waves per SIMD): GPU suite, C2 fast/strict, C3.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/cap3
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/cap3/pytest.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/cap3/pytest.log
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift; w=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload $w --steps 4 --warmup 1 --no-cpu 2>/dev/null | python -c "$show" "$name"
}
run c2_fast c2 X=1
run c2_strict c2 TURTLE_AMD_MATH=strict
run c3_fast c3 X=1
run c3_strict c3 TURTLE_AMD_MATH=strict
