#!/bin/bash
# Creep-lane threshold on C3 (10 M rays through a stack) and C2 at 4 M rays.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift; w=$1; shift; r=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload $w --rays $r --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "$show" "$name"
}
for c in 8 12 16 24 32 64; do run c3_creep$c c3 10000000 TURTLE_AMD_CREEP_LANES=$c; done
for c in 8 16 32; do run c2_4M_creep$c c2 4000000 TURTLE_AMD_CREEP_LANES=$c; done
bash scripts/exp_c3_phases.sh
