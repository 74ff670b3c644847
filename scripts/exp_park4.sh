#!/bin/bash
# Hand-over threshold above 512 for the large batches: C3 (10 M rays, stack) and C2 at 4 M rays.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift; w=$1; shift; r=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload $w --rays $r --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "$show" "$name"
}
for p in 512 768 1024 2048; do run c3_park$p c3 10000000 TURTLE_AMD_PARK=$p; done
run c3_creep0 c3 10000000 TURTLE_AMD_CREEP_LANES=0
for p in 512 1024 2048; do run c2_4M_park$p c2 4000000 TURTLE_AMD_PARK=$p; done
for p in 768 1024; do run c2_1M_park$p c2 1000000 TURTLE_AMD_PARK=$p; done
