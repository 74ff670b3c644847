#!/bin/bash
# Steps per trip of the lean creep loop (variants of the library built with kCreepUnroll = 2, 3, 6).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
run() { name=$1; shift; w=$1; shift; r=$1; shift
  env "$@" timeout -k 10 300 python bench.py --workload $w --rays $r --steps 8 --warmup 2 --no-cpu 2>/dev/null | python -c "$show" "$name"
}
for u in 4 2 3 6 4; do
  lib=$GRAFT_REPO_ROOT/scratch/var/lib_u$u.so; [ $u = 4 ] && lib=$GRAFT_REPO_ROOT/turtle_amd/libturtle_amd.so
  run c2_unroll$u c2 1000000 TURTLE_AMD_LIBRARY=$lib
done
for u in 4 2 6; do
  lib=$GRAFT_REPO_ROOT/scratch/var/lib_u$u.so; [ $u = 4 ] && lib=$GRAFT_REPO_ROOT/turtle_amd/libturtle_amd.so
  run c3_unroll$u c3 10000000 TURTLE_AMD_LIBRARY=$lib
done
