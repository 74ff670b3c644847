#!/bin/bash
# Kernel time on C2 for a few settings of the line / pass-policy tunables.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2), "samples/step", round(k.get("samples_per_step",0),3))'
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --rays ${RAYS:-1000000} 2>/dev/null | python -c "$show" "$name"
}
run base
run park0 TURTLE_AMD_PARK=0
for r in 500 1000 2000; do run range$r TURTLE_AMD_LINE_RANGE=$r;  done
for w in 1 2 3 6; do run wait$w TURTLE_AMD_WAIT_SHARE=$w; done
for w in 1 2 3; do run park0-wait$w TURTLE_AMD_PARK=0 TURTLE_AMD_WAIT_SHARE=$w; done
for l in 2 4 16 64; do run lanes$l TURTLE_AMD_TAIL_LANES=$l; done
RAYS=16000000 run big
RAYS=16000000 run big-wait2 TURTLE_AMD_WAIT_SHARE=2
