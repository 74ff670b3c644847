cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
for rep in 1 2; do
for cfg in "1 20 512" "1 30 512" "1 40 512" "1 30 384" "1 30 256" "2 30 512"; do
  set -- $cfg
  TURTLE_AMD_CREEP=$1 TURTLE_AMD_CREEP_LANES=8 TURTLE_AMD_RADIUS=$2 TURTLE_AMD_PARK=$3 timeout -k 10 300 python bench.py --steps 5 --warmup 1 --no-cpu --rays 1000000 2>/dev/null | python -c "$show" "creep$1-R$2-park$3"
done
done
