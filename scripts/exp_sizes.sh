cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
for r in 1000000 4000000 16000000; do
    timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --rays $r 2>/dev/null | python -c "$show" ${1:-run}
done
