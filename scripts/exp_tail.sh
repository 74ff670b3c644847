cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for r in 250000 1000000 4000000 16000000; do
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --rays $r 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('rays',d['config']['rays_per_gpu'],'ms',round(d['kernel']['ms'],2),'Gsteps/s',round(d['kernel']['gpu_steps_per_s']/1e9,2))"
done
TURTLE_AMD_TRACE_WAVES=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --rays 4000000 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('waves1 rays',d['config']['rays_per_gpu'],'ms',round(d['kernel']['ms'],2),'Gsteps/s',round(d['kernel']['gpu_steps_per_s']/1e9,2))"
