cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
show='import sys,json; d=json.loads(sys.stdin.readline()); k=d["kernel"]; print(sys.argv[1], "rays",d["config"]["rays_per_gpu"],"kernel_ms",round(k["ms"],2),"Gsteps/s",round(k["gpu_steps_per_s"]/1e9,2))'
for srt in 0 8 64 256; do
  for r in 1000000 16000000; do
    timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu --rays $r --sort $srt 2>/dev/null | python -c "$show" sort$srt
  done
done
for srt in 0 64; do
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc_sort$srt -- python3 bench.py --steps 2 --warmup 1 --no-cpu --rays 16000000 --sort $srt > /dev/null 2>&1
echo "sort $srt (16M rays):"; python3 scripts/pmc_summary.py gpurun_out/pmc_sort$srt k_trace | grep -v "^=="
rm -rf gpurun_out/pmc_sort$srt
done
