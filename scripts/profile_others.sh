#!/bin/bash
# Evidence for profiles/ beside the headline: the C3 and C5 bench lines and C5's
# kernel breakdown.  usage (on the GPU box): bash scripts/profile_others.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-rXX}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 400 python3 bench.py --workload c3 --no-cpu --steps 5 > $out/c3_bench.json 2> $out/c3_bench.err
echo "c3 exit $?"; cut -c1-400 $out/c3_bench.json
timeout -k 10 400 python3 bench.py --workload c5 --no-cpu --steps 2 --warmup 1 > $out/c5_bench.json 2> $out/c5_bench.err
echo "c5 exit $?"; cut -c1-400 $out/c5_bench.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- \
    python3 bench.py --workload c5 --no-cpu --steps 1 --warmup 1 > $out/c5_stats.log 2>&1
echo "c5 stats exit $?"
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/c5_kernel_stats.csv 2>/dev/null
head -8 $out/c5_kernel_stats.csv | cut -c1-200
rm -rf $out/stats
