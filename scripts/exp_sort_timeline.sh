#!/bin/bash
# Per-pass durations of a C2 trace under settings of the sorted hand-over
# usage: exp_sort_timeline.sh "<ENV=..>" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sort_tl; mkdir -p $out
i=0
for setting in "$@"; do
  i=$((i+1))
  ( export $setting
    timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t$i -- \
      python3 bench.py --workload ${WL:-c2} --also none --no-cpu --in-flight 1 --steps 4 --warmup 1 > $out/log_$i.txt 2>&1 )
  echo "== $setting (exit $?)"
  python3 scripts/trace_timeline.py $out/t$i "k_trace|k_cross|k_flow" | tail -${TL:-6}
  rm -rf $out/t$i
done
