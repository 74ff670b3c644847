#!/bin/bash
# Phase-by-phase durations (rocprofv3 --kernel-trace) of C2's trace for several library builds.
# usage: exp_timeline_ab.sh <path to libturtle_amd.so | ""> ...    ("" = the in-tree build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for lib in "$@"; do
  i=$((i+1))
  out=gpurun_out/tl_$i; rm -rf $out; mkdir -p $out
  if [ -n "$lib" ]; then export TURTLE_AMD_LIBRARY=$lib; else unset TURTLE_AMD_LIBRARY; fi
  rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --steps 5 --warmup 2 --no-cpu --workload ${WL:-c2} > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
  echo "== ${lib:-in-tree}"
  python3 scripts/trace_timeline.py $out | tail -8
done
