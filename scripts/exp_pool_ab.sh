#!/bin/bash
# A/B of pool settings / variant builds on one box: kernel ms of C2, C4 and C3, each alone.
# usage: exp_pool_ab.sh "<ENV=..>" ...  (e.g. TURTLE_AMD_POOL=0, TURTLE_AMD_LIBRARY=$PWD/variants/x.so)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pool_ab; mkdir -p $out
for wl in ${WLS:-c2 c4 c3}; do
  steps=3; [ $wl = c2 ] && steps=10
  for setting in "$@"; do
    ( export $setting; timeout -k 10 300 python3 bench.py --steps $steps --warmup 2 --no-cpu --in-flight 1 --workload $wl > $out/log.txt 2>&1 ) || { echo "$wl $setting FAILED"; tail -3 $out/log.txt; exit 1; }
    python3 - "$wl" "$setting" $out/log.txt <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(f"{sys.argv[1]:3s} {d['kernel']['ms']:8.3f} ms  {d['value']:.4g} steps/s   {sys.argv[2].replace('TURTLE_AMD_LIBRARY=', '').split('/')[-1]}")
PY
  done
done
