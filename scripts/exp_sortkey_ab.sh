#!/bin/bash
# The ordered hand-over (TURTLE_AMD_SORT_KEY 0 / 1) at several batch sizes, one map and a stack.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pool_ab; mkdir -p $out
for spec in "c2 250000" "c2 1000000" "c2 2000000" "c2 4000000" "c2 8000000" "c3 1000000" "c3 4000000"; do
  set -- $spec
  for k in 0 1; do
    TURTLE_AMD_SORT_KEY=$k timeout -k 10 300 python3 bench.py --steps 6 --warmup 2 --no-cpu --in-flight 1 --workload $1 --rays $2 > $out/log.txt 2>&1 || { echo FAILED; tail -3 $out/log.txt; exit 1; }
    python3 - "$1 $2 sort_key=$k" $out/log.txt <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print(f"{sys.argv[1]:28s} {d['kernel']['ms']:8.3f} ms  {d['value']:.4g} steps/s")
PY
  done
done
