#!/bin/bash
# A/B of library builds / settings on the same box: bench kernel ms of each.
# usage: exp_ab.sh "<ENV=.. ENV=..>" ...   (TURTLE_AMD_LIBRARY=path selects another build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/ab; mkdir -p $out
i=0
for setting in "$@"; do
  i=$((i+1))
  ( export $setting; timeout -k 10 300 python3 bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu --in-flight ${IN_FLIGHT:-1} --workload ${WL:-c2} --rays ${RAYS:-0} > $out/log_$i.txt 2>&1 )
  echo "== $setting (exit $?)"
  python3 - $out/log_$i.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print("   kernel ms", round(d["kernel"]["ms"], 3), "ms/step", round(d["ms_per_step"], 3), "steps/s %.4g" % d["value"], "samples", d["kernel"].get("samples_per_launch", d["kernel"].get("samples_per_pass")))
PY
done
