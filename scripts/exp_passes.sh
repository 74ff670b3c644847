#!/bin/bash
# Launch-by-launch timeline of a C2 trace (rocprofv3 kernel trace) under a few
# settings of the pass structure.  usage: exp_passes.sh "<ENV=.. ENV=..>" ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/passes; mkdir -p $out
i=0
for setting in "$@"; do
  i=$((i+1)); rm -rf $out/t
  ( export $setting; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/t -- \
      python3 bench.py --steps 5 --warmup 2 --no-cpu --workload ${WL:-c2} --rays ${RAYS:-0} > $out/log_$i.txt 2>&1 )
  echo "== $setting (exit $?)"
  python3 scripts/trace_timeline.py $out/t | tee $out/timeline_$i.txt
  python3 - $out/log_$i.txt <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print("   bench: kernel ms", round(d["kernel"]["ms"], 3), "steps/s %.4g" % d["value"])
PY
done
rm -rf $out/t
