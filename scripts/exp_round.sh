#!/bin/bash
# One GPU call: the parity suite, then C5 kernel breakdown, C3 and C2 bench lines.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/round; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest exit $?"; tail -3 $out/pytest.log
RAYS=${RAYS:-10000000} bash scripts/exp_c5.sh
for w in c3 c2; do
  timeout -k 10 300 python3 bench.py --workload $w --no-cpu --steps 5 > $out/$w.json 2> $out/$w.err
  python3 - $out/$w.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["workload"][:40], "ms", round(d["kernel"]["ms"], 3), "steps/s %.3g" % d["value"])
PY
done
