#!/bin/bash
# Run-time knobs of a trace, one at a time against the defaults (kernel ms of bench.py --no-cpu)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for wl in ${WLS:-c2 c4 c3}; do
  steps=3; [ $wl = c2 ] && steps=10
  echo "#### $wl"
  STEPS=$steps WL=$wl bash scripts/exp_ab.sh "X=1" "$@" "X=2"
done
