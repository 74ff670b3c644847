#!/bin/bash
# The sorted hand-over from phase A to the lined pass: off (0) and the figure that sorts
# usage: exp_sort_ab.sh [suite: 1|0]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "${1:-1}" = 1 ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
  tail -5 gpurun_out/pytest_gpu.log
fi
for wl in c2 c4 c3; do
  steps=3; [ $wl = c2 ] && steps=10
  echo "#### $wl"
  STEPS=$steps WL=$wl bash scripts/exp_ab.sh "TURTLE_AMD_SORT_LONG=0" "X=1" "TURTLE_AMD_SORT_LONG=60" \
      "TURTLE_AMD_SORT_LONG=200" "TURTLE_AMD_SORT_LONG=400" "TURTLE_AMD_SORT_LONG=0" "X=2"
done
