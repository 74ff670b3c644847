#!/bin/bash
# Pass T of a trace (the last rays of the lined pass, packed): off (GIVE_AT=0) and where waves give
# usage: exp_tail_ab.sh [suite: 1|0] [base.so]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
if [ "${1:-1}" = 1 ]; then
  timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
  tail -5 gpurun_out/pytest_gpu.log
fi
BASE=${2:-$GRAFT_REPO_ROOT/scratch/ab/sortA.so}
for wl in c2 c4 c3; do
  steps=3; [ $wl = c2 ] && steps=10
  echo "#### $wl"
  STEPS=$steps WL=$wl bash scripts/exp_ab.sh "TURTLE_AMD_LIBRARY=$BASE" "TURTLE_AMD_GIVE_AT=0" "X=1" "TURTLE_AMD_GIVE_AT=4" \
      "TURTLE_AMD_GIVE_AT=16" "TURTLE_AMD_GIVE_AT=32" "TURTLE_AMD_GIVE_AT=16 TURTLE_AMD_TAIL_WAVES=2" "TURTLE_AMD_LIBRARY=$BASE"
done
