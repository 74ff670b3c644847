#!/bin/bash
# A/B of the working tree's library against a baseline build (default: round 2's) on C2, C3, C4
# usage: r03_ab.sh [baseline.so] [suite: 1|0]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
BASE=${1:-$GRAFT_REPO_ROOT/scratch/ab/libturtle_amd_r02.so}
if [ "${2:-1}" = 1 ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
  tail -5 gpurun_out/pytest_gpu.log
fi
STEPS=10 WL=c2 bash scripts/exp_ab.sh "TURTLE_AMD_LIBRARY=$BASE" "X=1"
STEPS=3 WL=c3 bash scripts/exp_ab.sh "TURTLE_AMD_LIBRARY=$BASE" "X=1"
STEPS=3 WL=c4 bash scripts/exp_ab.sh "TURTLE_AMD_LIBRARY=$BASE" "X=1"
