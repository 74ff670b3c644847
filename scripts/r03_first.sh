#!/bin/bash
# round 3, first visit: the suite, then A/B against round 2's library on C2, C3, C4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
R02=$GRAFT_REPO_ROOT/scratch/ab/libturtle_amd_r02.so
STEPS=10 WL=c2 bash scripts/exp_ab.sh "TURTLE_AMD_LIBRARY=$R02" "X=1" "TURTLE_AMD_MATH=strict TURTLE_AMD_LIBRARY=$R02" "TURTLE_AMD_MATH=strict"
STEPS=3 WL=c3 bash scripts/exp_ab.sh "TURTLE_AMD_LIBRARY=$R02" "X=1"
STEPS=3 WL=c4 bash scripts/exp_ab.sh "TURTLE_AMD_LIBRARY=$R02" "X=1"
