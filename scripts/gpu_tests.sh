#!/bin/bash
# GPU parity tests with bounded run time and progress written to gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 ${1:-300} python -m pytest tests -m gpu -x -q -s > gpurun_out/pytest_gpu.log 2>&1
echo "pytest exit $?" | tee -a gpurun_out/pytest_gpu.log
grep -E "1M rays|passed|failed|Error|assert|hgt 10k|fast transform" gpurun_out/pytest_gpu.log | head -20
