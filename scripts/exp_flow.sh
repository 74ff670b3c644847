#!/bin/bash
# The chain of passes of a large batch (run_trace, FLOW): the property test, the suite with the chain
# forced on every trace, then A/B on C4 and C3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_properties.py -m gpu -q -x -k "chain" > gpurun_out/pytest_chain.log 2>&1
echo "chain test exit $?"; tail -3 gpurun_out/pytest_chain.log
if [ "${1:-1}" = 1 ]; then
  TURTLE_AMD_FLOW_MIN=1 timeout -k 10 700 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_flow.log 2>&1
  echo "suite (chain forced) exit $?"; tail -3 gpurun_out/pytest_gpu_flow.log
fi
WLS="${WLS:-c4 c3}" bash scripts/exp_knobs.sh TURTLE_AMD_FLOW_ROUNDS=0 TURTLE_AMD_FLOW_ROUNDS=1 TURTLE_AMD_FLOW_ROUNDS=2 TURTLE_AMD_FLOW_ROUNDS=5 TURTLE_AMD_FLOW_ROUNDS=8 TURTLE_AMD_FLOW_ROUNDS=0
