#!/bin/bash
# counters of the lined kernel for one probe of exp_lean_ab.py (ONLY=med64 ...) under each library given
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/leanpmc; mkdir -p $out
for L in "$@"; do
  name=$(basename $L .so)
  rm -rf $out/$name
  TURTLE_AMD_LIBRARY=$L timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/$name -- python3 scripts/exp_lean_ab.py > $out/$name.log 2>&1
  echo "== $name (exit $?)"; tail -1 $out/$name.log
  python3 scripts/pmc_summary.py $out/$name "k_trace<1, true, true" | tail -12
  rm -rf $out/$name
done
