#!/bin/bash
# PMC passes over the bench (run on the GPU box).  Counters are collected in
# their own runs (no --stats/--kernel-trace mix beyond what --pmc implies).
# usage: scripts/pmc_profile.sh <tag> [bench args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p $out
run() { # name counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- \
      python3 bench.py --steps 2 --warmup 1 --no-cpu $BENCH_ARGS > $out/$name.log 2>&1
  echo "$name exit $?"
}
BENCH_ARGS="$*"
run A SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run B SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
run C FETCH_SIZE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
run D WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python3 scripts/pmc_summary.py $out | tee $out/summary.txt
