#!/bin/bash
# PMC passes over the scattering workload's kernels (k_step, k_bisect).
# usage (on the GPU box): bash scripts/profile_c5.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-c5pmc}
out=gpurun_out/$tag
mkdir -p $out
ARGS="--workload c5 --rays ${RAYS:-10000000} --scatter-steps ${GENS:-8} --steps 1 --warmup 1 --no-cpu ${EXTRA}"
run() { name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- \
      python3 bench.py $ARGS > $out/pmc_$name.log 2>&1; echo "pmc $name exit $?"; }
run A SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run B SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
run C FETCH_SIZE
run D WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run E TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
for k in k_step k_bisect; do python3 scripts/pmc_summary.py $out $k; done > $out/pmc_summary.txt
cat $out/pmc_summary.txt
rm -rf $out/pmc_A $out/pmc_B $out/pmc_C $out/pmc_D $out/pmc_E
