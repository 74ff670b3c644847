#!/bin/bash
# Memory-side PMC passes over k_step (C5): TLB, L1 latency/stalls, L2 queue depth.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-c5mem}
out=gpurun_out/$tag
mkdir -p $out
ARGS="--workload c5 --rays ${RAYS:-10000000} --scatter-steps ${GENS:-8} --steps 1 --warmup 1 --no-cpu ${EXTRA}"
# (TA_* counters hang the run on this pool: left out)
run() { name=$1; shift
  case " ${PASSES:-F G H J K} " in *" $name "*) ;; *) return;; esac
  timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- \
      python3 bench.py $ARGS > $out/pmc_$name.log 2>&1; echo "pmc $name exit $?"; }
run F TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_GATE_EN1_sum
run G TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
run H TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum
run J TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS TCP_TCR_RDRET_STALL TCP_LFIFO_STALL_CYCLES
run K GRBM_GUI_ACTIVE TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
python3 scripts/pmc_summary.py $out k_step > $out/pmc_summary.txt
cat $out/pmc_summary.txt
rm -rf $out/pmc_?
