#!/bin/bash
# Evidence for profiles/: bench line, rocprofv3 kernel stats, PMC passes.
# usage (on the GPU box): bash scripts/profile_round.sh <tag> [workload: c2 | c3 | c4 | c5 | c5_step_n] [rays] [passes: "A B C D" | none]
# MATH=strict: the reference's arithmetic (the bench leg c2!strict)
# Every pass stays within the per-block counter slots of gfx950 (SQ 8, TCC 4: FETCH_SIZE
# takes 3, WRITE_SIZE 2; GRBM 2) and asks for nothing of the TA block.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-rXX}
wl=${2:-c2}
rays=${3:-0}
passes=${4:-A B C D}
math=${MATH:-fast}
export TURTLE_AMD_MATH=$math
WL="--workload $wl --rays $rays --also none --in-flight 1"   # one batch at a time: the kernels alone on the GPU
kernel="k_trace|k_cross"; [ "$wl" = c5 ] && kernel=k_walk
if [ "$wl" = c5_step_n ]; then
  # the walk through turtle_stepper_step_n, the caller's directions: two kernels a generation
  WL="--workload c5 --step-n 64 --rays $rays --also none --in-flight 1"; kernel="k_step|k_bisect"
fi
if [ "$wl" = c5_walk_n ]; then
  # ... through turtle_stepper_walk_n: one double of state a ray between the calls
  WL="--workload c5 --step-n 64 --compact --rays $rays --also none --in-flight 1"; kernel="k_step|k_bisect"
fi
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 600 python3 bench.py $WL > $out/bench.json 2> $out/bench.err
echo "bench exit $?"; tail -c 400 $out/bench.json; echo
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- \
    python3 bench.py $WL --no-cpu > $out/stats.log 2>&1
echo "stats exit $?"
cp $(find $out/stats -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv 2>/dev/null
python3 scripts/trace_timeline.py $out/stats "$kernel" > $out/timeline.txt 2>&1; tail -4 $out/timeline.txt
head -4 $out/kernel_stats.csv | cut -c1-220
run() { name=$1; shift
  case " $passes " in *" $name "*) ;; *) return;; esac
  timeout -k 10 400 rocprofv3 --pmc "$@" --output-format csv -d $out/pmc_$name -- \
      python3 bench.py $WL --steps 3 --warmup 1 --no-cpu > $out/pmc_$name.log 2>&1; echo "pmc $name exit $?"; }
run A SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run B SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH GRBM_GUI_ACTIVE
run C FETCH_SIZE SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT
run D WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run E TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
if [ "$passes" != none ]; then
  python3 scripts/pmc_summary.py $out "$kernel" > $out/pmc_summary.txt
  nrays=$(python3 -c "import json; print(json.loads(open('$out/bench.json').read().strip().splitlines()[-1])['config']['rays_per_gpu'])")
  python3 scripts/pmc_to_json.py $out/pmc_summary.txt $wl $nrays $math > $out/pmc.json
  cat $out/pmc_summary.txt
fi
rm -rf $out/stats $out/pmc_A $out/pmc_B $out/pmc_C $out/pmc_D $out/pmc_E
