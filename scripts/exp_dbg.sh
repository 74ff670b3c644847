#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/dbg
TAMD_DEBUG_BISECT=1 timeout -k 10 300 python3 bench.py --workload c5 --steps 1 --warmup 0 --no-cpu --rays ${RAYS:-10000000} --scatter-steps 120 > gpurun_out/dbg/out.txt 2> gpurun_out/dbg/err.txt
grep bisect gpurun_out/dbg/err.txt | awk 'NR%10==0'
