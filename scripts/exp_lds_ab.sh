#!/bin/bash
# A/B of round 2's LDS block-staging variant (north_star's "DEM tiles staged from HBM into LDS"),
# in its own tree of that time: scratch/ab/lds_parent = commit e73f9b4 (8x8-node blocks, no staging),
# scratch/ab/lds_tree = commit e3280f7 (overlapping blocks of 7x7 cells + per-lane LDS-DMA staging) +
# profiles/variants/r02_lds_blockstage_switch.patch (TURTLE_AMD_LDS_STAGE=0|1).  Build both with
# `make -C turtle_amd/csrc` first (no GPU needed).  One gpurun call; log -> gpurun_out/r03_lds_ab.txt
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r03_lds_ab.txt; : > $out
show='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); k=d["kernel"]; print("   kernel ms %.3f  steps/s %.4g  (%s)" % (k["ms"], d["value"], d["config"]["workload"][:60]))'
run() { # tree, label, env, bench args
  echo "== $2" | tee -a $out
  ( cd $GRAFT_REPO_ROOT/scratch/ab/$1 && env $3 timeout -k 10 300 python3 bench.py --no-cpu $4 2>/dev/null | python3 -c "$show" ) | tee -a $out
}
for rep in 1 2; do
run lds_parent "parent (e73f9b4): 8x8-node blocks, cell fetch from HBM/L2; C2 1M rays" "X=1" "--steps 6 --warmup 2"
run lds_tree "variant, staging OFF: overlapping 7x7-cell blocks only; C2" "TURTLE_AMD_LDS_STAGE=0" "--steps 6 --warmup 2"
run lds_tree "variant, staging ON: each lane's block in LDS by LDS-DMA; C2" "TURTLE_AMD_LDS_STAGE=1" "--steps 6 --warmup 2"
done
run lds_parent "parent; C2 at 4M rays" "X=1" "--steps 3 --warmup 1 --rays 4000000"
run lds_tree "variant, staging OFF; C2 at 4M rays" "TURTLE_AMD_LDS_STAGE=0" "--steps 3 --warmup 1 --rays 4000000"
run lds_tree "variant, staging ON; C2 at 4M rays" "TURTLE_AMD_LDS_STAGE=1" "--steps 3 --warmup 1 --rays 4000000"
run lds_parent "parent; C3 (10M rays, 4x4 mosaic)" "X=1" "--workload c3 --steps 2 --warmup 1"
run lds_tree "variant, staging OFF; C3" "TURTLE_AMD_LDS_STAGE=0" "--workload c3 --steps 2 --warmup 1"
run lds_tree "variant, staging ON; C3" "TURTLE_AMD_LDS_STAGE=1" "--workload c3 --steps 2 --warmup 1"
run lds_parent "parent; C5 (10M rays x 64 single steps, 10x10 mosaic)" "X=1" "--workload c5 --steps 1 --warmup 1 --scatter-steps 64"
run lds_tree "variant (the step kernels read the overlapping blocks; no staging there); C5" "TURTLE_AMD_LDS_STAGE=0" "--workload c5 --steps 1 --warmup 1 --scatter-steps 64"
cat $out > /dev/null
