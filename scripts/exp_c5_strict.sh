#!/bin/bash
# strict arithmetic, C5: k_step with and without the 4-waves-per-SIMD register cap
export RAYS=${RAYS:-4000000} TOP=2 TURTLE_AMD_MATH=strict
echo "== strict, capped";   TAG=c5_s1 bash scripts/exp_c5.sh | head -2
echo "== strict, uncapped"; TAG=c5_s2 TURTLE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/var/lib_noattr.so bash scripts/exp_c5.sh | head -2
