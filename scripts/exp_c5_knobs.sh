#!/bin/bash
# k_step / k_bisect of C5 under a few launch variants, one after the other.
export RAYS=${RAYS:-10000000} TOP=2
echo "== base";            TAG=c5_base bash scripts/exp_c5.sh | head -2
echo "== no waves_per_eu"; TAG=c5_noattr TURTLE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/var/lib_noattr.so bash scripts/exp_c5.sh | head -2
echo "== sorted origins";  TAG=c5_sort EXTRA="--sort 1024" bash scripts/exp_c5.sh | head -2
