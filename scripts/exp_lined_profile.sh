#!/bin/bash
# Diagnostic: per-wave time split of the lined pass (needs scratch/prof/libturtle_amd.so,
# built in the build container with: make -C turtle_amd/csrc prof)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for setting in "${@:-X=1}"; do
  echo "== $setting"
  ( export $setting; TURTLE_AMD_LIBRARY=$GRAFT_REPO_ROOT/scratch/prof/libturtle_amd.so timeout -k 10 200 python3 scripts/exp_lined_profile.py )
done
