#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for s in "$@"; do
  echo "== $s"
  ( export $s; timeout -k 10 300 python3 scripts/exp_order.py 2>&1 | tee -a gpurun_out/order.log )
done
