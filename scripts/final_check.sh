#!/bin/bash
# Round-end rehearsal on the GPU box: the GPU suite, smoke(), the default bench line,
# and the 2-rank path (both ranks on the one GPU, gloo for the tally's all-reduce).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/final; mkdir -p $out
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest exit $?"; tail -2 $out/pytest_gpu.log
timeout -k 10 300 python3 -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $out/smoke.log 2>&1; echo "smoke exit $?"; tail -1 $out/smoke.log
timeout -k 10 600 python3 bench.py > $out/bench.json 2> $out/bench.err; echo "bench exit $?"; cut -c1-330 $out/bench.json
TURTLE_BENCH_BACKEND=gloo timeout -k 10 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu > $out/bench_2rank.json 2> $out/bench_2rank.err
echo "2-rank exit $?"; cut -c1-330 $out/bench_2rank.json
