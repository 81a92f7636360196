#!/bin/bash
# round-2 GPU check A: new tests, bench at N=1 with the three regions, and a 2-rank rehearsal of the N>1 bench path
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
python -m pytest tests/test_gpu_multigpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r2a_tests.log
python bench.py --steps 3 --warmup 1 > gpurun_out/r2a_bench1.json 2> gpurun_out/r2a_bench1.err
NTEDIT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29555 bench.py --gpus 2 --steps 2 --warmup 1 --bases 8e8 --filter-bytes 1073741824 \
  > gpurun_out/r2a_bench2_rehearsal.json 2> gpurun_out/r2a_bench2_rehearsal.err
tail -5 gpurun_out/r2a_tests.log; cat gpurun_out/r2a_bench1.json; tail -3 gpurun_out/r2a_bench1.err; cat gpurun_out/r2a_bench2_rehearsal.json; tail -5 gpurun_out/r2a_bench2_rehearsal.err
