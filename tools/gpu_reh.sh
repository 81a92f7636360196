#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
NTEDIT_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29555 bench.py --gpus 2 --steps 2 --warmup 1 --bases 1.2e9 --filter-bytes 2147483648 > /tmp/reh.out 2>/tmp/reh.err
mkdir -p gpurun_out; cp /tmp/reh.out gpurun_out/rehearsal_n2.json
cut -c1-1500 /tmp/reh.out
grep -v "^\[W\|amdgpu.ids\|Gloo\|^$" /tmp/reh.err | head -30 | cut -c1-300
