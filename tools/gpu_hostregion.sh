#!/bin/bash
# host buffer in -> edit records out (bench.py's kernel_region_host) for a list of `--tune` settings, on the GPU box:
# tools/gpu_hostregion.sh "h2d_chunks=0" "h2d_chunks=8" ...
cd "$GRAFT_REPO_ROOT" || exit 1
for t in "$@"; do
  timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-e2e --tune $t 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$t', d['ms_per_step'], json.dumps(d.get('kernel_region_host'))[:260])"
done
