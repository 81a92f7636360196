#!/bin/bash
# Event rounds in pieces (sweeps of piece i next to pass 1 of piece i + 1): parity of the small cases with the pieces forced,
# then bench.py's 3 Gbp step per setting of --tune machine_pieces.   tools/gpu_pieces.sh [pieces ...]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "test_polish_matches_oracle" 2>&1 | tail -3
timeout 400 python tests/tools/fuzz_parity.py --gpu --minutes 3 --seed 9091 2>&1 | tail -3
for p in ${@:-1 2 3 4}; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune machine_pieces=$p 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('pieces=$p', d['ms_per_step'], d['phases_ms'])"
done
