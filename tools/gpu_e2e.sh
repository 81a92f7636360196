#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multigpu.py -x -q -m gpu -k "cli or demo or golden" --timeout 120 2>&1 | tail -2
timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['end_to_end'])); print(d['value'])"
