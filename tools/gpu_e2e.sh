#!/bin/bash
# CLI-facing GPU tests + the bench line's end-to-end region
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multigpu.py -x -q -m gpu -k "cli or demo or golden or full_size" --timeout 300 2>&1 | tail -2
timeout 500 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather --e2e-bgzf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(json.dumps(d['end_to_end'])); print(json.dumps(d.get('end_to_end_bgzf'))); print(json.dumps(d.get('kernel_region_host'))); print(d['value'])"
