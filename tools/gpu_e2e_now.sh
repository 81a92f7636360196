#!/bin/bash
# CLI / renderer parity tests, then the default bench line's regions (end to end included)
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multigpu.py -q -m gpu -x -k "cli or config2 or shard or driver or golden or demo" 2>&1 | grep -v amdgpu.ids | tail -3
python bench.py --no-cpu-baseline --no-gather 2>gpurun_out/bench_err.log | tee gpurun_out/bench_e2e_now.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'])
e=d.get('end_to_end'); print(e['value'], e['median_value'], e['stage_s'], e['region_s_all_runs'])"
