#!/bin/bash
# round 4: CLI / full-size parity with the mapped writer, then the default bench line (regions included)
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "cli or full_size_every or config2" 2>&1 | grep -v amdgpu.ids | tail -6
python bench.py --no-cpu-baseline --no-gather ${BENCH_ARGS:-} 2>gpurun_out/bench_err.log | tee gpurun_out/r4_bench_mid.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['phases_ms'])
print(json.dumps(d.get('kernel_region_host'))[:700])
print(json.dumps(d.get('end_to_end'))[:2500])"
