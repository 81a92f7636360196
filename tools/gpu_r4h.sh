#!/bin/bash
# round 4: packed batches -- CLI tests, a short GPU fuzz (the CLI packs its batches), the default bench line with regions
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_multigpu.py -q -m gpu -x -k "cli or packed or make_genome or shard or driver" 2>&1 | grep -v amdgpu.ids | tail -4
timeout 400 python tests/tools/fuzz_parity.py --gpu --minutes 3 --seed 515151 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-gather ${BENCH_ARGS:-} 2>gpurun_out/bench_err.log | tee gpurun_out/r4_bench_mid.json | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['phases_ms'])
print(json.dumps(d.get('kernel_region_host'))[:400])
print(json.dumps(d.get('kernel_region_host_packed'))[:600])
e=d.get('end_to_end'); print(e['value'], e['median_value'], e['stage_s'], e['region_s_all_runs'])"
