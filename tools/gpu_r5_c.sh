#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu -s -k "nonpow2 or reserve" > gpurun_out/r5c_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5c_tests.log
B="--steps 4 --warmup 2 --no-regions --no-cpu-baseline --no-gather"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline']['pipeline']
print('$1', 'value', d['value'], 'ms', d['ms_per_step'], d['phases_ms'], 'partition', p['partition_ms'], 'probe', p['probe_ms'])"; }
{
for i in 1 2; do
python bench.py $B 2>/dev/null | line "4GiB sweep"
python bench.py $B --tune probe_sweep=0 2>/dev/null | line "4GiB nosweep"
done
python bench.py $B --filter-bytes 17179869184 2>/dev/null | line "16GiB sweep"
python bench.py $B --filter-bytes 17179869184 --tune probe_sweep=0 2>/dev/null | line "16GiB nosweep"
python bench.py $B --filter-bytes 4640000000 2>/dev/null | line "4.64GB sweep"
python bench.py $B --filter-bytes 4640000000 --tune probe_sweep=0 2>/dev/null | line "4.64GB nosweep"
} > gpurun_out/r5c_bench.txt 2>&1
