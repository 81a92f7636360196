#!/bin/bash
# 3 Gbp step against filters above 4 GiB (VERDICT r4 missing #1): the reference tool's own size for 3 Gbp
# (4.64 GB, not a power of two), 8 GiB, 16 GiB; 4 GiB as the base line
cd "$GRAFT_REPO_ROOT" || exit 1
for fb in ${SIZES:-4294967296 4640000000 8589934592 17179869184}; do
python bench.py --filter-bytes $fb --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather ${BENCH_ARGS:-} 2>gpurun_out/bigfilter_$fb.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['roofline']['pipeline']
print('$fb', 'value', d['value'], 'ms', d['ms_per_step'], d['phases_ms'], 'partition', p['partition_ms'], 'probe', p['probe_ms'], 'launches', d['roofline']['launches_per_step'], 'kernels', p['kernels'][0], d['events'])"
tail -2 gpurun_out/bigfilter_$fb.err
done
