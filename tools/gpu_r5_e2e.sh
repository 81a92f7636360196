#!/bin/bash
# round 5: end-to-end region of the `ntedit` binary with the page-locked batch buffers + reserve, A/B by hand:
# default, --pack, NTEDIT_NO_PINNED_BATCHES=1; CLI parity tests first
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "cli or reserve or packed" 2>&1 | grep -v amdgpu.ids | tail -3
NTEDIT_BENCH_KEEP_E2E=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather > /tmp/b.json 2> /tmp/b.err
W=$(grep -o "inputs kept in .*" /tmp/b.err | sed 's/inputs kept in //')
python -c "
import json; d=json.load(open('/tmp/b.json')); e=d['end_to_end']
print('bench default: value', e['value'], 'median', e['median_value'], e['stage_s'], e['region_s_all_runs'])
print('kernel_region_host', d['kernel_region_host']['value'], d['kernel_region_host']['ms_per_call'], 'packed', d['kernel_region_host_packed']['value'], d['kernel_region_host_packed']['ms_per_call'], 'pack_s', d['kernel_region_host_packed']['pack_s_host_threads'])"
run() {
  for i in 1 2 3; do
    rm -f $W/x_edited.fa $W/x_changes.tsv $W/x_variants.vcf
    env $1 ./ntedit_amd/ntedit -f $W/draft.fa -r $W/truth.bf -b $W/x --report $2 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[$1 $2]', round(d['seconds'],4), 'Gb/s', round(d['bases']/d['seconds']/1e9,2), 'read', d['read_s'], 'gpu calls', d['polish_call_s'], 'write', d['write_s'], 'gpu_ms', d['gpu_ms'])"
  done
}
run "A=1" ""
cmp $W/x_edited.fa $W/out_edited.fa && echo "outputs identical"
run "A=1" "--pack"
cmp $W/x_edited.fa $W/out_edited.fa && echo "outputs identical (pack)"
run "NTEDIT_NO_PINNED_BATCHES=1" ""
run "A=1" "--batch-bases 1073741824"
# the disk's own sequential write rate (3 GB, page cache as the CLI uses it, then with O_DIRECT)
dd if=/dev/zero of=$W/dd.bin bs=8M count=384 2>&1 | tail -1
dd if=/dev/zero of=$W/dd2.bin bs=8M count=384 oflag=direct 2>&1 | tail -1
df -h $W | tail -1
rm -rf $W
