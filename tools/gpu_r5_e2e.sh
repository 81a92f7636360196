#!/bin/bash
# end-to-end after: parallel event index, flattened unit writes, 1 Gbase batches; then the machine tail profile
cd "$GRAFT_REPO_ROOT" || exit 1
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "cli or golden or demo" 2>&1 | grep -v amdgpu.ids | tail -2
NTEDIT_BENCH_KEEP_E2E=1 timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-gather > /tmp/b.json 2> /tmp/b.err
W=$(grep -o "inputs kept in .*" /tmp/b.err | sed 's/inputs kept in //')
python -c "
import json; d=json.load(open('/tmp/b.json')); e=d['end_to_end']
print('bench default: value', e['value'], 'median', e['median_value'], e['stage_s'], e['region_s_all_runs'], 'wall', e['process_wall_s'])"
run() {
  for i in 1 2 3; do
    rm -f $W/x_edited.fa $W/x_changes.tsv $W/x_variants.vcf
    NTEDIT_HIP_DEBUG=1 ./ntedit_amd/ntedit -f $W/draft.fa -r $W/truth.bf -b $W/x --report $1 2>$W/err.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[$1]', round(d['seconds'],4), 'Gb/s', round(d['bases']/d['seconds']/1e9,2), 'read', d['read_s'], 'gpu calls', d['polish_call_s'], 'write', d['write_s'])"
  done
  grep "render:" $W/err.txt
}
run ""
cmp $W/x_edited.fa $W/out_edited.fa && cmp $W/x_changes.tsv $W/out_changes.tsv && echo "outputs identical"
run "--batch-bases 536870912"
run "--batch-bases 2147483648"
run "-t 16"
rm -rf $W
tools/gpu_r5_tail.sh
