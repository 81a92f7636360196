#!/bin/bash
# end-to-end region of the `ntedit` binary on bench.py's 3 Gbp draft for several batch-size settings ("" = the default ramp):
# tools/gpu_e2e_batches.sh "" "--batch-bases 1073741824" ...
cd "$GRAFT_REPO_ROOT" || exit 1
NTEDIT_BENCH_KEEP_E2E=1 timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gather > /tmp/b.json 2> /tmp/b.err
W=$(grep -o "inputs kept in .*" /tmp/b.err | sed 's/inputs kept in //')
python -c "import json; d=json.load(open('/tmp/b.json')); print('bench default:', json.dumps(d['end_to_end'])[:330])"
for opt in "$@"; do
  for i in 1 2 3; do
    rm -f $W/x_edited.fa $W/x_changes.tsv $W/x_variants.vcf
    ./ntedit_amd/ntedit -f $W/draft.fa -r $W/truth.bf -b $W/x --report $opt 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[$opt]', round(d['seconds'],4), 'read', d['read_s'], 'gpu calls', d['polish_call_s'], 'write', d['write_s'])"
  done
done
cmp $W/x_edited.fa $W/out_edited.fa && echo "outputs identical"
rm -rf $W
