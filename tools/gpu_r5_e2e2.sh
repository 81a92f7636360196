#!/bin/bash
# writer-stage sweep: batch size x render threads, with the renderer's own wait / write split (NTEDIT_HIP_DEBUG)
cd "$GRAFT_REPO_ROOT" || exit 1
NTEDIT_BENCH_KEEP_E2E=1 timeout 900 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gather --no-e2e > /tmp/b.json 2> /tmp/b.err
W=$(mktemp -d /tmp/e2e.XXXX)
python - "$W" <<'P'
import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np, ntedit_amd
from ntedit_amd.synth import SyntheticJob
W = sys.argv[1]
pol = ntedit_amd.Polisher(0)
pol.set_params(ntedit_amd.default_params())
job = SyntheticJob(pol, 3e9, k=25, hash_num=3, filter_bytes=1 << 32, seed=20251031, draft_seed=20251032)
pol.filter_save_file(os.path.join(W, "truth.bf"))
h = job.batch.cpu().numpy()
with open(os.path.join(W, "draft.fa"), "wb") as f:
    for i, (o, l) in enumerate(zip(job.offsets.tolist(), job.lens.tolist())):
        f.write(b">contig%d len=%d\n" % (i, l)); f.write(h[o:o + l + 1].tobytes())
pol.close()
P
sync
run() {
  for i in 1 2; do
    rm -f $W/x_edited.fa $W/x_changes.tsv $W/x_variants.vcf
    NTEDIT_HIP_DEBUG=1 ./ntedit_amd/ntedit -f $W/draft.fa -r $W/truth.bf -b $W/x --report $1 2>$W/err.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[$1]', round(d['seconds'],4), 'Gb/s', round(d['bases']/d['seconds']/1e9,2), 'read', d['read_s'], 'gpu calls', d['polish_call_s'], 'write', d['write_s'])"
    grep "render:" $W/err.txt | awk '{w+=$(NF-7); e+=$(NF-2); n++} END {print "   render calls", n, "waited", w, "wrote", e}'
  done
}
for t in 4 8 16; do
  run "-t $t"
  run "-t $t --batch-bases 1073741824"
  run "-t $t --batch-bases 2147483648"
done
grep "render:" $W/err.txt | head -3
rm -rf $W
