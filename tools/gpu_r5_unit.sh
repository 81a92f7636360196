#!/bin/bash
# end to end with render units / contig parts of 1 (shipped), 2, 4, 8 MiB (LD_PRELOAD of library variants)
cd "$GRAFT_REPO_ROOT" || exit 1
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
for lib in "" libntedit_hip_u2.so libntedit_hip_u4.so libntedit_hip_u8.so ""; do
  for i in 1 2 3; do
    rm -f $W/x_edited.fa $W/x_changes.tsv $W/x_variants.vcf
    LD_PRELOAD=${lib:+$PWD/ntedit_amd/$lib} NTEDIT_HIP_DEBUG=1 ./ntedit_amd/ntedit -f $W/draft.fa -r $W/truth.bf -b $W/x --report 2>$W/err.txt | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('[${lib:-shipped}]', round(d['seconds'],4), 'Gb/s', round(d['bases']/d['seconds']/1e9,2), 'read', d['read_s'], 'gpu calls', d['polish_call_s'], 'write', d['write_s'])"
  done
  grep "render:" $W/err.txt | tail -2
  md5sum $W/x_edited.fa $W/x_changes.tsv | cut -c1-32 | tr '\n' ' '; echo
done
rm -rf $W
