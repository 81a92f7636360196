#!/bin/bash
# L2 / fabric counters of the screening kernels: tools/pmc_probe.sh <tag> [bench args...]   (two PMC passes)
set -u
TAG=${1:-p}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--screen-only --steps 1 --warmup 1 --no-cpu-baseline --no-gather --no-regions $*"
i=0
for PMC in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCC_BUBBLE_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d $OUT/pmc$i -o pmc -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc$i.err
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$OUT/pmc_summary.txt" "$PMC" <<'PY'
import csv, sys, collections
f, out, pmc = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "?")[:40]
    if "k_bin_probe" not in k and "k_wc_scatter" not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r.get("Dispatch_Id"))
with open(out, "a") as o:
    o.write("## pmc pass: %s\n" % pmc)
    for k in agg:
        o.write("%s dispatches=%d " % (k, len(cnt[k])) + " ".join("%s=%.6g" % kv for kv in sorted(agg[k].items())) + "\n")
PY
  fi
  rm -rf $OUT/pmc$i
done
cat $OUT/pmc_summary.txt
