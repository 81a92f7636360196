#!/bin/bash
# SQ / LDS counters of the screening kernels: tools/pmc_sq.sh <tag> [bench args...]
set -u
TAG=${1:-p}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--screen-only --steps 1 --warmup 1 --no-cpu-baseline --no-gather --no-regions $*"
i=0
for PMC in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
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
  else
    echo "## failed: $PMC" >> $OUT/pmc_summary.txt; tail -2 $OUT/pmc$i.err >> $OUT/pmc_summary.txt
  fi
  rm -rf $OUT/pmc$i
done
cat $OUT/pmc_summary.txt
