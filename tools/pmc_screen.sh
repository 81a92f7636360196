#!/bin/bash
# PMC passes over the 3 Gbp screening only (bench.py --screen-only): tools/pmc_screen.sh <tag>
# (counters in their own runs with --kernel-trace only, as the pool requires)
set -u
TAG=${1:-scr}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--screen-only --steps 1 --warmup 1 --no-cpu-baseline --no-gather --no-regions ${BENCH_ARGS:-}"
rocprofv3 -L > $OUT/counters_available.txt 2>&1
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_ATOMIC_sum TCC_WRITE_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" \
           "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_BUSY_sum TCC_TAG_STALL_sum TCC_NORMAL_EVICT_sum TCC_STREAMING_REQ_sum" \
           "TA_BUSY_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --output-format csv --pmc $PMC -d $OUT/pmc$i -o pmc -- python $ROOT/bench.py $ARGS > /dev/null 2> $OUT/pmc$i.err
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" "$OUT/pmc_summary.txt" "$PMC" <<'PY'
import csv, sys, collections
f, out, pmc = sys.argv[1], sys.argv[2], sys.argv[3]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r.get("Kernel_Name", "?")[:60]
    if "nte::" not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k].add(r.get("Dispatch_Id"))
with open(out, "a") as o:
    o.write("## pmc pass: %s\n" % pmc)
    for k in agg:
        o.write("%s dispatches=%d " % (k, len(cnt[k])) + " ".join("%s=%.6g" % kv for kv in sorted(agg[k].items())) + "\n")
PY
  else
    echo "## pmc pass failed: $PMC" >> $OUT/pmc_summary.txt; tail -3 $OUT/pmc$i.err >> $OUT/pmc_summary.txt
  fi
  rm -rf $OUT/pmc$i
done
cat $OUT/pmc_summary.txt
