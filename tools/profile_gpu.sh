#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes for the bench workload.
# usage: tools/profile_gpu.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="${@:---bases 3e9 --steps 2 --warmup 1 --no-cpu-baseline --no-gather --no-regions --no-reserve}"  # (--no-reserve: its warm-up batch would count as a fourth, tiny launch of every kernel)
# pass 1: kernel trace + stats (no counters)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py $ARGS > $OUT/trace_bench.json 2> $OUT/trace.err
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# pass 2..n: PMC counters, each in its own run (kernel-trace only, as the pool requires)
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
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
rm -rf $OUT/trace
# per-launch HBM traffic of our kernels (fabric read requests x 64 B + WRITE_SIZE KiB x 1024), for bench.py's roofline.traffic
python - "$OUT/pmc_summary.txt" "$OUT/roofline_traffic.json" "$TAG" "$OUT/trace_bench.json" <<'PY'
import json, re, sys
txt, out, tag = open(sys.argv[1]).read(), sys.argv[2], sys.argv[3]
line = json.loads(open(sys.argv[4]).readline())  # (the bench line of the traced run: workload and build of the kernels)
agg = {}
for row in txt.splitlines():
    m = re.match(r"(.*?) dispatches=(\d+) (.*)", row)
    if not m or "nte::" not in m.group(1):
        continue
    name = re.sub(r"^void ", "", m.group(1)).split("(")[0].replace("nte::", "")
    d = agg.setdefault(name, {"dispatches": int(m.group(2))})
    for kv in m.group(3).split():
        k, v = kv.split("=")
        d[k] = float(v)
res = {}
for name, d in agg.items():
    n = d["dispatches"]
    if "TCC_EA0_RDREQ_sum" in d and "WRITE_SIZE" in d:
        res[name] = {"launches": n, "fetch_bytes_per_launch": d["TCC_EA0_RDREQ_sum"] * 64 / n,
                     "fetch_size_kib_per_launch": d.get("FETCH_SIZE", 0) / n,
                     "write_bytes_per_launch": d["WRITE_SIZE"] * 1024 / n,
                     "tcc_hit_rate": d["TCC_HIT_sum"] / max(1.0, d["TCC_HIT_sum"] + d["TCC_MISS_sum"]) if "TCC_HIT_sum" in d else None}
json.dump({"source": "profiles/%s_bench_3Gbp_pmc_summary.txt (rocprofv3 --pmc, one counter group per run; reads = TCC_EA0_RDREQ x 64 B, writes = WRITE_SIZE KiB x 1024)" % tag,
           "workload_bytes": line["config"]["workload_bytes"], "build_id": line.get("build_id"),
           "kernels": res}, open(out, "w"), indent=1)
PY
ls -la $OUT
head -30 $OUT/kernel_stats.csv
cat $OUT/pmc_summary.txt
