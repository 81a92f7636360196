#!/bin/bash
# do the partition and probe kernels of "bin_overlap" run at the same time?  kernel trace with timestamps
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/overlap
mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
NTEDIT_HIP_LIB=$ROOT/ntedit_amd/${1:-libntedit_hip_wpe5.so} rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python $ROOT/bench.py --screen-only --steps 1 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune bin_overlap=1 > /dev/null 2> $OUT/err.log
f=$(find $OUT/t -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_bin_probe" in r["Kernel_Name"] or "k_wc_scatter" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-8:]:
    print("%-14s start %9.3f ms  end %9.3f ms  (%.3f ms)  queue %s" % (r["Kernel_Name"][:14].replace("void nte::", ""), (int(r["Start_Timestamp"]) - t0) / 1e6,
          (int(r["End_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("Queue_Id", "?")))
PY
rm -rf $OUT/t
