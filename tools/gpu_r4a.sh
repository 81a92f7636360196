#!/bin/bash
# round 4, first look at the per-lane runs: a parity subset, then the 3 Gbp step with lanes off / clean / clean+dirty, SNV and counting 250 Mbp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r4a; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "polish_matches_oracle or screen_bitmap or demo" 2>&1 | grep -v "amdgpu.ids" | tail -5 > $OUT/parity_subset.log; cat $OUT/parity_subset.log
for t in "lanes=0" "lanes=1" "lanes=2" "lanes=2 --tune defer_run=1" "lanes=2 --tune defer_run=4" "lanes=2 --tune defer_run=0"; do
  echo "== $t"
  NTEDIT_HIP_DEBUG=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune $t 2> $OUT/err_${t// /_}.log |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['phases_ms'], d.get('events'))"
  grep "chunk 1/1" $OUT/err_${t// /_}.log | tail -1 | cut -c1-400
done
for t in "lanes=0" "lanes=2"; do
  echo "== snv $t"
  timeout 300 python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune $t 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['phases_ms'])"
  echo "== counting $t"
  timeout 300 python bench.py --counting --bases 250e6 --contig-len 100000 --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather --tune $t 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['phases_ms'])"
done
