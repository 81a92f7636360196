#!/bin/bash
# round 4: SNV / counting 250 Mbp side lines with tuning variants ("k=v k=v" per variant), after a parity subset
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/sidelines; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "polish_matches_oracle or screen_bitmap or demo" 2>&1 | grep -v "amdgpu.ids" | tail -5 > $OUT/parity_subset.log; cat $OUT/parity_subset.log
fi
for v in "$@"; do
  t=""; for kv in $v; do t="$t --tune $kv"; done
  echo "== snv $v"
  NTEDIT_HIP_DEBUG=1 timeout 300 python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather $t 2>$OUT/err_snv.log |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['phases_ms'], d.get('events'))"
  grep -E "chunk 1/1" $OUT/err_snv.log | tail -1 | cut -c1-400
  echo "== counting $v"
  NTEDIT_HIP_DEBUG=1 timeout 300 python bench.py --counting --bases 250e6 --contig-len 100000 --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather $t 2>$OUT/err_cbf.log |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['phases_ms'], d.get('events'))"
  grep -E "chunk 1/1" $OUT/err_cbf.log | tail -1 | cut -c1-400
done
