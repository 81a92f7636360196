#!/bin/bash
# round 4: parity subset, then the 3 Gbp step / configs[4]-like / SNV / counting with tuning variants given as arguments ("k=v k=v" per variant)
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/tune; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "polish_matches_oracle or screen_bitmap or demo" 2>&1 | grep -v "amdgpu.ids" | tail -5 > $OUT/parity_subset.log; cat $OUT/parity_subset.log
fi
for v in "$@"; do
  t=""; for kv in $v; do t="$t --tune $kv"; done
  echo "== $v"
  NTEDIT_HIP_DEBUG=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather $t ${BENCH_ARGS:-} 2> $OUT/err.log |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['phases_ms'], d.get('events'))"
  grep -E "chunk 1/1|wave-kernel|inside failing" $OUT/err.log | tail -3 | cut -c1-700
done
