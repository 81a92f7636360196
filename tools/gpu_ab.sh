#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for lib in libntedit_hip.so libntedit_hip_SYNCTHREADS.so libntedit_hip_SEL4.so; do
echo "== $lib"
NTEDIT_HIP_LIB=$PWD/ntedit_amd/$lib timeout 100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binned_screen" 2>&1 | tail -1
NTEDIT_HIP_LIB=$PWD/ntedit_amd/$lib timeout 300 python bench.py --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['pipeline']['partition_ms'], d['roofline']['pipeline']['probe_ms'])"
done
