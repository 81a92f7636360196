#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu -k "binned or screen or reserve" > gpurun_out/r5b_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r5b_tests.log
SIZES="4640000000 8589934592 17179869184 34359738368" tools/gpu_bigfilter.sh > gpurun_out/r5b_bigfilter.txt 2>&1
for pl in 0 1 2; do
BENCH_ARGS="--tune probe_parts_log2=$pl" SIZES="8589934592" tools/gpu_bigfilter.sh >> gpurun_out/r5b_bigfilter.txt 2>&1
done
