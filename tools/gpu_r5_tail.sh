#!/bin/bash
# the machine's launches at the shard sizes of an 8-GPU run: per-launch times (NTEDIT_HIP_DEBUG) and, from the profile build,
# the wave kernel's events by duration (log2 cycles)
cd "$GRAFT_REPO_ROOT" || exit 1
for b in 3.75e8 3e9; do
echo "== $b shipped build"
NTEDIT_HIP_DEBUG=1 python bench.py --bases $b --steps 2 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>&1 >/dev/null | grep "chunk 1/1" | tail -2
echo "== $b profile build"
NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_prof.so NTEDIT_HIP_DEBUG=1 python bench.py --bases $b --steps 1 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>&1 >/dev/null | grep "chunk 1/1\|wave-kernel\|inside failing" | tail -4
done
