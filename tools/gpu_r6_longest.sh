#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6_longest
for C in "genome 0 3e9"; do
  set -- $C
  NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_prof.so NTEDIT_HIP_REGIONS=1 NTEDIT_HIP_DEBUG=1 timeout 900 python tools/gpu_longest_events.py $1 $2 $3 > gpurun_out/r6_longest/$1_$3.txt 2>gpurun_out/r6_longest/$1_$3.err
  grep -A30 "top regions" gpurun_out/r6_longest/$1_$3.txt
done
