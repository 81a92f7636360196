#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_q
mkdir -p $O
for V in mt5 mt6; do
  for S in iid genome; do
  NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_$V.so NTEDIT_HIP_DEBUG=1 python bench.py --structure $S --steps 3 --warmup 1 --no-regions --no-cpu-baseline --no-gather > $O/b_${V}_$S.json 2> $O/b_${V}_$S.err
  python -c "
import json; j=json.load(open('$O/b_${V}_$S.json')); print('$V $S', j['ms_per_step'], j['phases_ms'])"
  grep -E "events [0-9]+ \(round" $O/b_${V}_$S.err | tail -1 | cut -c60-300
  done
done
