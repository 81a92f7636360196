#!/bin/bash
# round 6: wave-kernel phase / duration profile (profile build) per structure class, 1 Gbp
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_classes_prof
mkdir -p $O
Z='"simple":0,"sat":0,"dispersed":0,"segdup":0,"novel":0'
for C in iid simple sat dispersed; do
  case $C in simple) V=0.03;; sat) V=0.03;; dispersed) V=0.05;; iid) V=0;; esac
  if [ $C = iid ]; then F="{$Z}"; else F="{$Z,\"$C\":$V}"; fi
  NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_prof.so NTEDIT_HIP_DEBUG=1 timeout 600 python bench.py --structure genome --structure-fractions "$F" --bases 1.0e9 --steps 1 --warmup 1 --no-regions --no-cpu-baseline --no-gather \
     2> $O/bench_$C.err >/dev/null
  echo "== $C"; grep -E "events [0-9]+ \(round|wave-kernel|inside failing|machine filter|parked" $O/bench_$C.err | tail -6 | cut -c1-700
done
