#!/bin/bash
# round 6: which class of the genome-like structure costs the event machine what (1 Gbp, one class at a time)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_classes
mkdir -p $O
Z='"simple":0,"sat":0,"dispersed":0,"segdup":0,"novel":0'
for C in simple sat dispersed segdup novel; do
  case $C in simple) V=0.03;; sat) V=0.03;; dispersed) V=0.05;; segdup) V=0.02;; novel) V=0.005;; esac
  F="{$Z,\"$C\":$V}"
  NTEDIT_HIP_DEBUG=1 timeout 600 python bench.py --structure genome --structure-fractions "$F" --bases 1.0e9 --steps 2 --warmup 1 --no-regions --no-cpu-baseline --no-gather \
     > $O/bench_$C.json 2> $O/bench_$C.err
  echo "== only $C"; python - <<PY
import json
j=json.load(open("$O/bench_$C.json"))
print(j["ms_per_step"], j["phases_ms"], j["events"])
PY
  grep -E "events [0-9]+ \(round" $O/bench_$C.err | tail -1 | cut -c1-300
done
