#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_fuzz
mkdir -p $O
timeout 1500 python tests/tools/fuzz_parity.py --gpu --minutes 20 --seed 70707 2>&1 | tail -4 > $O/fuzz.log; cat $O/fuzz.log
timeout 1400 python tests/tools/fuzz_genome_like.py --minutes 20 --seed 7373 > $O/fuzz_genome_like.log 2>&1; tail -2 $O/fuzz_genome_like.log
