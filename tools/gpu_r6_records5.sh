#!/bin/bash
# round 6: inline_tries 8 (host-side default; the kernels' build id stays): GPU tier, bench line, machine probe counts, side lines
cd "$GRAFT_REPO_ROOT" || exit 1
T=r6final5
mkdir -p gpurun_out/$T
timeout 2400 python -m pytest tests -q -m gpu -x -s 2>&1 | grep -v "amdgpu.ids" > gpurun_out/$T/gpu_tests.log; tail -2 gpurun_out/$T/gpu_tests.log
timeout 900 python bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err; cut -c1-200 gpurun_out/$T/bench.json
bash tools/gpu_machine_probes.sh > gpurun_out/$T/machine_probes.log 2>&1
cp gpurun_out/machine_probes.json gpurun_out/$T/
B="--no-regions --no-cpu-baseline --no-gather"
python bench.py --counting --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 $B > gpurun_out/$T/bench_counting_250Mbp.json 2>/dev/null
for N in 3.0e9 1.0e9; do
  python bench.py --structure genome --bases $N --steps 3 --warmup 1 $B > gpurun_out/$T/bench_genome_$N.json 2> /dev/null
done
NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_prof.so NTEDIT_HIP_DEBUG=1 python bench.py --structure genome --steps 1 --warmup 1 $B 2>&1 >/dev/null | grep -E "machine filter" | tail -1 > gpurun_out/$T/genome_machine_probes.txt
timeout 400 python tests/tools/fuzz_parity.py --gpu --minutes 5 --seed 888 2>&1 | tail -1 > gpurun_out/$T/fuzz.log
for f in bench_counting_250Mbp bench_genome_3.0e9 bench_genome_1.0e9; do python -c "
import json; j=json.load(open('gpurun_out/$T/$f.json')); print('$f', j['ms_per_step'], j['value'], j['phases_ms'])"; done
cat gpurun_out/$T/genome_machine_probes.txt gpurun_out/$T/fuzz.log
