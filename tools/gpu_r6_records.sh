#!/bin/bash
# round 6: the records of the shipped build -- GPU test tier, default bench line, kernel-trace stats, fuzz (gpu_final.sh);
# machine probe counts of the profile build; side lines (-s 1, counting, genome-like 3 Gbp / 1 Gbp, the same i.i.d.);
# shard sizes; the 2-rank bench rehearsal; genome-like fuzz; PMC passes + roofline traffic (profile_gpu.sh)
cd "$GRAFT_REPO_ROOT" || exit 1
T=r6final
bash tools/gpu_final.sh $T ${FUZZ_MIN:-5}
bash tools/gpu_machine_probes.sh > gpurun_out/$T/machine_probes.log 2>&1
cp gpurun_out/machine_probes.json gpurun_out/$T/
B="--no-regions --no-cpu-baseline --no-gather"
python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 5 --warmup 2 $B > gpurun_out/$T/bench_snv_250Mbp.json 2>/dev/null
python bench.py --counting --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 $B > gpurun_out/$T/bench_counting_250Mbp.json 2>/dev/null
for S in genome iid; do for N in 3.0e9 1.0e9; do
  NTEDIT_HIP_DEBUG=1 python bench.py --structure $S --bases $N --steps 3 --warmup 1 $B --tune bin_timing=1 > gpurun_out/$T/bench_${S}_$N.json 2> gpurun_out/$T/bench_${S}_$N.err
done; done
NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_prof.so NTEDIT_HIP_DEBUG=1 python bench.py --structure genome --steps 1 --warmup 1 $B 2>&1 >/dev/null | grep -E "machine filter|wave-kernel|inside failing" > gpurun_out/$T/genome_machine_probes.txt
bash tools/gpu_small.sh > gpurun_out/$T/small_shards.txt 2>&1
bash tools/gpu_reh.sh > gpurun_out/$T/rehearsal_n2.log 2>&1; cp gpurun_out/rehearsal_n2.json gpurun_out/$T/ 2>/dev/null
timeout 500 python tests/tools/fuzz_genome_like.py --minutes 5 --seed 6262 > gpurun_out/$T/fuzz_genome_like.log 2>&1
bash tools/profile_gpu.sh r6 > gpurun_out/$T/profile_gpu.log 2>&1
tail -3 gpurun_out/$T/gpu_tests.log; cut -c1-400 gpurun_out/$T/bench.json; cat gpurun_out/$T/small_shards.txt; cat gpurun_out/$T/fuzz.log; tail -1 gpurun_out/$T/fuzz_genome_like.log
