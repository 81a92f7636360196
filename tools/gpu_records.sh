#!/bin/bash
# round 4: the records of the shipped build -- GPU test tier, default bench line, kernel-trace stats, fuzz (gpu_final.sh);
# PMC passes + roofline traffic (profile_gpu.sh); machine probe counts of the profile build; side lines; shard sizes
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_final.sh r4final 4
bash tools/gpu_machine_probes.sh > gpurun_out/r4final/machine_probes.log 2>&1
bash tools/gpu_machine_probes.sh --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) >> gpurun_out/r4final/machine_probes.log 2>&1
bash tools/gpu_machine_probes.sh --counting --bases 250e6 --contig-len 100000 >> gpurun_out/r4final/machine_probes.log 2>&1
cp gpurun_out/machine_probes.json gpurun_out/r4final/
python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather > gpurun_out/r4final/bench_snv_250Mbp.json 2>/dev/null
python bench.py --counting --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather > gpurun_out/r4final/bench_counting_250Mbp.json 2>/dev/null
bash tools/gpu_small.sh > gpurun_out/r4final/small_shards.txt 2>&1
NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_prof.so NTEDIT_HIP_DEBUG=1 python bench.py --steps 1 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>&1 >/dev/null | grep -E "wave-kernel|inside failing|machine filter" > gpurun_out/r4final/wave_kernel_phases.txt
bash tools/profile_gpu.sh r4 > gpurun_out/r4final/profile_gpu.log 2>&1
tail -3 gpurun_out/r4final/gpu_tests.log; cut -c1-400 gpurun_out/r4final/bench.json; cat gpurun_out/r4final/small_shards.txt; cat gpurun_out/r4final/fuzz.log
