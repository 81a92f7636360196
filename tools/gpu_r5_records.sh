#!/bin/bash
# round 5: the records of the shipped build -- GPU test tier, default bench line, kernel-trace stats, fuzz (gpu_final.sh);
# PMC passes + roofline traffic (profile_gpu.sh); machine probe counts of the profile build; side lines; shard sizes;
# filters above 4 GiB; the outlier sequence of round 4 (tools/gpu_outlier.py)
cd "$GRAFT_REPO_ROOT" || exit 1
T=r5final
bash tools/gpu_final.sh $T ${FUZZ_MIN:-5}
bash tools/gpu_machine_probes.sh > gpurun_out/$T/machine_probes.log 2>&1
cp gpurun_out/machine_probes.json gpurun_out/$T/
python bench.py --snv --bases 250e6 --contig-len 100000 --filter-bytes $((1<<29)) --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather > gpurun_out/$T/bench_snv_250Mbp.json 2>/dev/null
python bench.py --counting --bases 250e6 --contig-len 100000 --steps 5 --warmup 2 --no-regions --no-cpu-baseline --no-gather > gpurun_out/$T/bench_counting_250Mbp.json 2>/dev/null
bash tools/gpu_small.sh > gpurun_out/$T/small_shards.txt 2>&1
SIZES="4294967296 4640000000 8589934592 17179869184" bash tools/gpu_bigfilter.sh > gpurun_out/$T/big_filters.txt 2>&1
NTEDIT_HIP_DEBUG=1 python tools/gpu_outlier.py 2 2>&1 | grep -v "chunk 1/1\|binned chunk\|amdgpu.ids" > gpurun_out/$T/outlier_sequence.log
NTEDIT_HIP_LIB=$PWD/ntedit_amd/libntedit_hip_prof.so NTEDIT_HIP_DEBUG=1 python bench.py --steps 1 --warmup 1 --no-regions --no-cpu-baseline --no-gather 2>&1 >/dev/null | grep -E "wave-kernel|inside failing|machine filter" > gpurun_out/$T/wave_kernel_phases.txt
bash tools/gpu_reh.sh > gpurun_out/$T/rehearsal_n2.log 2>&1; cp gpurun_out/rehearsal_n2.json gpurun_out/$T/ 2>/dev/null
bash tools/profile_gpu.sh r5 > gpurun_out/$T/profile_gpu.log 2>&1
tail -3 gpurun_out/$T/gpu_tests.log; cut -c1-400 gpurun_out/$T/bench.json; cat gpurun_out/$T/small_shards.txt; cat gpurun_out/$T/fuzz.log
