cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5final
timeout 2400 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "amdgpu.ids" | tail -8 > gpurun_out/r5final/gpu_tests.log; cat gpurun_out/r5final/gpu_tests.log
timeout 700 python tests/tools/fuzz_parity.py --gpu --cbf --seed 6161 --minutes 8 --iters 0 2>&1 | tail -3 | tee gpurun_out/r5final/fuzz_cbf.log
timeout 700 python tests/tools/fuzz_parity.py --gpu --seed 6262 --minutes 8 --iters 0 2>&1 | tail -3 | tee gpurun_out/r5final/fuzz.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
