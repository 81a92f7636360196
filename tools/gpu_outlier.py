#!/usr/bin/env python
"""Reproduction of round 4's screening outliers: configs[2] screened in 184 / 553 ms instead of
9.8 ms when its test ran right behind the 3 Gbp every-contig test in the same process.  The sequence of the test
suite -- a 3 Gbp context that is closed, then a NEW context's first call on 250 Mbp -- looped, with the library's
allocation times (NTEDIT_HIP_DEBUG) and per-stage screening times (bin_timing) next to the HIP-event times.
Run on the GPU box: NTEDIT_HIP_DEBUG=1 python tools/gpu_outlier.py [loops]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import ntedit_amd
    from ntedit_amd.synth import SyntheticJob

    loops = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    big = float(os.environ.get("OUTLIER_BIG", "3e9"))
    for it in range(loops):
        pol = ntedit_amd.Polisher(0)
        pol.set_params(ntedit_amd.default_params())
        job = SyntheticJob(pol, big, k=25, hash_num=3, filter_bytes=1 << 32)
        for rep in range(2):
            t0 = time.perf_counter()
            res = pol.polish_batch(None, job.offsets, job.lens, device_ptr=job.device_ptr, n=job.n_bytes)
            st = res.stats()
            res.free()
            print("[loop %d] %.1f Gbp call %d: wall %.1f ms, total %.1f, screen %.1f (partition %.1f + probe %.1f), machine %.1f"
                  % (it, big / 1e9, rep, (time.perf_counter() - t0) * 1e3, st.ms_total, st.ms_screen, st.ms_partition, st.ms_probe,
                     st.ms_machine), flush=True)
        pol.close()
        del job  # (like the tests: no torch.cuda.empty_cache())
        pol = ntedit_amd.Polisher(0)
        pol.set_params(ntedit_amd.default_params())
        pol.set_tuning("bin_timing", 1 if it % 2 == 0 else 0)
        job = SyntheticJob(pol, 250e6, k=25, hash_num=3, filter_bytes=1 << 32, contig_len=100_000, n_runs=False)
        for rep in range(3):
            t0 = time.perf_counter()
            res = pol.polish_batch(None, job.offsets, job.lens, device_ptr=job.device_ptr, n=job.n_bytes)
            st = res.stats()
            res.free()
            print("[loop %d] configs[2] call %d: wall %.1f ms, total %.1f, screen %.1f (partition %.1f + probe %.1f), machine %.1f"
                  % (it, rep, (time.perf_counter() - t0) * 1e3, st.ms_total, st.ms_screen, st.ms_partition, st.ms_probe,
                     st.ms_machine), flush=True)
        pol.close()
        del job


if __name__ == "__main__":
    main()
