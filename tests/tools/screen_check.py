#!/usr/bin/env python
"""Binned-screening checks, one case per child process with a hard timeout (a kernel that does not terminate must not
take the GPU box with it).  GPU box only:  python tests/tools/screen_check.py [--timeout 90]
Every case compares ntedit_hip_screen (screen_mode 2, the write-combining partition + L2-resident probe) with the
oracle's bitmap.  Test infrastructure: imports the oracle."""
import argparse
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = [
    # name, make_case kwargs, tunings
    ("small_pow2", dict(bfbytes=1 << 17, n=60000), {}),
    ("one_slice_forced_chunks", dict(bfbytes=1 << 17, n=60000), {"bin_chunk": 3 * 16384}),
    ("many_slices", dict(bfbytes=1 << 27, n=150000, flavor="N rep"), {}),
    ("nonpow2", dict(bfbytes=100000007 * 8, n=150000, flavor="N rep"), {"bin_chunk": 5 * 16384}),
    ("h1", dict(bfbytes=1 << 22, n=100000, hashes=1), {}),
    ("h5_k40", dict(bfbytes=1 << 24, n=100000, hashes=5, k=40), {}),
    ("force_xcc", dict(bfbytes=1 << 27, n=120000), {"force_xcc": 4}),
    ("overflow_50", dict(bfbytes=1 << 26, n=200000, flavor="N rep"), {"bin_cap_percent": 50}),
    ("overflow_5", dict(bfbytes=1 << 26, n=200000, flavor="N rep"), {"bin_cap_percent": 5}),
    ("big_3M", dict(bfbytes=1 << 28, n=1000000, contigs=3), {}),
    ("counting_p1", dict(bfbytes=1 << 20, n=40000, flavor="cbf"), {}),
    ("counting_p2", dict(bfbytes=1 << 21, n=40000, flavor="cbf N"), {"min_threshold": 2}),
]


def child(idx):
    import numpy as np
    import helpers as H
    import ntedit_amd
    name, kw, tune = CASES[idx]
    tune = dict(tune)
    tmp = tempfile.mkdtemp(prefix="screen_check_")
    case = H.make_case(tmp, 9100 + idx, **kw)
    bf = H.load_bf(case["bf"])
    blob, offs, lens, names = H.pack_batch(H.read_fasta(case["draft"]))
    min_thr = tune.pop("min_threshold", 1) if isinstance(tune, dict) else 1
    want = H.oracle_screen(blob, bf, min_threshold=min_thr)
    pol = ntedit_amd.Polisher(0)
    pol.set_filter(bf["data"], bf["hash_num"], bf["k"], counting=bool(bf.get("counting")))
    pol.set_params(ntedit_amd.default_params(screen_mode=2, min_threshold=min_thr))
    pol.set_tuning("bin_timing", 1)
    for k_, v in tune.items():
        pol.set_tuning(k_, v)
    for kv in filter(None, os.environ.get("SCREEN_CHECK_TUNE", "").split(",")):  # (extra knobs for every case)
        k_, _, v = kv.partition("=")
        pol.set_tuning(k_, int(v))
    t0 = time.time()
    got = pol.screen(blob)
    dt = time.time() - t0
    pol.close()
    bad = int((got != want).sum())
    print("%s: %d bases, %.3f s, %d differing words of %d, absent %d" % (name, len(blob), dt, bad, len(want), int(np.unpackbits(want.view(np.uint8)).sum())))
    return 0 if bad == 0 else 3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--timeout", type=int, default=90)
    ap.add_argument("--child", type=int, default=-1)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    if args.child >= 0:
        sys.exit(child(args.child))
    import helpers as H
    H.oracle_lib()  # (build once, not in every child)
    fails = 0
    for i, (name, _, _) in enumerate(CASES):
        if args.only and args.only not in name:
            continue
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(i)], timeout=args.timeout,
                               capture_output=True, text=True)
            out = (r.stdout + r.stderr[-1500:]).strip()
            print(out if r.returncode == 0 else "FAIL %s rc=%d\n%s" % (name, r.returncode, out), flush=True)
            fails += r.returncode != 0
        except subprocess.TimeoutExpired:
            print("TIMEOUT %s after %d s" % (name, args.timeout), flush=True)
            fails += 1
            break  # the GPU may be wedged: stop here
    print("screen_check: %d failing" % fails)
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
