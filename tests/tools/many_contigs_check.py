"""TEST INFRASTRUCTURE (runs the CPU oracle as the checker; lives under tests/ for that reason).
Fragmented-assembly shape: very many short contigs (default 500,000 x ~600 bp) through the `ntedit`
binary, outputs compared with the oracle's CLI.  usage (GPU box): python tests/tools/many_contigs_check.py [n] [len]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import helpers as H  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 600
    work = "/tmp/ntedit_many"
    os.makedirs(work, exist_ok=True)
    rng = np.random.default_rng(3)
    truth = H.random_genome(rng, 4_000_000)
    H.write_fasta(os.path.join(work, "truth.fa"), [(b"t", truth)])
    H.mkbf([os.path.join(work, "truth.fa")], os.path.join(work, "t.bf"), k=25, hashes=3, nbytes=1 << 25)
    t = np.frombuffer(truth, dtype=np.uint8)
    starts = rng.integers(0, len(truth) - 2 * L, size=n)
    lens = rng.integers(L // 2, L * 3 // 2, size=n)
    with open(os.path.join(work, "draft.fa"), "wb") as f:
        for i in range(n):
            s = t[starts[i]:starts[i] + lens[i]].copy()
            m = rng.random(len(s)) < 2e-3
            s[m] = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(m.sum()))]
            f.write(b">c%d\n" % i)
            f.write(s.tobytes())
            f.write(b"\n")
    H.build_oracle()
    out = {}
    for tag, exe in (("gpu", os.path.join(ROOT, "ntedit_amd", "ntedit")), ("cpu", os.path.join(H.ORACLE_BUILD, "ntedit_oracle"))):
        t0 = time.perf_counter()
        r = subprocess.run([exe, "-f", os.path.join(work, "draft.fa"), "-r", os.path.join(work, "t.bf"), "-b",
                            os.path.join(work, tag), "--report"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-500:]
        if tag == "gpu":
            print([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        out[tag] = time.perf_counter() - t0
    same = all(open(os.path.join(work, "gpu" + s), "rb").read() == open(os.path.join(work, "cpu" + s), "rb").read()
               for s in ("_edited.fa", "_changes.tsv"))
    print("contigs %d, %.0f Mbases: ntedit %.2f s, oracle %.2f s, identical %s" %
          (n, float(lens.sum()) / 1e6, out["gpu"], out["cpu"], same))
    return 0 if same else 1


if __name__ == "__main__":
    sys.exit(main())
