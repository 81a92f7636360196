"""Pins the oracle's control logic against the only fixture the reference's own
test (demo/runme.sh:8-10) holds: demo/ecoli_ntedit_k25_changes.tsv.

The reads that built the reference's Bloom filter are not available, so a proxy
filter is built from the genome reconstructed from draft + changes.tsv
(tests/golden/recon_demo.py).  With it the oracle reproduces 4,986 of the
4,997 expected rows byte-for-byte and the complete expected edited genome; the
other 11 rows are accounted for one by one (helpers.check_demo_rows): same
position and edit with another support / alternate column (7 rows), or another
edit of the same locus that yields the same genome (2 + 2 rows -> 1 + 1)."""
import os
import subprocess
import sys

import helpers as H

DEMO = os.path.join(H.GOLDEN, "demo")
DRAFT = os.path.join(DEMO, "ecoliWithMismatches001Indels0001.fa.gz")
REF_TSV = os.path.join(DEMO, "ecoli_ntedit_k25_changes.tsv")


def test_oracle_reproduces_reference_demo(tmp_path, oracle_build):
    truth = str(tmp_path / "truth.fa")
    out = subprocess.run([sys.executable, os.path.join(H.GOLDEN, "recon_demo.py"), DRAFT, REF_TSV, truth],
                         check=True, capture_output=True, text=True).stdout
    assert "4997 rows, 0 convention mismatches" in out
    H.mkbf([truth], str(tmp_path / "p.bf"), k=25, hashes=3, nbytes=1 << 28)
    hp = H.default_params(max_insertions=4, max_deletions=5)  # demo/runme.sh: -d 5 -i 4
    H.run_oracle(DRAFT, str(tmp_path / "p.bf"), hp, str(tmp_path / "o"))
    ref = open(REF_TSV).read().splitlines()
    got = open(str(tmp_path / "o_changes.tsv")).read().splitlines()
    H.check_demo_rows(ref, got)
    # the edited genome is reproduced completely
    want_seq = open(truth).read().split("\n")[1]
    got_fa = open(str(tmp_path / "o_edited.fa")).read().split("\n")
    assert got_fa[0] == ">U00096.3_MG1655_k12"
    assert got_fa[1] == want_seq
    assert len(got_fa) == 3 and got_fa[2] == ""  # exactly two lines per record
