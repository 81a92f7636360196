"""N>1 path on CPU: world_size-2 gloo run of the filter broadcast + contig
sharding + host-side gather; the merged output must equal the single-process
oracle output byte for byte (input order = reference at -t 1)."""
import filecmp
import os
import subprocess
import sys

import numpy as np

import helpers as H
from ntedit_amd import dist as ndist


def test_shard_contigs_balances_by_bases():
    lens = [50, 1000, 10, 400, 600, 999, 5, 120]
    parts = ndist.shard_contigs(lens, 3, min_len=10)
    flat = sorted(int(i) for p in parts for i in p)
    assert flat == [0, 1, 2, 3, 4, 5, 7]  # contig 6 is below -z and dropped
    loads = [sum(lens[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 400
    for p in parts:
        assert list(p) == sorted(p)
    assert [list(map(int, p)) for p in ndist.shard_contigs(lens, 3, 10)] == [list(map(int, p)) for p in parts]
    # one rank: everything, in order
    assert list(ndist.shard_contigs(lens, 1, 0)[0]) == list(range(8))


def test_two_rank_gloo_run_matches_oracle(tmp_path, oracle_build):
    case = H.make_case(str(tmp_path), 909, contigs=5, n=30000, flavor="N")
    hp = H.default_params()
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"))
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(H.ROOT, "tests", "dist_worker.py"), case["draft"], case["bf"], str(tmp_path / "d")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / "d_edited.fa"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / "d_changes.tsv"), shallow=False)
    assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "d_variants.vcf"))
    assert not [f for f in os.listdir(str(tmp_path)) if ".shard" in f]
