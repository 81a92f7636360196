"""N>1 path on CPU: world_size-2 gloo runs of the filter broadcast + partition by bases (contigs cut into
segments at event-free boundaries) + host-side gather by index; the merged output must equal the
single-process oracle output byte for byte (input order = reference at -t 1)."""
import filecmp
import os
import subprocess
import sys

import numpy as np
import pytest

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


def _bf_case(tmp_path, seed, **kw):
    case = H.make_case(str(tmp_path), seed, **kw)
    return case, H.load_bf(case["bf"]), H.read_fasta(case["draft"])


def test_refine_cut_lands_in_a_clean_stretch(tmp_path, oracle_build):
    """a refined cut has lead k-mers in the filter in front of it and a halo of them behind it; it is a function
    of the draft and the filter only"""
    case, bf, recs = _bf_case(tmp_path, 515, contigs=1, n=40000, p_sub=5e-3, flavor="N")
    k = bf["k"]
    seq = recs[0][1]
    halo = ndist.halo_bases(k, 5, 5)
    lead = ndist.lead_bases(k)
    screen = lambda b: H.oracle_screen(b, bf)  # noqa: E731
    whole = np.unpackbits(screen(seq).view(np.uint8), bitorder="little").astype(bool)
    n_cut = 0
    for nominal in range(1000, len(seq) - 3000, 1777):
        c = ndist.refine_cut(seq, nominal, k, halo, screen, window=2048)
        if c is None:
            continue
        n_cut += 1
        assert nominal + lead <= c < nominal + 2048
        lo, hi = c - lead, c + halo - k
        assert not whole[lo:hi + 1].any()
        assert set(seq[lo:hi + k]) <= set(b"ACGTacgt")
        assert c == ndist.refine_cut(seq, nominal, k, halo, screen, window=2048)
    assert n_cut > 10
    # no clean stretch inside the window -> no cut; window running off the contig -> no cut
    assert ndist.refine_cut(seq, 1000, k, halo, lambda b: np.full((len(b) + 63) // 64, ~np.uint64(0)), window=2048) is None
    assert ndist.refine_cut(seq, len(seq) - 100, k, halo, screen) is None


def test_plan_pieces_cuts_large_contigs_and_balances(tmp_path, oracle_build):
    case, bf, recs = _bf_case(tmp_path, 516, contigs=3, n=50000)
    k = bf["k"]
    halo = ndist.halo_bases(k, 5, 5)
    screen = lambda b: H.oracle_screen(b, bf)  # noqa: E731
    pieces = ndist.plan_pieces(recs, 4, 100, k, halo, screen, seg_bases=9000)
    # the 40-base record is below -z; every other contig is covered exactly once, in order
    by = {}
    for p in pieces:
        by.setdefault(p.contig, []).append(p)
    assert sorted(by) == [0, 1, 2]
    for ci, ps in by.items():
        assert ps[0].start == 0 and ps[-1].end == len(recs[ci][1]) and len(ps) >= 4
        assert all(a.end == b.start for a, b in zip(ps, ps[1:]))
        assert [p.seg for p in ps] == list(range(len(ps))) and all(p.n_seg == len(ps) for p in ps)
    load = [sum(p.end - p.start for p in pieces if p.owner == r) for r in range(4)]
    assert max(load) - min(load) <= max(p.end - p.start for p in pieces)  # LPT: within one piece
    # deterministic
    again = ndist.plan_pieces(recs, 4, 100, k, halo, screen, seg_bases=9000)
    assert [(p.contig, p.start, p.end, p.owner) for p in again] == [(p.contig, p.start, p.end, p.owner) for p in pieces]
    # one rank: no cuts
    assert all(p.n_seg == 1 for p in ndist.plan_pieces(recs, 1, 100, k, halo, screen))


def _two_ranks(tmp_path, case, extra, world=2, lazy=False, gather=None):
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env.pop("DIST_LAZY", None)
    env.pop("DIST_GATHER", None)
    if lazy:
        env["DIST_LAZY"] = "1"
    if gather:
        env["DIST_GATHER"] = gather
    port = 29600 + (os.getpid() % 300)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(H.ROOT, "tests", "dist_worker.py"), case["draft"], case["bf"], str(tmp_path / "d")] + extra
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / "d_edited.fa"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / "d_changes.tsv"), shallow=False)
    body = [l for l in open(str(tmp_path / "o_variants.vcf")).read().splitlines() if not l.startswith("#")]
    assert open(str(tmp_path / "d_variants.vcf")).read().splitlines() == body
    assert not [f for f in os.listdir(str(tmp_path)) if ".shard" in f]
    return [[int(x) for x in l.split()] for l in open(str(tmp_path / "d.stats")).read().splitlines()]


def test_two_rank_gloo_run_matches_oracle(tmp_path, oracle_build):
    case = H.make_case(str(tmp_path), 909, contigs=5, n=30000, flavor="N")
    H.run_oracle(case["draft"], case["bf"], H.default_params(), str(tmp_path / "o"))
    stats = _two_ranks(tmp_path, case, [])
    assert sum(s[0] for s in stats) == 5 and all(s[1] == 0 for s in stats)  # whole contigs only


def test_two_rank_split_contigs_reassemble_byte_identically(tmp_path, oracle_build):
    """contigs far larger than a rank's share are cut into segments at event-free boundaries, the segments are
    polished on different ranks, and the gather re-assembles every contig byte-identically"""
    case = H.make_case(str(tmp_path), 910, contigs=2, n=90000, flavor="N lower", p_sub=4e-3, p_ins=6e-4, p_del=6e-4,
                       bfbytes=1 << 21)
    H.run_oracle(case["draft"], case["bf"], H.default_params(), str(tmp_path / "o"))
    stats = _two_ranks(tmp_path, case, ["7000"])
    assert all(s[1] >= 5 for s in stats)  # both ranks polished segments
    loads = [s[3] for s in stats]
    assert abs(loads[0] - loads[1]) < 0.1 * sum(loads)
    # refined cuts verify (a cut can still fail where an edit chain runs through the clean stretch -- the small,
    # false-positive-rich filters of the other tests provoke that -- and is then re-run like a blind one)
    assert sum(s[2] for s in stats) == 0


def test_two_rank_bad_cuts_are_caught_and_rerun(tmp_path, oracle_build):
    """cuts placed blindly (the planner is told every k-mer is in the filter) in an error-dense draft: the
    library's verification rejects the segments whose cut is not event-free, they are re-run joined with their
    successors, and the result is still byte-identical"""
    case = H.make_case(str(tmp_path), 911, contigs=2, n=60000, p_sub=2e-2, p_ins=3e-3, p_del=3e-3)
    H.run_oracle(case["draft"], case["bf"], H.default_params(), str(tmp_path / "o"))
    stats = _two_ranks(tmp_path, case, ["3000", "blind"])
    assert sum(s[2] for s in stats) > 0  # some cuts were rejected


@pytest.mark.parametrize("world", [2, 4])
def test_ranks_read_their_own_pieces_and_gather_in_parallel(tmp_path, oracle_build, world):
    """round 6 (VERDICT r5 "next" 5): every rank opens the draft's INDEX (ntedit_hip_fasta_open), plans the same partition from
    the lengths, reads only its own pieces and the windows of its cuts (ntedit_hip_fasta_read), and copies its rendered
    pieces to their offsets in the final files itself (dist.gather_parallel: the per-piece byte counts are all-gathered,
    nobody reads another rank's bytes) -- world 2 and 4 over gloo, contigs cut into segments, byte-identical to the oracle"""
    case = H.make_case(str(tmp_path), 912 + world, contigs=3, n=80000, flavor="N lower", p_sub=4e-3, p_ins=6e-4, p_del=6e-4,
                       bfbytes=1 << 21)
    H.run_oracle(case["draft"], case["bf"], H.default_params(), str(tmp_path / "o"))
    stats = _two_ranks(tmp_path, case, ["7000"], world=world, lazy=True, gather="parallel")
    assert len(stats) == world and all(s[1] >= 2 for s in stats)
    total = None
    read = []
    for r in range(world):
        got, total = [int(x) for x in open(str(tmp_path / ("d.read%d" % r))).read().split()]
        read.append(got)
    # a rank reads its share (+ halos and cut windows: 64 KB per cut looked for, which is most of it at this size),
    # never the draft
    share = [s[3] for s in stats]
    n_cuts = sum(s[0] for s in stats)
    assert all(read[r] < share[r] + n_cuts * (1 << 16) + 4096 for r in range(world)), (read, share)
    assert sum(share) <= total


def test_parallel_gather_equals_the_merge(tmp_path, oracle_build):
    """the two gathers write the same bytes (eager draft, whole contigs and bad cuts that are re-run joined)"""
    case = H.make_case(str(tmp_path), 911, contigs=2, n=60000, p_sub=2e-2, p_ins=3e-3, p_del=3e-3)
    H.run_oracle(case["draft"], case["bf"], H.default_params(), str(tmp_path / "o"))
    stats = _two_ranks(tmp_path, case, ["3000", "blind"], gather="parallel")
    assert sum(s[2] for s in stats) > 0  # (joined re-runs went through the parallel gather)


def test_draft_index_reads_ranges_like_the_loader(tmp_path):
    """ntedit_hip_fasta_open + _read (mapped file, index only) against ntedit_hip_fasta_load (everything in memory): same
    records, any range of any record, also through the fallback for inputs the mapped reader refuses (gzip)"""
    import gzip
    from ntedit_amd.run import Draft, read_fasta_fast
    rng = np.random.default_rng(5)
    recs = [(b"r%d some text" % i, H.random_genome(rng, int(n))) for i, n in enumerate([5, 1000, 70, 12345, 1, 257])]
    plain = str(tmp_path / "d.fa")
    H.write_fasta(plain, recs, width=61)
    gz = str(tmp_path / "d.fa.gz")
    with open(plain, "rb") as f, gzip.open(gz, "wb") as g:
        g.write(f.read())
    for path in (plain, gz):
        want = read_fasta_fast(path)
        d = Draft(path)
        try:
            assert [h for h, _ in want] == d.headers and [len(s) for _, s in want] == d.lens
            for i, (_, s) in enumerate(want):
                seq = d.records()[i][1]
                assert seq[:] == s
                for _ in range(20):
                    a = int(rng.integers(0, len(s) + 1))
                    b = int(rng.integers(a, len(s) + 1))
                    assert seq[a:b] == s[a:b]
                    assert bytes(seq.lazy(a, b)) == s[a:b] and len(seq.lazy(a, b)) == b - a
        finally:
            d.close()


def test_merge_cli_shards_counts_header_lines(tmp_path):
    """the gather of `ntedit --shard` outputs must not sniff headers by prefix -- a contig may be called
    "#chr1", and its VCF rows start with '#' like the header lines do"""
    from ntedit_amd.merge import merge_cli_shards, TSV_HEADER_LINES, VCF_HEADER_LINES
    vcf_hdr = b"".join(b"##h%d\n" % i for i in range(VCF_HEADER_LINES - 1)) + b"#CHROM\tPOS\n"
    tsv_hdr = b"ID\tbpPosition+1\tOriginalBase\n"
    assert TSV_HEADER_LINES == 1
    shards = []
    # shard 0 holds contigs 0 and 2, shard 1 holds contig 1; contig 0 is called "#c0"
    content = {0: (b">#c0\nACGT\n", b"#c0\t3\tA\n", b"#c0\t3\t.\tA\tC\n"),
               1: (b">c1\nGG\n", b"c1\t1\tG\n", b"c1\t1\t.\tG\tT\n"),
               2: (b">c2\nTTT\n", b"", b"")}
    for s, ords in enumerate(([0, 2], [1])):
        pre = str(tmp_path / ("s%d" % s))
        with open(pre + "_edited.fa", "wb") as fa, open(pre + "_changes.tsv", "wb") as tsv, \
                open(pre + "_variants.vcf", "wb") as vcf, open(pre + ".index.tsv", "wb") as idx:
            tsv.write(tsv_hdr)
            vcf.write(vcf_hdr)
            idx.write(b"# ordinal fa tsv vcf\n")
            for o in ords:
                f, t, v = content[o]
                fa.write(f)
                tsv.write(t)
                vcf.write(v)
                idx.write(b"%d %d %d %d\n" % (o, len(f), len(t), len(v)))
        shards.append(pre)
    out = str(tmp_path / "out")
    assert merge_cli_shards(out, shards) == 3
    assert open(out + "_edited.fa", "rb").read() == b"".join(content[o][0] for o in range(3))
    assert open(out + "_changes.tsv", "rb").read() == tsv_hdr + b"".join(content[o][1] for o in range(3))
    assert open(out + "_variants.vcf", "rb").read() == vcf_hdr + b"".join(content[o][2] for o in range(3))
