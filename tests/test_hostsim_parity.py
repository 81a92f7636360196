"""CPU tier: the product's event machine + host renderer (host build of
nte_machine.h / render.cpp, tests/hostsim) against the oracle, byte for byte.
Same configurations as the GPU parity test."""
import filecmp
import os

import numpy as np

import pytest

import helpers as H


@pytest.mark.parametrize("two_pass", [False, True, "nowin"])
@pytest.mark.parametrize("ci", range(len(H.PARITY_CONFIGS)))
def test_hostsim_matches_oracle(tmp_path, ci, two_pass, oracle_build, monkeypatch):
    # "nowin": without the per-position character window every candidate is evaluated by
    # the general rope-walking code (the path taken near contig ends on the GPU)
    monkeypatch.delenv("HOSTSIM_NO_WINDOW", raising=False)
    if two_pass == "nowin":
        monkeypatch.setenv("HOSTSIM_NO_WINDOW", "1")
        two_pass = False
    # two_pass: the launch scheme of the GPU path (events needing an indel sweep are
    # postponed to a second, sweep-only pass) must not change anything
    if two_pass:
        monkeypatch.setenv("HOSTSIM_TWO_PASS", "1")
    else:
        monkeypatch.delenv("HOSTSIM_TWO_PASS", raising=False)
    case_kw, par_kw = H.PARITY_CONFIGS[ci]
    case = H.make_case(str(tmp_path), 3000 + ci, **case_kw)
    hp = H.default_params(**par_kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"), case["rep"])
    rep = H.load_bf(case["rep"]) if case["rep"] else None
    rc, nev, nap = H.run_hostsim(H.read_fasta(case["draft"]), H.load_bf(case["bf"]), hp, str(tmp_path / "h"), rep)
    assert rc == 0
    assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / "h_changes.tsv"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / "h_edited.fa"), shallow=False)
    # _variants.vcf: everything but the date / input-path header lines
    assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "h_variants.vcf"))
    assert nev >= nap > 0


def test_partition_independence(tmp_path, oracle_build):
    """The event decomposition must not change the result: any start grid and a
    different batch layout (contig order / concatenation offsets) give identical edits."""
    case = H.make_case(str(tmp_path), 77, p_sub=5e-3, p_ins=1e-3, p_del=1e-3)
    bf = H.load_bf(case["bf"])
    recs = H.read_fasta(case["draft"])
    outs = []
    for grid in (1, 4, 32, 4096):
        hp = H.default_params(start_grid=grid)
        rc, _, _ = H.run_hostsim(recs, bf, hp, str(tmp_path / ("g%d" % grid)))
        assert rc == 0
        outs.append(open(str(tmp_path / ("g%d_changes.tsv" % grid))).read())
    assert len(set(outs)) == 1
    # reversed contig order: same per-contig rows
    rc, _, _ = H.run_hostsim(recs[::-1], bf, H.default_params(), str(tmp_path / "rev"))
    assert rc == 0
    a = sorted(outs[0].splitlines()[1:])
    b = sorted(open(str(tmp_path / "rev_changes.tsv")).read().splitlines()[1:])
    assert a == b


def test_vcf_annotations(tmp_path, oracle_build):
    """-l: variants found in the annotation VCF get its INFO field appended, others ^NA;
    plain and gzipped annotation files; SNV and polish mode"""
    import gzip
    for snv in (0, 1):
        case = H.make_case(str(tmp_path / ("c%d" % snv)), 4242 + snv, n=8000, contigs=2, p_sub=5e-3, p_ins=1e-3, p_del=1e-3)
        hp = H.default_params(snv=snv)
        H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "pre"))
        rows = [l.split("\t") for l in H.vcf_body(str(tmp_path / "pre_variants.vcf")) if not l.startswith("#")]
        assert rows
        # annotate every second variant: CHROM POS ID REF ALT QUAL FILTER INFO
        ann = str(tmp_path / ("ann%d.vcf" % snv))
        with open(ann, "w") as f:
            f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n")
            for i, r in enumerate(rows):
                if i % 2 == 0:
                    alt = r[4].split(",")[0]
                    f.write("%s\t%s\trs%d\t%s\t%s\t.\t.\tCLNSIG=test%d;X=1\n" % (r[0], r[1], i, r[3].upper(), alt.upper(), i))
        with open(ann, "rb") as fi, gzip.open(ann + ".gz", "wb") as fo:
            fo.write(fi.read())
        for a in (ann, ann + ".gz"):
            H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"), annot_path=a)
            rc, _, _ = H.run_hostsim(H.read_fasta(case["draft"]), H.load_bf(case["bf"]), hp, str(tmp_path / "h"), annot_path=a)
            assert rc == 0
            ob = H.vcf_body(str(tmp_path / "o_variants.vcf"))
            assert ob == H.vcf_body(str(tmp_path / "h_variants.vcf"))
            assert any("CLNSIG=test" in l for l in ob) and any("^NA" in l for l in ob)


def test_saturated_filter_chains(tmp_path, oracle_build):
    """A saturated filter (h=1, ~50 % of the bits set) with a low acceptance bar: the serial run chains edit
    after edit and hardly ever returns to a clean state, so nearly every speculative event would walk to
    the end of its contig.  The launch budget parks them; the resolver re-runs the applied ones.  Found by
    tests/tools/fuzz_parity.py (seed 200059)."""
    case = H.make_case(str(tmp_path), 200059, n=27022, contigs=2, k=40, hashes=1, p_sub=0.01, p_ins=0.0, p_del=0.002,
                       flavor="iupac sec", bfbytes=54390)
    kw = dict(max_insertions=4, max_deletions=0, min_contig_len=41, missing_threshold=9.0, edit_threshold=25.0)
    H.run_oracle(case["draft"], case["bf"], H.default_params(**kw), str(tmp_path / "o"), case["rep"])
    rc, nev, nap = H.run_hostsim(H.read_fasta(case["draft"]), H.load_bf(case["bf"]), H.default_params(event_budget=512, **kw),
                                 str(tmp_path / "h"), H.load_bf(case["rep"]))
    assert rc == 0
    assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / "h_changes.tsv"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / "h_edited.fa"), shallow=False)
    assert nap < nev / 10  # almost everything was speculation


@pytest.mark.parametrize("kw", [dict(snv=1, mask=1), dict(mask=1), dict(snv=1, mode=2), dict()])
def test_last_kmer_is_never_a_seed(tmp_path, oracle_build, kw):
    case = H.make_tail_case(str(tmp_path))
    hp = H.default_params(min_contig_len=0, **kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"))
    rc, _, _ = H.run_hostsim(H.read_fasta(case["draft"]), H.load_bf(case["bf"]), hp, str(tmp_path / "h"))
    assert rc == 0
    for suf in ("_changes.tsv", "_edited.fa"):
        assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("h" + suf)), shallow=False), suf
    assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "h_variants.vcf"))


@pytest.mark.parametrize("unit,threads,kw", [(1 << 20, 4, dict()), (20000, 3, dict(mask=1)), (1 << 20, 1, dict(mode=1)),
                                             (5000, 8, dict(snv=1))])
def test_many_short_contigs(tmp_path, oracle_build, monkeypatch, unit, threads, kw):
    """thousands of contigs: the renderer works in units of consecutive contigs, concurrently, and writes them
    in input order"""
    monkeypatch.setenv("HOSTSIM_RENDER_UNIT", str(unit))
    monkeypatch.setenv("HOSTSIM_RENDER_THREADS", str(threads))
    case = H.make_many_case(str(tmp_path), n_contigs=600 if kw.get("snv") else 3000)
    hp = H.default_params(min_contig_len=100, **kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"))
    rc, nev, nap = H.run_hostsim(H.read_fasta(case["draft"]), H.load_bf(case["bf"]), hp, str(tmp_path / "h"))
    assert rc == 0 and nap > 100
    for suf in ("_changes.tsv", "_edited.fa"):
        assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("h" + suf)), shallow=False), suf
    assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "h_variants.vcf"))


CONTIG_END_PARAMS = [dict(), dict(max_deletions=10, max_insertions=5), dict(mode=1), dict(mode=2, max_insertions=2),
                     dict(max_deletions=0), dict(max_insertions=0, max_deletions=3), dict(snv=1), dict(mask=1),
                     dict(k=40), dict(jump=1, max_deletions=8)]


@pytest.mark.parametrize("two_pass", [False, True])
@pytest.mark.parametrize("kw", CONTIG_END_PARAMS)
def test_errors_in_front_of_contig_ends(tmp_path, oracle_build, monkeypatch, kw, two_pass):
    """errors at every distance from a contig's end (helpers.make_contig_end_case): the window of a failing position is cut
    where the contig ends, the walks behind a deletion with it, the last k positions are given up -- as the reference's
    loops do when roll() fails (ntedit.cpp:1216-1247, 1494-1519, 1826-1858)"""
    kw = dict(kw)
    k = kw.pop("k", 25)
    if two_pass:
        monkeypatch.setenv("HOSTSIM_TWO_PASS", "1")
    else:
        monkeypatch.delenv("HOSTSIM_TWO_PASS", raising=False)
    case = H.make_contig_end_case(str(tmp_path), k=k)
    hp = H.default_params(min_contig_len=0, **kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"))
    rc, nev, nap = H.run_hostsim(H.read_fasta(case["draft"]), H.load_bf(case["bf"]), hp, str(tmp_path / "h"))
    assert rc == 0 and nap > 20
    for suf in ("_changes.tsv", "_edited.fa"):
        assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("h" + suf)), shallow=False), suf
    assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "h_variants.vcf"))


@pytest.mark.parametrize("ci", [0, 5, 9, 10, 22, 28, 31])
def test_edit_records_rebuild_the_tsv(tmp_path, ci, oracle_build):
    """the POD edit records of the C ABI (ntedit_hip_result_edits; here straight from the product's renderer on
    the host simulation) carry exactly what the _changes.tsv rows say"""
    import ctypes
    case_kw, par_kw = H.PARITY_CONFIGS[ci]
    case = H.make_case(str(tmp_path), 8000 + ci, **case_kw)
    bf = H.load_bf(case["bf"])
    rep = H.load_bf(case["rep"]) if case["rep"] else None
    hp = H.default_params(**par_kw)
    recs = H.read_fasta(case["draft"])
    dump = str(tmp_path / "edits.bin")
    H.hostsim_lib().hostsim_set_render_extras(None, None, None, ctypes.c_uint(0), dump.encode())
    rc, _, _ = H.run_hostsim(recs, bf, hp, str(tmp_path / "h"), rep)
    assert rc == 0
    edits, pool = H.load_edits_dump(dump)
    names = [n for n, s in recs if len(s) >= hp.min_contig_len]
    tsv = open(str(tmp_path / "h_changes.tsv"), "rb").read()
    header = tsv[:tsv.index(b"\n") + 1]
    assert H.tsv_from_edits(edits, pool, names, header) == tsv
    assert len(edits) > 5
    kinds = set(int(x) for x in edits["kind"])
    assert kinds <= {1, 2, 3, 4} and 1 in kinds
    # positions are non-decreasing within a contig
    for c in set(int(x) for x in edits["contig"]):
        pos = edits["draft_pos"][edits["contig"] == c].astype(np.int64)
        assert (np.diff(pos) >= -1).all()


def test_filter_file_size_not_a_multiple_of_8(tmp_path, oracle_build):
    """a .bf whose `bytes` is not a multiple of 8: the header's size IS the modulus of the slot arithmetic (btllib's
    file constructor takes it as it is; only filters it builds are rounded up)"""
    case = H.make_case(str(tmp_path), 8100, n=20000, contigs=2, bfbytes=1 << 16)
    raw = open(case["bf"], "rb").read()
    body = raw.index(b"[HeaderEnd]\n") + len(b"[HeaderEnd]\n")
    odd = (1 << 16) - 3
    with open(str(tmp_path / "odd.bf"), "wb") as f:
        f.write(raw[:body].replace(b"bytes = %d" % (1 << 16), b"bytes = %d" % odd))
        f.write(raw[body:body + odd])
    bf = H.load_bf(str(tmp_path / "odd.bf"))
    assert bf["bytes"] == odd and bf["data"].size == odd
    hp = H.default_params()
    H.run_oracle(case["draft"], str(tmp_path / "odd.bf"), hp, str(tmp_path / "o"))
    rc, _, _ = H.run_hostsim(H.read_fasta(case["draft"]), bf, hp, str(tmp_path / "h"))
    assert rc == 0
    assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / "h_changes.tsv"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / "h_edited.fa"), shallow=False)
    # and it is NOT what the rounded-up modulus gives
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "r"))
    assert not filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / "r_changes.tsv"), shallow=False)


def _gap_case(tmp, gap, seed=91):
    """a contig with a scaffold gap of `gap` Ns, errors right in front of and right behind it"""
    rng = np.random.default_rng(seed)
    truth = H.random_genome(rng, 30000)
    H.write_fasta(os.path.join(tmp, "truth.fa"), [(b"t0", truth)])
    H.mkbf([os.path.join(tmp, "truth.fa")], os.path.join(tmp, "t.bf"), k=25, hashes=3, nbytes=1 << 17)
    d = bytearray(H.mutate(rng, truth, 1e-3, 2e-4, 2e-4))
    cut = len(d) // 2
    for p in (cut - 30, cut - 9, cut + 4, cut + 33):  # substitutions whose k-mers reach the gap
        d[p] = ord("ACGT"[("ACGT".index(chr(d[p])) + 1) % 4])
    draft = bytes(d[:cut]) + b"N" * gap + bytes(d[cut:])
    H.write_fasta(os.path.join(tmp, "draft.fa"), [(b"scaffold1 with a gap", draft), (b"c2", H.mutate(rng, truth[:5000], 2e-3, 0, 0))])
    return os.path.join(tmp, "draft.fa"), os.path.join(tmp, "t.bf")


def test_scaffold_gap(tmp_path, oracle_build):
    """A run of a million Ns with errors on both sides: an event that ends in front of the gap stops inside it as soon
    as its state is clean (no k-mer in there is in the absent bitmap) instead of rolling through it base by base -- the
    serial program's arrival behind the gap is the start of another event.  Same bytes as the oracle."""
    draft, bf = _gap_case(str(tmp_path), 1_000_000)
    for kw in (dict(), dict(mode=1, mask=1), dict(snv=1)):
        hp = H.default_params(**kw)
        H.run_oracle(draft, bf, hp, str(tmp_path / "o"))
        rc, nev, nap = H.run_hostsim(H.read_fasta(draft), H.load_bf(bf), hp, str(tmp_path / "h"))
        assert rc == 0 and nap > 10
        for suf in ("_changes.tsv", "_edited.fa"):
            assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("h" + suf)), shallow=False), (suf, kw)
        assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "h_variants.vcf"))


@pytest.mark.parametrize("kw", H.SWEEP_RICH_PARAMS[:3])
def test_sweep_rich_counting_filter(tmp_path, oracle_build, kw):
    """Counting filter, -p 2 cutting into the k-mers that are there, an error-rich draft: runs in which position after
    position needs an indel sweep, behind substitutions as well (bench.py --counting in small; a tenth of the events of
    such a batch are most of its machine time).  Same bytes as the oracle."""
    case = H.make_sweep_rich_case(str(tmp_path))
    hp = H.default_params(min_threshold=2, **kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"))
    rc, nev, nap = H.run_hostsim(H.read_fasta(case["draft"]), H.load_bf(case["bf"]), hp, str(tmp_path / "h"))
    assert rc == 0 and nap > 10
    for suf in ("_changes.tsv", "_edited.fa"):
        assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("h" + suf)), shallow=False), (suf, kw)
    assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "h_variants.vcf"))
