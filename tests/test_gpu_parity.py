"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bit-exact bar: bitmaps, filter bytes, _edited.fa and _changes.tsv must be
byte-identical.  Nothing here reads /root/reference."""
import ctypes
import filecmp
import os
import sys

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def polisher():
    import ntedit_amd
    p = ntedit_amd.Polisher(0)
    yield p
    p.close()


def _hip_params(**kw):
    import ntedit_amd
    return ntedit_amd.default_params(**kw)


def _load_filters(pol, case):
    pol.load_filter_file(case["bf"], 0)
    if case["rep"]:
        pol.load_filter_file(case["rep"], 1)


def _fresh(pol_cls=None, force_rounds=True, general=False):
    """a polisher for the small cases of this file: the event rounds (which the library only uses from ~2 M events of a
    batch on) are forced, so that their selection / verification kernels see every configuration"""
    import ntedit_amd
    pol = ntedit_amd.Polisher(0)
    if force_rounds:
        pol.set_tuning("force_rounds", 1)
    if general:
        # the general instantiation of the machine kernels instead of the one specialised for the configuration
        pol.set_tuning("machine_cfg", 0)
    return pol


@pytest.mark.parametrize("ci", range(len(H.PARITY_CONFIGS)))
def test_polish_matches_oracle(tmp_path, ci, oracle_build):
    case_kw, par_kw = H.PARITY_CONFIGS[ci]
    case = H.make_case(str(tmp_path), 1000 + ci, **case_kw)
    hp = H.default_params(**par_kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"), case["rep"])
    # (both ways of running the events: in rounds, all at once; every third configuration on the general machine kernels
    # although a specialised instantiation would do)
    pol = _fresh(force_rounds=ci % 2 == 0, general=ci % 3 == 0)
    try:
        _load_filters(pol, case)
        pol.set_params(_hip_params(**par_kw))
        st = pol.polish_records(H.read_fasta(case["draft"]), str(tmp_path / "g"))
    finally:
        pol.close()
    assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / "g_changes.tsv"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / "g_edited.fa"), shallow=False)
    assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "g_variants.vcf"))
    assert st.events >= st.events_applied


@pytest.mark.parametrize("ci", [0, 1, 3, 16, 23, 24, 25, 26, 31, 32, 33, 34])
def test_screen_bitmap_matches_oracle(tmp_path, ci, polisher, oracle_build):
    case_kw, par_kw = H.PARITY_CONFIGS[ci]
    case = H.make_case(str(tmp_path), 2000 + ci, **case_kw)
    bf = H.load_bf(case["bf"])
    blob, offs, lens, names = H.pack_batch(H.read_fasta(case["draft"]))
    polisher.set_filter(bf["data"], bf["hash_num"], bf["k"], counting=bf["counting"])
    polisher.set_params(_hip_params(**par_kw))
    got = polisher.screen(blob)
    want = H.oracle_screen(blob, bf, par_kw.get("min_threshold", 1))
    assert got.shape == want.shape
    assert np.array_equal(got, want)
    assert int(np.unpackbits(got.view(np.uint8)).sum()) > 0


def test_screen_edge_sizes(tmp_path, polisher, oracle_build):
    """empty / shorter-than-k / exactly-k / tile-boundary inputs"""
    rng = np.random.default_rng(7)
    truth = H.random_genome(rng, 40000)
    H.write_fasta(str(tmp_path / "t.fa"), [(b"t", truth)])
    H.mkbf([str(tmp_path / "t.fa")], str(tmp_path / "t.bf"), k=25, hashes=3, nbytes=1 << 16)
    bf = H.load_bf(str(tmp_path / "t.bf"))
    polisher.set_filter(bf["data"], bf["hash_num"], bf["k"])
    draft = H.mutate(rng, truth, 5e-3, 1e-3, 1e-3)
    for n in (1, 24, 25, 26, 63, 64, 65, 16383, 16384, 16385, 16384 + 24, 16384 + 25, 32768, 39000):
        blob = draft[:n]
        got = polisher.screen(blob)
        want = H.oracle_screen(blob, bf)
        assert np.array_equal(got, want), n


def test_filter_build_matches_oracle(tmp_path, polisher, oracle_build):
    """GPU filter build (atomicOr) == the oracle's mkbf, byte for byte"""
    rng = np.random.default_rng(11)
    seqs = [H.random_genome(rng, 50000), H.random_genome(rng, 777)]
    s0 = bytearray(seqs[0])
    s0[100:130] = b"N" * 30
    s0[5000] = ord("R")
    s0[6000:6100] = bytes(s0[6000:6100]).lower()
    seqs[0] = bytes(s0)
    H.write_fasta(str(tmp_path / "g.fa"), [(b"a", seqs[0]), (b"b", seqs[1])])
    for k, h, nbytes in ((25, 3, 1 << 16), (31, 4, 100003 * 8), (40, 2, 1 << 15)):
        H.mkbf([str(tmp_path / "g.fa")], str(tmp_path / "o.bf"), k=k, hashes=h, nbytes=nbytes)
        want = H.load_bf(str(tmp_path / "o.bf"))
        polisher.filter_alloc(nbytes, h, k)
        polisher.filter_insert(seqs[0] + b"\n" + seqs[1] + b"\n")
        got = polisher.filter_download()
        assert np.array_equal(got, want["data"]), (k, h, nbytes)
        polisher.filter_save_file(str(tmp_path / "g.bf"))
        again = H.load_bf(str(tmp_path / "g.bf"))
        assert again["k"] == k and again["hash_num"] == h and np.array_equal(again["data"], want["data"])


def test_demo_ecoli(tmp_path, oracle_build):
    """configs[1]: E. coli demo draft, proxy filter resident in HBM, vs the oracle and vs the
    reference's committed changes.tsv (every one of its 4,997 rows accounted for: helpers.check_demo_rows)."""
    import subprocess, sys
    demo = os.path.join(H.GOLDEN, "demo")
    draft = os.path.join(demo, "ecoliWithMismatches001Indels0001.fa.gz")
    ref_tsv = os.path.join(demo, "ecoli_ntedit_k25_changes.tsv")
    subprocess.run([sys.executable, os.path.join(H.GOLDEN, "recon_demo.py"), draft, ref_tsv,
                    str(tmp_path / "truth.fa")], check=True)
    H.mkbf([str(tmp_path / "truth.fa")], str(tmp_path / "p.bf"), k=25, hashes=3, nbytes=1 << 28)
    hp = H.default_params(max_insertions=4, max_deletions=5)
    H.run_oracle(draft, str(tmp_path / "p.bf"), hp, str(tmp_path / "o"))
    pol = _fresh()
    try:
        pol.load_filter_file(str(tmp_path / "p.bf"))
        pol.set_params(_hip_params(max_insertions=4, max_deletions=5))
        st = pol.polish_records(H.read_fasta(draft), str(tmp_path / "g"))
    finally:
        pol.close()
    assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / "g_changes.tsv"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / "g_edited.fa"), shallow=False)
    H.check_demo_rows(open(ref_tsv).read().splitlines(), open(str(tmp_path / "g_changes.tsv")).read().splitlines())
    assert st.bases > 4_600_000


@pytest.mark.parametrize("ci", [0, 1, 3, 16, 23, 24, 25, 100, 101])
def test_binned_screen_matches_oracle(tmp_path, ci, oracle_build):
    """the L2-partitioned screening pipeline (write-combining partition of the probes by filter slice, then the
    L2-resident probe) gives the same bitmap as the oracle, incl. several chunks, non-power-of-two filters,
    1..6 hashes (6: falls back to the direct kernel) and a one-slice filter (every record of a workgroup meets
    the same ring: records wait for their ring slot all the time).  Cases 100/101: filters of many slices."""
    import ntedit_amd
    case_kw, _ = H.PARITY_CONFIGS[ci] if ci < 100 else (dict(bfbytes=(1 << 27) if ci == 100 else 100000007 * 8, n=150000,
                                                           flavor="N rep"), {})
    case = H.make_case(str(tmp_path), 4000 + ci, **case_kw)
    bf = H.load_bf(case["bf"])
    blob, offs, lens, names = H.pack_batch(H.read_fasta(case["draft"]))
    want = H.oracle_screen(blob, bf)
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_tuning("bin_chunk", 3 * 16384)
        pol.set_filter(bf["data"], bf["hash_num"], bf["k"])
        pol.set_params(ntedit_amd.default_params(screen_mode=2))
        got = pol.screen(blob)
        pol.set_tuning("bin_chunk", 0)
        one_chunk = pol.screen(blob)
        pol.set_tuning("probe_parts_log2", 1 + ci % 3)  # slices probed in 2 / 4 / 8 parts (filters beyond 4 GiB)
        in_parts = pol.screen(blob)
        pol.set_tuning("probe_parts_log2", 3 - ci % 3)
        in_parts_2 = pol.screen(blob)
        pol.set_tuning("probe_parts_log2", 0xFFFFFFFF)
        pol.set_tuning("bin_scatter", 1)  # (the barrier-free partition kernel)
        free_one = pol.screen(blob)
        pol.set_tuning("bin_chunk", 3 * 16384)
        free_chunks = pol.screen(blob)
        pol.set_params(ntedit_amd.default_params(screen_mode=1))
        direct = pol.screen(blob)
    finally:
        pol.close()
    assert np.array_equal(direct, want)
    assert np.array_equal(got, want)
    assert np.array_equal(one_chunk, want)
    assert np.array_equal(in_parts, want)
    assert np.array_equal(in_parts_2, want)
    assert np.array_equal(free_one, want)
    assert np.array_equal(free_chunks, want)


@pytest.mark.parametrize("xcc", [0, 3, 7, 12])
def test_binned_probe_on_any_xcd_count(tmp_path, xcc, oracle_build):
    """the probe stage must visit every slice whatever XCDs the device exposes: with every wavefront pretending
    to run on the same XCD (a device in CPX mode has one; id 12 is one a real device never reports) the bitmap is
    still the oracle's"""
    import ntedit_amd
    case = H.make_case(str(tmp_path), 4700 + xcc, bfbytes=1 << 27, n=120000, flavor="N")
    bf = H.load_bf(case["bf"])
    blob, offs, lens, names = H.pack_batch(H.read_fasta(case["draft"]))
    want = H.oracle_screen(blob, bf)
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_filter(bf["data"], bf["hash_num"], bf["k"])
        pol.set_params(ntedit_amd.default_params(screen_mode=2))
        pol.set_tuning("force_xcc", xcc + 1)
        got = pol.screen(blob)
    finally:
        pol.close()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("scatter", [0, 1])
@pytest.mark.parametrize("percent", [50, 5])
def test_binned_overflow_list(tmp_path, percent, scatter, oracle_build):
    """record runs sized far below what the pairs get: the excess goes through the overflow list (direct probes);
    same bitmap.  A low-complexity draft does this in the field."""
    import ntedit_amd
    case = H.make_case(str(tmp_path), 4800 + percent, bfbytes=1 << 26, n=200000, flavor="N rep")
    bf = H.load_bf(case["bf"])
    blob, offs, lens, names = H.pack_batch(H.read_fasta(case["draft"]))
    want = H.oracle_screen(blob, bf)
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_filter(bf["data"], bf["hash_num"], bf["k"])
        pol.set_params(ntedit_amd.default_params(screen_mode=2))
        pol.set_tuning("bin_cap_percent", percent)
        pol.set_tuning("bin_scatter", scatter)
        got = pol.screen(blob)
    finally:
        pol.close()
    assert np.array_equal(got, want)


@pytest.mark.parametrize("scatter", [0, 1])
def test_binned_overflow_list_runs_out(tmp_path, scatter, oracle_build):
    """an overflow list that is too small for what the runs send it (round 6: the list is handed out in blocks of 256
    entries per partition wavefront, one global atomic per block): the record chunks that lost probes -- and only
    those -- are screened again by the direct kernel before anything reads the bitmap; nothing is remembered, the next
    call is back on the partitioned pipeline.  Bitmap and polished files identical to the oracle's."""
    import ntedit_amd
    case = H.make_case(str(tmp_path), 4850 + scatter, bfbytes=1 << 26, n=200000, contigs=3, flavor="N rep")
    bf = H.load_bf(case["bf"])
    recs = H.read_fasta(case["draft"])
    blob, offs, lens, names = H.pack_batch(recs)
    want = H.oracle_screen(blob, bf)
    hp = H.default_params()
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"), case["rep"])
    pol = _fresh()
    try:
        pol.set_filter(bf["data"], bf["hash_num"], bf["k"])
        pol.set_params(ntedit_amd.default_params(screen_mode=2))
        pol.set_tuning("bin_scatter", scatter)
        pol.set_tuning("bin_cap_percent", 5)
        pol.set_tuning("bin_chunk", 3 * 16384)
        pol.set_tuning("bin_ovf_cap", 4096)          # 16 blocks: every chunk runs out
        assert np.array_equal(pol.screen(blob), want)
        st = pol.polish_records(recs, str(tmp_path / "g"))
        assert st.screen_binned and st.screen_chunks_direct == st.screen_launches and st.screen_launches >= 4
        pol.set_tuning("bin_ovf_cap", 120000)        # some chunks fit, some do not
        assert np.array_equal(pol.screen(blob), want)
        st_mixed = pol.polish_records(recs, str(tmp_path / "g1"))
        pol.set_tuning("bin_ovf_cap", 0)             # the list at its real size: nothing is lost, nothing was remembered
        assert np.array_equal(pol.screen(blob), want)
        st2 = pol.polish_records(recs, str(tmp_path / "g2"))
        assert st2.screen_binned and st2.screen_chunks_direct == 0 and st2.screen_overflow_records > 0
        pol.set_tuning("bin_cap_percent", 0)         # and the runs at their real size: no overflow on this draft
        st3 = pol.polish_records(recs, str(tmp_path / "g3"))
        assert st3.screen_binned and st3.screen_chunks_direct == 0
    finally:
        pol.close()
    print("chunks re-screened with a list of 120000 entries: %d of %d" % (st_mixed.screen_chunks_direct, st_mixed.screen_launches))
    for g in ("g", "g1", "g2", "g3"):
        assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / (g + "_changes.tsv")), shallow=False)
        assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / (g + "_edited.fa")), shallow=False)


def test_binned_homopolymer_draft(tmp_path, oracle_build):
    """a draft of very few distinct k-mers (homopolymer and dinucleotide runs): all probes of a workgroup meet a
    handful of rings and runs -- ring waits, run overflow -- and the bitmap is still the oracle's"""
    import ntedit_amd
    rng = np.random.default_rng(77)
    parts = [b"A" * 60000, b"AC" * 30000, bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 40000)), b"T" * 50000, b"ACG" * 20000]
    blob = b"\n".join(parts) + b"\n"
    truth = str(tmp_path / "t.fa")
    with open(truth, "w") as fh:
        for i, s_ in enumerate(parts):
            fh.write(">c%d\n%s\n" % (i, s_[: len(s_) // 2].decode()))
    H.mkbf([truth], str(tmp_path / "f.bf"), k=25, hashes=3, nbytes=1 << 26)
    bf = H.load_bf(str(tmp_path / "f.bf"))
    want = H.oracle_screen(blob, bf)
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_filter(bf["data"], bf["hash_num"], bf["k"])
        pol.set_params(ntedit_amd.default_params(screen_mode=2))
        got = pol.screen(blob)
        pol.set_tuning("bin_scatter", 1)
        got_free = pol.screen(blob)
    finally:
        pol.close()
    assert np.array_equal(got, want)
    assert np.array_equal(got_free, want)


@pytest.mark.parametrize("ci", [0, 8, 10, 23])
def test_polish_with_binned_screen(tmp_path, ci, oracle_build):
    case_kw, par_kw = H.PARITY_CONFIGS[ci]
    case = H.make_case(str(tmp_path), 5000 + ci, **case_kw)
    hp = H.default_params(**par_kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"), case["rep"])
    pol = _fresh()
    try:
        _load_filters(pol, case)
        pol.set_params(_hip_params(screen_mode=2, **par_kw))
        pol.polish_records(H.read_fasta(case["draft"]), str(tmp_path / "g"))
    finally:
        pol.close()
    assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / "g_changes.tsv"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / "g_edited.fa"), shallow=False)


@pytest.mark.parametrize("screen_mode", [0, 2])
@pytest.mark.parametrize("ci", [0, 1, 9, 10, 20, 22])
def test_polish_chunk_pipeline(tmp_path, ci, screen_mode, oracle_build):
    """many small pipeline chunks (screening of chunk j+1 overlaps the event machine of chunk j
    on a second stream) must give the same bytes as one chunk"""
    case_kw, par_kw = H.PARITY_CONFIGS[ci]
    case_kw = dict(case_kw, contigs=7, n=30000)
    case = H.make_case(str(tmp_path), 6000 + ci, **case_kw)
    hp = H.default_params(**par_kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"), case["rep"])
    pol = _fresh()
    pol.set_tuning("chunk_bytes", 50000)
    try:
        _load_filters(pol, case)
        pol.set_params(_hip_params(screen_mode=screen_mode, **par_kw))
        st = pol.polish_records(H.read_fasta(case["draft"]), str(tmp_path / "g"))
        assert st.screen_launches >= 4
        # and again on the warm context (buffers already sized)
        pol.polish_records(H.read_fasta(case["draft"]), str(tmp_path / "g2"))
    finally:
        pol.close()
    for g in ("g", "g2"):
        assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / (g + "_changes.tsv")), shallow=False)
        assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / (g + "_edited.fa")), shallow=False)


@pytest.mark.parametrize("ci", [0, 2, 3, 5, 9, 17, 31])
def test_polish_packed_batch(tmp_path, ci, oracle_build):
    """the batch crosses PCIe in the packed form (4-bit codes + a case bit per base, unpacked on the device) -- whole, and
    in pieces under the screening: the same bytes come out (N runs, lower case, IUPAC codes; a draft with a byte the
    form cannot carry is handed over as bytes)"""
    case_kw, par_kw = H.PARITY_CONFIGS[ci]
    case = H.make_case(str(tmp_path), 7700 + ci, **case_kw)
    hp = H.default_params(**par_kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"), case["rep"])
    recs = H.read_fasta(case["draft"])
    for pieces in (0, 1):
        pol = _fresh()
        pol.send_packed = True
        if pieces:
            pol.set_tuning("h2d_piece", 16384)
        try:
            _load_filters(pol, case)
            pol.set_params(_hip_params(**par_kw))
            pol.polish_records(recs, str(tmp_path / "g"))
        finally:
            pol.close()
        for suf in ("_changes.tsv", "_edited.fa"):
            assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("g" + suf)), shallow=False), (suf, pieces)
        assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "g_variants.vcf"))


@pytest.mark.parametrize("chunked", [0, 1])
@pytest.mark.parametrize("screen_mode", [1, 2])
def test_polish_batch_arriving_in_pieces(tmp_path, screen_mode, chunked, oracle_build):
    """a batch in host memory crosses PCIe in pieces while the pieces that have arrived are screened (direct kernel:
    tile ranges per piece; partitioned pipeline: record chunks that grow, each waiting for its own pieces): same bytes.
    chunked: the batch is polished in pipeline chunks at the same time (the pieces cross on a stream of their own, the
    event machine of a chunk runs while later chunks arrive)"""
    case_kw, par_kw = H.PARITY_CONFIGS[0]
    case = H.make_case(str(tmp_path), 6500 + screen_mode, **dict(case_kw, contigs=6, n=120000))
    hp = H.default_params(**par_kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"), case["rep"])
    pol = _fresh()
    pol.set_tuning("h2d_piece", 4 * 16384)
    if chunked:
        pol.set_tuning("chunk_bytes", 250000)
    try:
        _load_filters(pol, case)
        pol.set_params(_hip_params(screen_mode=screen_mode, **par_kw))
        st = pol.polish_records(H.read_fasta(case["draft"]), str(tmp_path / "g"))
        # (partitioned pipeline, round 6: a record chunk is what has arrived when the chunk before it is through -- on an
        # input this small that is everything after the first chunk)
        assert st.screen_launches >= (2 if screen_mode == 2 and not chunked else 3)
        # ... and with the fixed schedule of rounds 3-5 (one piece, three, eight, the rest)
        pol.set_tuning("h2d_fixed_schedule", 1)
        st2 = pol.polish_records(H.read_fasta(case["draft"]), str(tmp_path / "g2"))
        assert st2.screen_launches >= 3
    finally:
        pol.close()
    for g in ("g", "g2"):
        assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / (g + "_changes.tsv")), shallow=False)
        assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / (g + "_edited.fa")), shallow=False)


def test_golden_cases_on_gpu(tmp_path):
    """committed golden vectors (tests/golden/cases) through the C ABI"""
    import test_golden as TG
    for name in TG.CASES:
        d = os.path.join(H.GOLDEN, "cases", name)
        hp = TG.params_from_file(os.path.join(d, "params.txt"))
        pol = _fresh()
        try:
            pol.load_filter_file(os.path.join(d, "filter.bf"), 0)
            if os.path.exists(os.path.join(d, "secondary.bf")):
                pol.load_filter_file(os.path.join(d, "secondary.bf"), 1)
            kw = {f[0]: getattr(hp, f[0]) for f in hp._fields_}
            pol.set_params(_hip_params(**kw))
            pol.polish_records(H.read_fasta(os.path.join(d, "draft.fa")), str(tmp_path / name))
        finally:
            pol.close()
        assert filecmp.cmp(os.path.join(d, "expected_changes.tsv"), str(tmp_path / (name + "_changes.tsv")), shallow=False), name
        assert filecmp.cmp(os.path.join(d, "expected_edited.fa"), str(tmp_path / (name + "_edited.fa")), shallow=False), name


def test_cli_drop_in(tmp_path, oracle_build):
    """the `ntedit` host binary: reference flag surface, gzipped multi-FASTA with comments and a
    short contig, default output prefix (ntedit.cpp:2496-2502), -e, outputs byte-identical to the oracle"""
    import gzip
    import shutil
    import subprocess
    cli = os.path.join(H.ROOT, "ntedit_amd", "ntedit")
    assert os.path.exists(cli), "build the CLI first (make -C ntedit_amd/csrc)"
    case = H.make_case(str(tmp_path), 7001, flavor="sec N lower", contigs=4, n=40000)
    gz = str(tmp_path / "draft.fa.gz")
    with open(case["draft"], "rb") as fi, gzip.open(gz, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    hp = H.default_params(max_insertions=4, max_deletions=7, mode=1, min_contig_len=50)
    H.run_oracle(gz, case["bf"], hp, str(tmp_path / "o"), case["rep"])
    r = subprocess.run([cli, "-f", gz, "-r", case["bf"], "-e", case["rep"], "-i", "4", "-d", "7", "-m", "1", "-z", "50",
                        "-k", "99", "-t", "48", "-c", "3", "-v", "0", "--batch-bases", "70000", "--report"],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    prefix = "draft.fa.gz_k25_z50_rt.bf_i4_d7_m1"  # <draft>_k<k>_z<z>_r<bf>_i<i>_d<d>_m<m>
    assert filecmp.cmp(str(tmp_path / "o_changes.tsv"), str(tmp_path / (prefix + "_changes.tsv")), shallow=False)
    assert filecmp.cmp(str(tmp_path / "o_edited.fa"), str(tmp_path / (prefix + "_edited.fa")), shallow=False)
    assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / (prefix + "_variants.vcf")))
    vcf = open(str(tmp_path / (prefix + "_variants.vcf"))).read().splitlines()
    assert vcf[0] == "##fileformat=VCFv4.2" and vcf[3] == "##reference=file:" + gz and len(vcf) > 20
    # parameter clamping messages (ntedit.cpp:2485-2493) and a counting filter with -p/-q
    case2 = H.make_case(str(tmp_path / "c2"), 7002, flavor="cbf", contigs=2, n=20000)
    hp2 = H.default_params(min_threshold=2, max_threshold=5)
    H.run_oracle(case2["draft"], case2["bf"], hp2, str(tmp_path / "o2"))
    r = subprocess.run([cli, "-f", case2["draft"], "-r", case2["bf"], "-b", str(tmp_path / "g2"), "-i", "9", "-d", "12",
                        "-p", "2", "-q", "5"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "i parameter too high" in r.stderr and "d parameter too high" in r.stderr
    H.run_oracle(case2["draft"], case2["bf"], H.default_params(min_threshold=2, max_threshold=5, max_insertions=9,
                                                               max_deletions=12), str(tmp_path / "o3"))
    assert filecmp.cmp(str(tmp_path / "o3_changes.tsv"), str(tmp_path / "g2_changes.tsv"), shallow=False)
    assert filecmp.cmp(str(tmp_path / "o3_edited.fa"), str(tmp_path / "g2_edited.fa"), shallow=False)


def _differing_contigs(fa_a, fa_b, tsv_a, tsv_b, limit=5):
    """(count, first few names) of contigs whose FASTA record or TSV rows differ -- only read on a mismatch
    (the files are GB-sized: streamed)"""
    bad = []
    n_bad = 0
    with open(fa_a, "rb") as fa, open(fa_b, "rb") as fb:
        while True:
            ha, hb = fa.readline(), fb.readline()
            if not ha and not hb:
                break
            sa, sb = fa.readline(), fb.readline()
            if ha != hb or sa != sb:
                n_bad += 1
                if len(bad) < limit:
                    bad.append((ha[:40] or hb[:40]).decode(errors="replace").strip())

    def rows(path):
        d = {}
        with open(path, "rb") as f:
            f.readline()
            for line in f:
                d.setdefault(line.split(b"\t", 1)[0], []).append(line)
        return d
    ra, rb = rows(tsv_a), rows(tsv_b)
    for name in set(ra) | set(rb):
        if ra.get(name) != rb.get(name):
            n_bad += 1
            if len(bad) < limit:
                bad.append("tsv:" + name.decode(errors="replace"))
    return n_bad, bad


def _compare_every_contig(pol, job, tmp_path, tag, capsys, rep=False, counting=False, **kw):
    """Polish the whole HBM-resident batch on the GPU, render ALL of it, run the multi-threaded oracle on ALL
    of it with the same filter(s) (downloaded from HBM), and compare the complete _edited.fa, _changes.tsv and
    VCF body byte for byte: every contig of the batch, 0 differences allowed.  Returns the GPU stats."""
    import time
    nc = len(job.lens)
    names = [b"contig%d" % i for i in range(nc)]
    t0 = time.time()
    res = pol.polish_batch(None, job.offsets, job.lens, device_ptr=job.device_ptr, n=job.n_bytes)
    host = job.batch.cpu().numpy()
    fa, tsv, vcf = (str(tmp_path / (tag + s)) for s in ("_edited.fa", "_changes.tsv", "_variants.vcf"))
    pol.write_tsv_header(tsv)
    open(vcf, "wb").close()
    res.write(host, job.offsets, job.lens, names, fa, tsv, append=True, vcf_path=vcf)
    st = res.stats()
    res.free()
    t_gpu = time.time() - t0
    bits = pol.filter_download(0)
    rbits = pol.filter_download(1) if rep else None
    k, h, _, _ = pol.filter_info(0)
    ofa, otsv, ovcf = (str(tmp_path / (tag + s)) for s in ("_ora_edited.fa", "_ora_changes.tsv", "_ora_body.vcf"))
    threads = H.usable_cpus()
    t0 = time.time()
    H.oracle_lib().ora_set_flat_counting(ctypes.c_int(1 if counting else 0))
    try:
        done = H.oracle_polish_flat_mt_files(host, job.offsets, job.lens, names, bits, h, k, threads, fa_path=ofa,
                                             tsv_path=otsv, vcf_path=ovcf, rep_bits=rbits, rep_hash_num=h, **kw)
    finally:
        H.oracle_lib().ora_set_flat_counting(ctypes.c_int(0))
    t_ora = time.time() - t0
    assert done == job.n_bases
    same = (filecmp.cmp(fa, ofa, shallow=False), filecmp.cmp(tsv, otsv, shallow=False),
            filecmp.cmp(vcf, ovcf, shallow=False))
    with capsys.disabled():
        print("\n[%s] %d contigs, %.0f Mbases: GPU step %.1f ms (%.0f Mbases/s; screen %.1f ms, machine %.1f ms), "
              "GPU+render %.1f s, oracle on %d threads %.1f s (%.1f Mbases/s); %d subs %d ins %d del; files identical: %s"
              % (tag, nc, job.n_bases / 1e6, st.ms_total, job.n_bases / st.ms_total / 1e3, st.ms_screen, st.ms_machine,
                 t_gpu, threads, t_ora, job.n_bases / t_ora / 1e6, st.substitutions, st.insertions, st.deletions,
                 same), flush=True)
    if not all(same):
        n_bad, bad = _differing_contigs(fa, ofa, tsv, otsv)
        raise AssertionError("%s: %d contig(s) differ from the oracle, e.g. %s (vcf identical: %s)" %
                             (tag, n_bad, bad, same[2]))
    assert os.path.getsize(tsv) > 1000
    for f in (ofa, otsv, ovcf, fa):  # (GB-sized: do not keep them for pytest's tmp_path retention)
        os.remove(f)
    return st, host, names


def _partition_independence(pol, job, host, names, whole):
    """the per-kind edit counts of the whole batch equal the sum over a 2-way split of its contigs"""
    import torch
    nc = len(job.lens)
    parts = []
    for lo, hi in ((0, nc // 2), (nc // 2, nc)):
        o0 = int(job.offsets[lo])
        o1 = int(job.offsets[hi - 1]) + int(job.lens[hi - 1]) + 1
        offs_h = job.offsets[lo:hi] - np.uint64(o0)
        half = job.batch[o0:o1].clone()  # (device batches have to start 16-byte aligned)
        torch.cuda.synchronize()
        r = pol.polish_batch(None, offs_h, job.lens[lo:hi], device_ptr=half.data_ptr(), n=o1 - o0)
        del half
        r.write(host[o0:o1], offs_h, job.lens[lo:hi], names[lo:hi], None, None)
        s = r.stats()
        parts.append((s.absent_kmers, s.substitutions, s.insertions, s.deletions))
        r.free()
    assert tuple(a + b for a, b in zip(*parts)) == whole


def test_full_size_every_contig(tmp_path, oracle_build, capsys):
    """BASELINE.json configs[3] at full size (3 Gbp draft, 369 contigs 50 kbp - 50 Mbp, k=25, 4 GiB filter):
      1. no false negatives: an unmutated draft has no absent k-mer and gets no edit;
      2. EVERY contig of the batch is byte-identical to the oracle's output (complete _edited.fa, _changes.tsv and
         VCF body; the oracle runs the whole 3 Gbp on all host cores with the same 4 GiB filter);
      3. partition independence: the per-kind edit counts of the whole batch equal the sum over a 2-way split."""
    import torch
    import ntedit_amd
    from ntedit_amd.synth import SyntheticJob

    total = float(os.environ.get("NTEDIT_FULL_BASES", "3e9"))
    fbytes = int(os.environ.get("NTEDIT_FULL_FILTER", str(1 << 32)))
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_params(ntedit_amd.default_params())
        # 1. truth genome against its own filter
        job = SyntheticJob(pol, total, k=25, hash_num=3, filter_bytes=fbytes, mutate=False, n_runs=False)
        res = pol.polish_batch(None, job.offsets, job.lens, device_ptr=job.device_ptr, n=job.n_bytes)
        st = res.stats()
        assert st.bases == job.n_bytes
        assert (st.absent_kmers, st.events, st.substitutions, st.insertions, st.deletions) == (0, 0, 0, 0, 0)
        res.free()
        del job
        torch.cuda.empty_cache()
        # 2. the mutated draft of the same genome (the filter is already built)
        job = SyntheticJob(pol, total, k=25, hash_num=3, filter_bytes=fbytes, build_filter=False)
        st, host, names = _compare_every_contig(pol, job, tmp_path, "configs3", capsys)
        assert st.insertions > 0 and st.deletions > 0
        # the synthetic error rates: 0.1% substitutions, 0.01% indels -- nearly all are repaired
        assert 0.8e-3 * job.n_bases < st.substitutions < 1.2e-3 * job.n_bases
        # 3.
        _partition_independence(pol, job, host, names,
                                (st.absent_kmers, st.substitutions, st.insertions, st.deletions))
    finally:
        pol.close()


def test_full_size_nonpow2_filter_every_contig(tmp_path, oracle_build, capsys):
    """The reference tool's own filter size for a 3 Gbp genome (src/ntedit_make_genome_bf.cpp:41-47,131-137: --fpr 0.01, h = 3
    -> 12.37 bits per element = 4.64 GB, not a power of two): slots by the exact reciprocal (filter_slot), 554 slices of
    8 MiB probed in two parts of 4 MiB each (k_bin_probe) -- every contig of the 3 Gbp draft byte-identical to the oracle
    run with the same filter."""
    import ntedit_amd
    from ntedit_amd.synth import SyntheticJob

    total = float(os.environ.get("NTEDIT_FULL_BASES", "3e9"))
    fbytes = int(os.environ.get("NTEDIT_NONPOW2_FILTER", "4640000000"))
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_params(ntedit_amd.default_params())
        job = SyntheticJob(pol, total, k=25, hash_num=3, filter_bytes=fbytes)
        pol.reserve(job.n_bytes, len(job.lens), on_device=1)
        st, host, names = _compare_every_contig(pol, job, tmp_path, "nonpow2", capsys)
        assert st.screen_binned
        assert 0.8e-3 * job.n_bases < st.substitutions < 1.2e-3 * job.n_bases
        assert st.insertions > 0 and st.deletions > 0
    finally:
        pol.close()


def test_genome_like_3gbp_every_contig(tmp_path, oracle_build, capsys):
    """A draft that looks like an assembly (synth.GenomeStructure: ~3 % simple-sequence arrays, ~3 % satellite arrays, ~5 %
    dispersed repeat families, ~2 % segmental duplications, ~0.5 % stretches the filter does not hold) at BASELINE's
    headline size: the partitioned screening stays on (its overflow list takes the repeats' probes, no record chunk is
    lost), and EVERY contig is byte-identical to the oracle's output."""
    import ntedit_amd
    from ntedit_amd.synth import SyntheticJob

    total = float(os.environ.get("NTEDIT_FULL_BASES", "3e9"))
    fbytes = int(os.environ.get("NTEDIT_FULL_FILTER", str(1 << 32)))
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_params(ntedit_amd.default_params())
        job = SyntheticJob(pol, total, k=25, hash_num=3, filter_bytes=fbytes, structure="genome")
        pol.reserve(job.n_bytes, len(job.lens), on_device=1)
        st, host, names = _compare_every_contig(pol, job, tmp_path, "genome3g", capsys)
        with capsys.disabled():
            print("[genome3g] structure (bases): %s; overflow entries %d, record chunks re-screened directly %d of %d"
                  % (job.structure.bases, st.screen_overflow_records, st.screen_chunks_direct, st.screen_launches), flush=True)
        assert st.screen_binned and st.screen_chunks_direct == 0
        assert st.screen_overflow_records > 1_000_000  # (the simple-sequence arrays)
        assert st.substitutions > 0.5e-3 * job.n_bases and st.insertions > 0 and st.deletions > 0
    finally:
        pol.close()


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_genome_like_cases_at_real_run_sizes(tmp_path, oracle_build, capsys, seed):
    """three draws of tests/tools/fuzz_genome_like.py (72-160 Mbases, repeat flavours at random shares, the partitioned
    screening at its real run sizes -- no bin_cap_percent --, one draw forced to half a draft of simple sequence, where a record
    chunk's overflow list runs out and the direct kernel takes the chunk): every contig identical to the oracle's"""
    sys.path.insert(0, os.path.join(H.ROOT, "tests", "tools"))
    import fuzz_genome_like as F
    rng = np.random.default_rng(seed)
    c = F.draw(rng)
    if seed == 13:
        c["fractions"]["simple"] = 0.5
        c["filter_bytes"] = 1 << 30
        c["bases"] = 96e6
    ok, info = F.run_case(c, str(tmp_path))
    with capsys.disabled():
        print("\n[genome-like %d] %s: %s" % (seed, c, info), flush=True)
    assert ok
    if seed == 13:
        assert "screening partitioned" in info


def test_config2_250mbp_every_contig(tmp_path, oracle_build, capsys):
    """BASELINE.json configs[2]: synthetic 250 Mbp draft of 2,500 x 100 kbp contigs (0.1% mismatches + 0.01% indels),
    k=25, 4 GiB filter, one MI355X -- every contig byte-identical to the oracle."""
    import ntedit_amd
    from ntedit_amd.synth import SyntheticJob

    fbytes = int(os.environ.get("NTEDIT_FULL_FILTER", str(1 << 32)))
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_params(ntedit_amd.default_params())
        job = SyntheticJob(pol, 250e6, k=25, hash_num=3, filter_bytes=fbytes, contig_len=100_000, n_runs=False)
        assert len(job.lens) == 2500
        st, host, names = _compare_every_contig(pol, job, tmp_path, "configs2", capsys)
        assert 0.8e-3 * job.n_bases < st.substitutions < 1.2e-3 * job.n_bases
        assert st.insertions > 0 and st.deletions > 0
    finally:
        pol.close()


def test_snv_mode_250mbp_every_contig(tmp_path, oracle_build, capsys):
    """SURVEY 8f-2 at size: -s 1 on a 250 Mbp draft (2,500 x 100 kbp, 512 MiB filter): every position of every
    contig is re-assessed (no screen gate), _edited.fa / _changes.tsv / VCF body byte-identical to the oracle
    (ntedit.cpp:1806,1865,1890-1914)."""
    import ntedit_amd
    from ntedit_amd.synth import SyntheticJob

    total = float(os.environ.get("NTEDIT_SNV_BASES", "250e6"))
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_params(ntedit_amd.default_params(snv=1, max_insertions=0, max_deletions=0))
        job = SyntheticJob(pol, total, k=25, hash_num=3, filter_bytes=1 << 29, contig_len=100_000, n_runs=False)
        st, host, names = _compare_every_contig(pol, job, tmp_path, "snv250", capsys, snv=1, max_insertions=0, max_deletions=0)
        assert st.substitutions > 0.5e-3 * job.n_bases
    finally:
        pol.close()


def test_counting_filter_250mbp_every_contig(tmp_path, oracle_build, capsys):
    """SURVEY 8f-1 at size: a 4 GiB counting filter (2^32 8-bit counters, synthetic contents: every truth k-mer has
    counters 1..4), -p 2 (k-mers seen once count as absent), 250 Mbp draft: the binned screening on counters and the
    event machine's median coverage, every contig byte-identical to the oracle (ntedit.cpp:357-361,373-376,455-463,1806)."""
    import ntedit_amd
    from ntedit_amd.synth import SyntheticJob, counting_filter_from_plain

    total = float(os.environ.get("NTEDIT_CBF_BASES", "250e6"))
    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_params(ntedit_amd.default_params())
        job = SyntheticJob(pol, total, k=25, hash_num=3, filter_bytes=1 << 29, contig_len=100_000, n_runs=False)
        counters = counting_filter_from_plain(pol, 25, 3)
        pol.set_params(ntedit_amd.default_params(min_threshold=2))
        st, host, names = _compare_every_contig(pol, job, tmp_path, "cbf250", capsys, counting=True, min_threshold=2)
        assert st.screen_binned
        assert st.substitutions > 1e-4 * job.n_bases  # (a quarter of the truth k-mers have count 1 < -p 2: many positions cannot be fixed)
        del counters
    finally:
        pol.close()


def test_make_genome_bf_cli(tmp_path, oracle_build):
    """ntedit-make-genome-bf: command line and console lines of the reference tool
    (src/ntedit_make_genome_bf.cpp), filter file byte-identical to the oracle's mkbf"""
    import math
    import subprocess
    rng = np.random.default_rng(21)
    g1 = [(b"a x", H.random_genome(rng, 120000)), (b"tiny", b"ACGTACGT"), (b"b", H.random_genome(rng, 3000).lower())]
    g2 = [(b"c", H.random_genome(rng, 40000)[:20000] + b"NNNNNNNNNN" + H.random_genome(rng, 20000))]
    H.write_fasta(str(tmp_path / "g1.fa"), g1, width=60)
    H.write_fasta(str(tmp_path / "g2.fa"), g2)
    tool = os.path.join(H.ROOT, "ntedit_amd", "ntedit-make-genome-bf")
    # explicit size
    out = str(tmp_path / "t.bf")
    r = subprocess.run([tool, "--genome", str(tmp_path / "g1.fa"), str(tmp_path / "g2.fa"), "-k", "25", "--hashes", "4",
                        "--bf", "100003", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "BF size (bytes): 100003" in r.stdout
    H.mkbf([str(tmp_path / "g1.fa"), str(tmp_path / "g2.fa")], str(tmp_path / "o.bf"), k=25, hashes=4, nbytes=100003)
    got, want = H.load_bf(out), H.load_bf(str(tmp_path / "o.bf"))
    assert got["k"] == 25 and got["hash_num"] == 4
    assert np.array_equal(got["data"], want["data"])
    assert open(out, "rb").read() == open(str(tmp_path / "o.bf"), "rb").read()
    occ = int(np.unpackbits(want["data"]).sum())
    fpr = (occ / (want["data"].size * 8)) ** 4
    line = [l for l in r.stdout.splitlines() if l.startswith("Bloom filter FPR: ")][0]
    assert abs(float(line.split(": ")[1]) - fpr) <= 1e-5 * fpr + 1e-12
    # size from the genome length (ntedit_make_genome_bf.cpp:41-47,131-135)
    r = subprocess.run([tool, "--genome", str(tmp_path / "g1.fa"), "-k", "31", "-o", out], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    n = sum(len(s) for _, s in g1)
    assert "Genome size (bp): %d" % n in r.stdout
    rr = -3.0 / math.log(1.0 - math.exp(math.log(0.01) / 3.0))
    size = int(math.ceil(n * rr) / 8)
    assert "BF size (bytes): %d" % size in r.stdout
    H.mkbf([str(tmp_path / "g1.fa")], str(tmp_path / "o.bf"), k=31, hashes=3, nbytes=size)
    assert np.array_equal(H.load_bf(out)["data"], H.load_bf(str(tmp_path / "o.bf"))["data"])
    # usage errors
    assert subprocess.run([tool, "-k", "25"], capture_output=True).returncode == 1
    assert subprocess.run([tool, "--genome", str(tmp_path / "g1.fa")], capture_output=True).returncode == 1


@pytest.mark.parametrize("kw", [dict(), dict(max_deletions=10, max_insertions=5), dict(mode=1), dict(snv=1), dict(k=40),
                                dict(jump=1, max_deletions=8)])
def test_errors_in_front_of_contig_ends_gpu(tmp_path, oracle_build, kw):
    """the GPU twin of test_hostsim_parity.test_errors_in_front_of_contig_ends: errors at every distance from a contig's
    end, where the character window is cut short (round 6; until then those positions went through the general
    rope-walking paths, a sweep of 341 candidates probed one k-mer after the other)"""
    kw = dict(kw)
    k = kw.pop("k", 25)
    case = H.make_contig_end_case(str(tmp_path), k=k)
    hp = H.default_params(min_contig_len=0, **kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"))
    for general in (False, True):
        pol = _fresh(general=general)
        try:
            _load_filters(pol, case)
            pol.set_params(_hip_params(min_contig_len=0, **kw))
            pol.polish_records(H.read_fasta(case["draft"]), str(tmp_path / "g"))
        finally:
            pol.close()
        for suf in ("_changes.tsv", "_edited.fa"):
            assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("g" + suf)), shallow=False), suf
        assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "g_variants.vcf"))


@pytest.mark.parametrize("kw", [dict(snv=1), dict(snv=1, max_insertions=0, max_deletions=0), dict(snv=1, mode=1),
                                dict(snv=1, jump=1), dict(snv=1, k=40)])
def test_snv_candidate_map(tmp_path, oracle_build, kw):
    """-s 1 on a plain filter (round 6): the first probes of every position's substitution candidates go through the
    partitioned pipeline (the candidate map, nte_bin_wc.inc MODE 1) and k_assess probes on from the second hash --
    forced here on small inputs ("candmap" 1), in several record chunks, with an overflow list that runs out (those
    chunks' part of the map becomes "unknown") and on a filter whose size is not a power of two: same bytes as the
    oracle, and as the run without the map"""
    kw = dict(kw)
    k = kw.pop("k", 25)
    for flavor, bfbytes in (("N lower iupac", 1 << 22), ("N rep", 3_000_001)):
        sub = tmp_path / ("c%d" % bfbytes)
        case = H.make_case(str(sub), 8100 + k, n=150000, contigs=3, k=k, flavor=flavor, bfbytes=bfbytes)
        hp = H.default_params(**kw)
        H.run_oracle(case["draft"], case["bf"], hp, str(sub / "o"))
        recs = H.read_fasta(case["draft"])
        for tag, tunes in (("map", dict(candmap=1)), ("chunks", dict(candmap=1, bin_chunk=3 * 16384)),
                           ("lost", dict(candmap=1, bin_chunk=3 * 16384, bin_cap_percent=5, bin_ovf_cap=4096)),
                           ("nomap", dict(candmap=0))):
            pol = _fresh()
            try:
                _load_filters(pol, case)
                pol.set_params(_hip_params(**kw))
                for key, val in tunes.items():
                    pol.set_tuning(key, val)
                st = pol.polish_records(recs, str(sub / tag))
            finally:
                pol.close()
            for suf in ("_changes.tsv", "_edited.fa"):
                assert filecmp.cmp(str(sub / ("o" + suf)), str(sub / (tag + suf)), shallow=False), (tag, suf)
            assert H.vcf_body(str(sub / "o_variants.vcf")) == H.vcf_body(str(sub / (tag + "_variants.vcf"))), tag
            if tag == "lost":
                assert st.screen_chunks_direct > 0


def test_parked_events_resolve_quickly(tmp_path, oracle_build):
    """Nearly every event parked by a tiny budget on a draft where hardly a k-mer is in the filter (the regime of fuzz seed
    42424200091: k=128, 1.7 % errors, budget 8; a third of its length here).  What the serial order applies there is ONE
    event per contig that never gets back to a clean state -- the reference's serial program itself, run by one
    wavefront -- so the time is that chain's, not the number of re-run rounds (1; PolishRun::collect widens the re-runs
    after 6).  Modes 1 / 2 evaluate a sweep's candidates across the lanes (try_indels_all), the overlay of changed draft
    characters is pruned behind the head cursor as it grows: the full-length case took 8 minutes and takes 13 s / 40 s
    (-m 0 / -m 1 -a 1); byte-identical."""
    import time
    case_kw = {'n': 14000, 'contigs': 3, 'k': 128, 'hashes': 2, 'p_sub': 0.01, 'p_ins': 0.005, 'p_del': 0.002,
               'flavor': 'lower sec', 'bfbytes': 131072}
    base = {'mode': 1, 'mask': 1, 'jump': 2, 'max_insertions': 5, 'max_deletions': 5, 'min_contig_len': 0,
            'missing_threshold': 9.0, 'edit_threshold': 25.0, 'start_grid': 16, 'event_budget': 8}
    case = H.make_case(str(tmp_path), 42424200091, **case_kw)
    for par_kw in (base, dict(base, mode=0, mask=0)):
        hp = H.default_params(**par_kw)
        H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"), case["rep"])
        pol = _fresh(force_rounds=False)
        try:
            _load_filters(pol, case)
            pol.set_params(_hip_params(**par_kw))
            t0 = time.time()
            pol.polish_records(H.read_fasta(case["draft"]), str(tmp_path / "g"))
            dt = time.time() - t0
        finally:
            pol.close()
        for suf in ("_changes.tsv", "_edited.fa"):
            assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("g" + suf)), shallow=False), (suf, par_kw)
        assert dt < 20.0, "the serial chains took %.1f s (%r)" % (dt, par_kw)


@pytest.mark.parametrize("kw", [dict(snv=1, mask=1), dict(mask=1), dict(snv=1, mode=2)])
def test_last_kmer_is_never_a_seed_gpu(tmp_path, oracle_build, kw):
    case = H.make_tail_case(str(tmp_path))
    hp = H.default_params(min_contig_len=0, **kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"))
    pol = _fresh()
    try:
        pol.set_params(_hip_params(min_contig_len=0, **kw))
        pol.load_filter_file(case["bf"], 0)
        pol.polish_records(H.read_fasta(case["draft"]), str(tmp_path / "h"))
    finally:
        pol.close()
    for suf in ("_changes.tsv", "_edited.fa"):
        assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("h" + suf)), shallow=False), suf
    assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "h_variants.vcf"))


def test_full_size_deep_search_secondary(tmp_path, oracle_build, capsys):
    """BASELINE.json configs[4] at full size: 3 Gbp draft, k=35, primary + secondary ("repeat", -e) 4 GiB
    filters, -i 5 -d 9 (after the reference's clamp of -i 9).  EVERY contig byte-identical to the oracle run on the
    whole batch with both filters, plus partition independence of the edit counts."""
    import ntedit_amd
    from ntedit_amd.synth import SyntheticJob

    total = float(os.environ.get("NTEDIT_FULL_BASES", "3e9"))
    fbytes = int(os.environ.get("NTEDIT_FULL_FILTER", str(1 << 32)))
    kw = dict(max_insertions=5, max_deletions=9)
    pol = ntedit_amd.Polisher(0)
    try:
        job = SyntheticJob(pol, total, k=35, hash_num=3, filter_bytes=fbytes, rep_filter_bytes=fbytes)
        pol.set_params(ntedit_amd.default_params(**kw))
        res = pol.polish_batch(None, job.offsets, job.lens, device_ptr=job.device_ptr, n=job.n_bytes)
        res.free()  # (warm-up: buffers)
        st, host, names = _compare_every_contig(pol, job, tmp_path, "configs4", capsys, rep=True, **kw)
        assert st.substitutions > 0 and st.insertions > 0 and st.deletions > 0
        _partition_independence(pol, job, host, names,
                                (st.absent_kmers, st.substitutions, st.insertions, st.deletions))
    finally:
        pol.close()


def test_cli_edge_inputs(tmp_path, oracle_build):
    """`ntedit` on inputs at the edges: empty file, only contigs below -z, a contig shorter than k, FASTQ,
    CRLF line ends, blank lines -- same files as the oracle's command line (which reads with its own reader)"""
    import subprocess
    cli = os.path.join(H.ROOT, "ntedit_amd", "ntedit")
    rng = np.random.default_rng(17)
    truth = H.random_genome(rng, 30000)
    H.write_fasta(str(tmp_path / "truth.fa"), [(b"t", truth)])
    H.mkbf([str(tmp_path / "truth.fa")], str(tmp_path / "t.bf"), k=25, hashes=3, nbytes=1 << 16)
    d1 = H.mutate(rng, truth[1000:9000], 3e-3, 3e-4, 3e-4)
    d2 = H.mutate(rng, truth[12000:12600], 3e-3, 0, 0)
    inputs = {
        "empty.fa": b"",
        "short_only.fa": b">a\nACGTACGTACGT\n>b x\nACGT\n",
        "below_k.fa": b">tiny\nACGTACGTACGTACGTACGT\n>ok\n" + d2 + b"\n",
        "reads.fq": b"@r1 first\n" + d1 + b"\n+\n" + b"I" * len(d1) + b"\n@r2\n" + d2 + b"\n+r2\n" + b"#" * len(d2) + b"\n",
        "crlf.fa": b">c1 windows\r\n" + b"\r\n".join(d1[i:i + 60] for i in range(0, len(d1), 60)) + b"\r\n>c2\r\n" + d2 + b"\r\n",
        "blank_lines.fa": b"\n\n>c1\n" + d1[:4000] + b"\n\n" + d1[4000:] + b"\n\n\n>c2\n\n" + d2 + b"\n",
    }
    for name, data in inputs.items():
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            f.write(data)
        for z in ("0", "100"):
            o = subprocess.run([os.path.join(H.ORACLE_BUILD, "ntedit_oracle"), "-f", path, "-r", str(tmp_path / "t.bf"), "-b",
                                str(tmp_path / "o"), "-z", z], capture_output=True, text=True)
            g = subprocess.run([cli, "-f", path, "-r", str(tmp_path / "t.bf"), "-b", str(tmp_path / "g"), "-z", z],
                               capture_output=True, text=True)
            assert o.returncode == 0 and g.returncode == 0, (name, z, o.stderr, g.stderr)
            for suf in ("_changes.tsv", "_edited.fa"):
                assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("g" + suf)), shallow=False), (name, z, suf)
            assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / "g_variants.vcf")), (name, z)


def test_cli_mapped_reader_equals_streaming(tmp_path, oracle_build):
    """plain multi-FASTA drafts are ingested by the mapped, multi-threaded reader (several batches, -t threads);
    the outputs are the streaming reader's (--no-map) and the oracle's, byte for byte"""
    import subprocess
    cli = os.path.join(H.ROOT, "ntedit_amd", "ntedit")
    case = H.make_many_case(str(tmp_path), n_contigs=1500, mean_len=700, seed=13)
    hp = H.default_params(min_contig_len=200)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"))
    for tag, extra in (("m", ["-t", "5"]), ("s", ["--no-map"]), ("m1", ["-t", "1", "--batch-bases", "100000"])):
        r = subprocess.run([cli, "-f", case["draft"], "-r", case["bf"], "-b", str(tmp_path / tag), "-z", "200",
                            "--batch-bases", "150000"] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for suf in ("_changes.tsv", "_edited.fa"):
            assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / (tag + suf)), shallow=False), (tag, suf)
        assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / (tag + "_variants.vcf"))), tag
    # the same draft bgzip-compressed (inflated member by member on all threads), and as one ordinary gzip stream
    from test_fasta_reader import bgzf_bytes
    import gzip
    data = open(case["draft"], "rb").read()
    for tag, blob in (("bg", bgzf_bytes(data)), ("gz", gzip.compress(data, 1))):
        path = str(tmp_path / (tag + ".fa.gz"))
        with open(path, "wb") as f:
            f.write(blob)
        r = subprocess.run([cli, "-f", path, "-r", case["bf"], "-b", str(tmp_path / tag), "-z", "200", "--batch-bases", "150000"],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for suf in ("_changes.tsv", "_edited.fa"):
            assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / (tag + suf)), shallow=False), (tag, suf)


def test_reserve_first_call_is_warm(tmp_path, oracle_build):
    """ntedit_hip_reserve: a context's FIRST batch costs what a warm one does (the buffers, page-locked result memory,
    kernel code and scratch are in place), its result is the one an unreserved context computes, and the screening's
    event window holds no allocation time even without it (round 4's 184 / 553 ms screening outliers were a fresh 7 GB
    record buffer being mapped inside that window: DESIGN.md 8)."""
    import time
    import ntedit_amd
    from ntedit_amd.synth import SyntheticJob

    def run(pol, job):
        t0 = time.perf_counter()
        res = pol.polish_batch(None, job.offsets, job.lens, device_ptr=job.device_ptr, n=job.n_bytes)
        st = res.stats()
        res.free()
        wall = (time.perf_counter() - t0) * 1e3
        return wall, st

    def key(st):
        return (st.absent_kmers, st.events, st.events_deferred)

    pol = ntedit_amd.Polisher(0)
    try:
        pol.set_params(ntedit_amd.default_params())
        job = SyntheticJob(pol, 128e6, k=25, hash_num=3, filter_bytes=1 << 30, contig_len=100_000, n_runs=False)
        cold_wall, cold = run(pol, job)
        assert cold.screen_binned
        warm = min((run(pol, job) for _ in range(3)), key=lambda x: x[0])
        bits = pol.filter_device_ptr(0)
    finally:
        pass
    pol2 = ntedit_amd.Polisher(0)
    try:
        pol2.set_filter_device(bits, 1 << 30, 3, 25)
        pol2.set_params(ntedit_amd.default_params())
        pol2.reserve(job.n_bytes, len(job.lens), on_device=1)
        first_wall, first = run(pol2, job)
        again_wall, again = run(pol2, job)
    finally:
        pol2.close()
        pol.close()
    print("\n[reserve] 128 Mbp: unreserved first call %.1f ms (screen window %.1f), warm %.1f ms (screen %.1f); reserved first call "
          "%.1f ms (screen %.1f), second %.1f ms" % (cold_wall, cold.ms_screen, warm[0], warm[1].ms_screen, first_wall,
                                                      first.ms_screen, again_wall), flush=True)
    assert key(first) == key(cold) == key(warm[1]) == key(again)
    # the reserved context's first call: within 25 % (+ 1 ms) of a warm call
    assert first_wall <= 1.25 * warm[0] + 1.0, (first_wall, warm[0])
    # a screening of this size must take what it takes -- never a multiple (allocations are outside its window)
    assert cold.ms_screen <= 3.0 * warm[1].ms_screen, (cold.ms_screen, warm[1].ms_screen)
    assert first.ms_screen <= 1.5 * warm[1].ms_screen + 0.5


def test_scaffold_gap_is_not_walked(tmp_path, oracle_build):
    """A scaffold gap of eight million Ns with errors on both sides.  An event that ends in front of the gap used to roll
    through it base by base -- one lane, a microsecond per base, while the launch waits -- because that is what the
    reference's loop does (ntedit.cpp:2119-2138, a millisecond on a CPU).  Now it stops inside the gap as soon as its state
    is clean: nothing in there is in the absent bitmap, and where the serial program comes out behind the gap another
    event starts.  Same bytes as the oracle, and the polish call takes milliseconds."""
    import time
    from test_hostsim_parity import _gap_case
    draft, bf = _gap_case(str(tmp_path), 8_000_000)
    recs = H.read_fasta(draft)
    for kw in (dict(), dict(snv=1)):
        hp = H.default_params(**kw)
        H.run_oracle(draft, bf, hp, str(tmp_path / "o"))
        pol = _fresh(force_rounds=False)
        try:
            pol.load_filter_file(bf)
            pol.set_params(_hip_params(**kw))
            blob, offs, lens, names = H.pack_batch(recs)
            pol.polish_batch(blob, offs, lens).free()  # (warm-up: buffers, kernel code)
            t0 = time.perf_counter()
            res = pol.polish_batch(blob, offs, lens)
            ms = (time.perf_counter() - t0) * 1e3
            st = res.stats()
            res.free()
            pol.polish_records(recs, str(tmp_path / "g"))
        finally:
            pol.close()
        for suf in ("_changes.tsv", "_edited.fa"):
            assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("g" + suf)), shallow=False), (suf, kw)
        print("\n[gap] 8 Mbp of N, %r: polish call %.1f ms (machine %.2f ms)" % (kw, ms, st.ms_machine), flush=True)
        assert st.ms_machine < 150.0, st.ms_machine


def test_reserve_with_hostile_parameters(tmp_path, oracle_build):
    """ntedit_hip_reserve runs an internal batch of random bases through whatever the caller configured.  With -m 2, -a 1,
    an event start at every absent position and a tiny event budget (fuzz seed 42424200069: reserve used to fail with
    "event machine ran out of room" there -- a megabase of unknown draft was 10 GB of mask records) it must neither fail
    nor crawl, and the batch polished after it must be the oracle's."""
    import time
    case_kw = {'n': 11612, 'contigs': 3, 'k': 25, 'hashes': 1, 'p_sub': 0.002, 'p_ins': 0.005, 'p_del': 0.0003,
               'flavor': 'iupac rep sec', 'bfbytes': 174386}
    par_kw = {'mode': 2, 'mask': 1, 'jump': 1, 'max_insertions': 0, 'max_deletions': 1, 'min_contig_len': 0,
              'missing_threshold': 25.0, 'edit_threshold': 25.0, 'start_grid': 1, 'event_budget': 600}
    case = H.make_case(str(tmp_path), 42424200069, **case_kw)
    hp = H.default_params(**par_kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"), case["rep"])
    pol = _fresh(force_rounds=False)
    try:
        _load_filters(pol, case)
        pol.set_params(_hip_params(**par_kw))
        t0 = time.time()
        pol.reserve(64 << 20, 1 << 16, on_device=2)
        dt = time.time() - t0
        pol.polish_records(H.read_fasta(case["draft"]), str(tmp_path / "g"))
    finally:
        pol.close()
    for suf in ("_changes.tsv", "_edited.fa"):
        assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("g" + suf)), shallow=False), suf
    assert dt < 20.0, "reserve took %.1f s" % dt


@pytest.mark.parametrize("kw", H.SWEEP_RICH_PARAMS)
def test_sweep_rich_counting_filter(tmp_path, oracle_build, kw):
    """Counting filter, -p 2 cutting into the k-mers that are there, an error-rich draft: runs in which position after
    position needs an indel sweep, behind substitutions as well (bench.py --counting in small; a tenth of the events of
    such a batch are most of its machine time).  All events in one round and in rounds; same bytes as the oracle."""
    case = H.make_sweep_rich_case(str(tmp_path))
    recs = H.read_fasta(case["draft"])
    hp = H.default_params(min_threshold=2, **kw)
    H.run_oracle(case["draft"], case["bf"], hp, str(tmp_path / "o"))
    for rounds in (False, True):
        pol = _fresh(force_rounds=rounds)
        try:
            pol.load_filter_file(case["bf"])
            pol.set_params(_hip_params(min_threshold=2, **kw))
            pol.polish_records(recs, str(tmp_path / ("g%d" % rounds)))
        finally:
            pol.close()
        for suf in ("_changes.tsv", "_edited.fa"):
            assert filecmp.cmp(str(tmp_path / ("o" + suf)), str(tmp_path / ("g%d%s" % (rounds, suf))), shallow=False), (suf, kw, rounds)
        assert H.vcf_body(str(tmp_path / "o_variants.vcf")) == H.vcf_body(str(tmp_path / ("g%d_variants.vcf" % rounds)))
