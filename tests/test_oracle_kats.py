"""Known-answer / self-consistency tests of the CPU oracle's primitives
(roll == re-seed, canonical(kmer) == canonical(revcomp(kmer)),
changelast == re-seed) and of the product's host+device hashing header against
the oracle (via the test-only host build)."""
import ctypes
import os

import numpy as np
import pytest

import helpers as H

COMP = bytes.maketrans(b"ACGTacgt", b"TGCAtgca")


def revcomp(s):
    return s.translate(COMP)[::-1]


def canon(lib, s, k):
    return (lib.ora_base_forward_hash(s, k) + lib.ora_base_reverse_hash(s, k)) & (2**64 - 1)


def test_split_rotate_is_a_bijection_with_period_1023(oracle_build):
    lib = H.oracle_lib()
    rng = np.random.default_rng(1)
    for x in [int(v) for v in rng.integers(0, 2**63, size=20, dtype=np.int64)] + [1, 2**32, 2**33, 2**63, 2**64 - 1]:
        assert lib.ora_sror(lib.ora_srol(x)) == x
        assert lib.ora_srol_n(x, 1) == lib.ora_srol(x)
        assert lib.ora_srol_n(x, 33 * 31) == x
        y = x
        for _ in range(7):
            y = lib.ora_srol(y)
        assert lib.ora_srol_n(x, 7) == y
        # the two halves never mix: bit count of each half is invariant
        lo, hi = x & (2**33 - 1), x >> 33
        y = lib.ora_srol_n(x, 11)
        assert bin(y & (2**33 - 1)).count("1") == bin(lo).count("1")
        assert bin(y >> 33).count("1") == bin(hi).count("1")


@pytest.mark.parametrize("k", [12, 25, 31, 32, 33, 40, 64, 100])
def test_roll_equals_reseed_and_strand_symmetry(k, oracle_build):
    lib = H.oracle_lib()
    rng = np.random.default_rng(k)
    s = H.random_genome(rng, 400)
    fh = lib.ora_base_forward_hash(s[:k], k)
    rh = lib.ora_base_reverse_hash(s[:k], k)
    for i in range(1, 400 - k):
        fh = lib.ora_next_forward_hash(fh, k, s[i - 1], s[i + k - 1])
        rh = lib.ora_next_reverse_hash(rh, k, s[i - 1], s[i + k - 1])
        km = s[i:i + k]
        assert fh == lib.ora_base_forward_hash(km, k)
        assert rh == lib.ora_base_reverse_hash(km, k)
        # forward hash of the reverse complement is the reverse hash, so fh+rh is strand-neutral
        assert lib.ora_base_forward_hash(revcomp(km), k) == rh
        assert canon(lib, km, k) == canon(lib, revcomp(km), k)
    # case-insensitive
    assert canon(lib, s[:k].lower(), k) == canon(lib, s[:k], k)


def test_published_seed_constants(oracle_build):
    """ntHash2's four seeds and the multi-hash constants (the only hard numbers
    this path has; btllib itself is unavailable => 'parity unpinned')."""
    lib = H.oracle_lib()
    lib.ora_seed.restype = ctypes.c_uint64
    lib.ora_seed.argtypes = [ctypes.c_ubyte]
    assert lib.ora_seed(ord("A")) == 0x3C8BFBB395C60474
    assert lib.ora_seed(ord("C")) == 0x3193C18562A02B4C
    assert lib.ora_seed(ord("G")) == 0x20323ED082572324
    assert lib.ora_seed(ord("T")) == 0x295549F54BE24456
    assert lib.ora_seed(ord("N")) == 0
    for c, comp in zip(b"ACGT", b"TGCA"):
        assert lib.ora_seed(c & 7) == lib.ora_seed(comp)
    hv = (ctypes.c_uint64 * 3)()
    lib.ora_extend_hashes.argtypes = [ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint64)]
    lib.ora_extend_hashes(0x0123456789ABCDEF, 25, 3, hv)
    base = 0x0123456789ABCDEF
    for i in (1, 2):
        t = (base * (i ^ ((25 * 0x90B45D39FB6DA1FA) & (2**64 - 1)))) & (2**64 - 1)
        assert hv[i] == t ^ (t >> 27)
    assert hv[0] == base


def test_filter_roundtrip_and_membership(tmp_path, oracle_build):
    rng = np.random.default_rng(3)
    g = H.random_genome(rng, 20000)
    H.write_fasta(str(tmp_path / "g.fa"), [(b"g", g)], width=60)
    for nbytes in (1 << 16, 100003 * 8):
        H.mkbf([str(tmp_path / "g.fa")], str(tmp_path / "g.bf"), k=25, hashes=3, nbytes=nbytes)
        bf = H.load_bf(str(tmp_path / "g.bf"))
        assert bf["k"] == 25 and bf["hash_num"] == 3 and bf["bytes"] == nbytes and not bf["counting"]
        # every k-mer of g is present; both strands
        bm = H.oracle_screen(g, bf)
        assert not bm.any()
        bm = H.oracle_screen(revcomp(g), bf)
        assert not bm.any()
        # a random sequence is mostly absent
        other = H.random_genome(rng, 20000)
        frac = np.unpackbits(H.oracle_screen(other, bf).view(np.uint8)).sum() / (20000 - 24)
        assert frac > 0.8


def test_host_device_header_matches_oracle_screen(tmp_path, oracle_build):
    """nte_common.h (the product's hashing + probe arithmetic, host build) == oracle,
    including IUPAC / N / lowercase / separators and a non-power-of-two filter."""
    sim = H.hostsim_lib()
    rng = np.random.default_rng(5)
    g = H.random_genome(rng, 30000)
    H.write_fasta(str(tmp_path / "g.fa"), [(b"g", g)])
    for k, h, nbytes in ((25, 3, 1 << 15), (40, 4, 50021 * 8), (13, 1, 1 << 12)):
        H.mkbf([str(tmp_path / "g.fa")], str(tmp_path / "g.bf"), k=k, hashes=h, nbytes=nbytes)
        bf = H.load_bf(str(tmp_path / "g.bf"))
        d = bytearray(H.mutate(rng, g, 5e-3, 1e-3, 1e-3))
        d[1000:1030] = b"N" * 30
        d[2000] = ord("\n")
        for i, c in enumerate(b"RYSWKMBDHVUXn-*"):
            d[3000 + 40 * i] = c
        d[5000:5200] = bytes(d[5000:5200]).lower()
        blob = bytes(d)
        want = H.oracle_screen(blob, bf)
        got = np.zeros_like(want)
        rc = sim.hostsim_screen(ctypes.c_char_p(blob), ctypes.c_uint64(len(blob)),
                                bf["data"].ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(bf["bytes"]),
                                ctypes.c_uint32(h), ctypes.c_uint32(k), got.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(0), ctypes.c_uint32(1))
        assert rc == 0
        assert np.array_equal(got, want)


def test_btllib_unit_test_vector(oracle_build):
    """btllib's own ntHash unit test (btllib tests/nthash.cpp): the three 5-mers of "ACATGCATGCA" with 3
    hashes each.  btllib is not part of /root/reference, so these nine words are quoted from its published
    test suite, not taken from a file here -- the hashing therefore stays formally "parity unpinned" --
    but nine matching 64-bit words cannot be a coincidence: seeds, split rotation, canonical = fh + rh and
    the multi-hash extension (i ^ k * MULTISEED, >> 27) are the ones btllib computes."""
    lib = H.oracle_lib()
    lib.ora_extend_hashes.argtypes = [ctypes.c_uint64, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint64)]
    seq, k = b"ACATGCATGCA", 5
    want = [
        (0xF59ECB45F0E22B9C, 0x4969C33AC240C129, 0x688D616F0D7E08C3),
        (0x38CC00F940AEBDAE, 0xAB7E1B110E086FC6, 0x011A1818BCFDD553),
        (0x603A48C5A11C794A, 0xE66016E61816B9C4, 0xC5B13CB146996FFE),
    ]
    fh = lib.ora_base_forward_hash(seq[:k], k)
    rh = lib.ora_base_reverse_hash(seq[:k], k)
    for i, w in enumerate(want):
        if i:
            # by rolling, the way the hot path gets there
            fh = lib.ora_next_forward_hash(fh, k, seq[i - 1], seq[i + k - 1])
            rh = lib.ora_next_reverse_hash(rh, k, seq[i - 1], seq[i + k - 1])
            assert fh == lib.ora_base_forward_hash(seq[i:i + k], k)
        hv = (ctypes.c_uint64 * 3)()
        lib.ora_extend_hashes((fh + rh) & (2**64 - 1), k, 3, hv)
        assert tuple(hv) == w


def test_btllib_kit_our_side_matches_the_oracle(tmp_path, oracle_build):
    """tests/tools/verify_btllib.sh compares real btllib with dump_ours (the product's nte_common.h + bfio.cpp on the
    host).  btllib is not in this image, so here only the kit's own half is checked: dump_ours agrees with the oracle on
    the kit's fixed inputs (hashes by seeding and rolling, the built filter file, contains())."""
    import ctypes
    import subprocess
    import sys
    kit = os.path.join(H.ROOT, "tests", "tools", "btllib_kit")
    w = str(tmp_path)
    subprocess.run(["g++", "-O2", "-std=c++17", "-o", os.path.join(w, "dump_ours"), os.path.join(kit, "dump_ours.cpp"),
                    os.path.join(H.ROOT, "ntedit_amd", "host", "bfio.cpp"), os.path.join(H.ROOT, "ntedit_amd", "host", "params.cpp")],
                   check=True)
    subprocess.run([sys.executable, os.path.join(kit, "make_inputs.py"), w], check=True)
    lib = H.oracle_lib()
    for k in (25, 33, 64):
        out = subprocess.run([os.path.join(w, "dump_ours"), "hashes", os.path.join(w, "acgt.txt"), str(k), "3"],
                             capture_output=True, text=True, check=True).stdout.splitlines()
        lines = [l for l in open(os.path.join(w, "acgt.txt")).read().splitlines() if len(l) >= k]
        it = iter(out)
        n = 0
        for s in lines:
            b = s.encode()
            for i in range(len(s) - k + 1):
                f = next(it).split()
                fh = lib.ora_base_forward_hash(b[i:i + k], k)
                rh = lib.ora_base_reverse_hash(b[i:i + k], k)
                assert int(f[2], 16) == fh and int(f[3], 16) == rh, (k, i)
                assert int(f[5], 16) == fh and int(f[6], 16) == rh, (k, i)  # rolled == seeded
                hv = (ctypes.c_uint64 * 3)()
                lib.ora_extend_hashes(ctypes.c_uint64((fh + rh) & (2 ** 64 - 1)), k, 3, hv)
                assert [int(x, 16) for x in f[8:11]] == list(hv)
                n += 1
            assert next(it) == "--"
        assert n > 800
    # the filter the kit builds = the oracle's mkbf (array bytes), and its contains() = the oracle's screening
    subprocess.run([os.path.join(w, "dump_ours"), "build", os.path.join(w, "genome.fa"), "25", "3", "1048576",
                    os.path.join(w, "our.bf")], check=True, capture_output=True)
    H.mkbf([os.path.join(w, "genome.fa")], os.path.join(w, "ora.bf"), k=25, hashes=3, nbytes=1048576)
    assert subprocess.run([sys.executable, os.path.join(kit, "cmp_bf.py"), os.path.join(w, "our.bf"), os.path.join(w, "ora.bf")],
                          capture_output=True).returncode == 0
    q = subprocess.run([os.path.join(w, "dump_ours"), "query", os.path.join(w, "ora.bf"), os.path.join(w, "draft.txt")],
                       capture_output=True, text=True, check=True).stdout.splitlines()
    bf = H.load_bf(os.path.join(w, "ora.bf"))
    qi = iter(q[1:])
    for s in open(os.path.join(w, "draft.txt")).read().splitlines():
        absent = np.unpackbits(H.oracle_screen(s.encode(), bf).view(np.uint8), bitorder="little")
        for i in range(len(s) - 25 + 1):
            got = int(next(qi))
            if "N" not in s[i:i + 25]:  # (the screening bitmap only speaks about k-mers of accepted bases)
                assert got == 1 - int(absent[i]), i
        assert next(qi) == "--"


def test_filter_slot_is_the_modulo(oracle_build):
    """nte::filter_slot -- the device's `hash % slots` without a division (nte_common.h): the general reciprocal form and the
    short form for filters of 2^32 .. 2^40 slots -- against Python's % on random and adversarial hash values, for sizes on
    both sides of every boundary (btllib: `hash % array_bits`, ntedit.cpp:368-371)."""
    import ctypes
    lib = ctypes.CDLL(H.build_hostsim())
    rng = np.random.default_rng(5)
    sizes = [1000003, (1 << 32) - 5, (1 << 32) + 1, (1 << 32) + 2, 37_120_000_000, 4_640_000_000 * 8, (1 << 36) + 12345,
             (1 << 40) - 1, (1 << 40) + 7, 3 * (1 << 33), (1 << 35), 100000007 * 8, 2 ** 41 + 11]
    for m in sizes:
        hv = rng.integers(0, 1 << 64, 200000, dtype=np.uint64)
        edge = [0, 1, m - 1, m, m + 1, 2 * m - 1, 2 * m, 3 * m - 1, (1 << 64) - 1, (1 << 64) - m, ((1 << 64) // m) * m - 1,
                ((1 << 64) // m) * m, (1 << 32) - 1, 1 << 32, (1 << 63) + m]
        hv[:len(edge)] = np.array([e % (1 << 64) for e in edge], dtype=np.uint64)
        out = np.zeros_like(hv)
        lib.hostsim_filter_slots(ctypes.c_ulonglong(m), hv.ctypes.data_as(ctypes.c_void_p), ctypes.c_ulonglong(len(hv)),
                                 out.ctypes.data_as(ctypes.c_void_p))
        want = np.array([int(x) % m for x in hv[:len(edge)]], dtype=np.uint64)
        assert (out[:len(edge)] == want).all(), m
        assert (out == hv % np.uint64(m)).all(), m
