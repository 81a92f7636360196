"""The host's own gzip decoder (ntedit_amd/host/gunzip.cpp) against zlib: it stands where the reference has zlib's
gzread under kseq (lib/kseq.h:44-50, KSEQ_INIT(gzFile, gzread) at ntedit.cpp:25), so whatever gzread returns for a
file -- the bytes, or a failure -- is what it must return."""
import ctypes
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

import helpers as H


def gunzip(path, tmp_path, block=1 << 16, in_bytes=0):
    lib = H.hostsim_lib()
    lib.hostsim_gunzip_buffers.restype = ctypes.c_longlong
    out = str(tmp_path / "out.bin")
    rc = lib.hostsim_gunzip_buffers(ctypes.c_char_p(path.encode()), ctypes.c_char_p(out.encode()), ctypes.c_uint(block),
                                    ctypes.c_uint(in_bytes))
    data = b""
    if rc != -1 and rc != -2:
        with open(out, "rb") as f:
            data = f.read()
    return rc, data


def deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=31, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, memlevel, strategy)
    return c.compress(data) + c.flush()


def dna(rng, n, width=60, repeats=True):
    a = rng.integers(0, 4, n, dtype=np.uint8)
    if repeats and n > 4000:
        pos = 2000
        while pos < n - 3000:
            L = int(rng.integers(20, 2000))
            src = int(rng.integers(0, pos - L)) if pos > L else 0
            a[pos:pos + L] = a[src:src + L]
            pos += int(rng.integers(500, 20000))
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[a].tobytes()
    return b">c1 a draft\n" + b"\n".join(s[i:i + width] for i in range(0, n, width)) + b"\n"


def payloads():
    rng = np.random.default_rng(7)
    text = (b"the quick brown fox jumps over the lazy dog; " * 3000) + bytes(rng.integers(32, 127, 20000, dtype=np.uint8))
    return {
        "empty": b"",
        "one": b"A",
        "dna_small": dna(rng, 5000),
        "dna": dna(rng, 3_000_000),
        "dna_norepeat": dna(rng, 400_000, repeats=False),
        "text": text,
        "random": bytes(rng.integers(0, 256, 300_000, dtype=np.uint8)),  # incompressible: stored blocks at any level
        "zeros": bytes(2_000_000),  # distance-1 matches of the maximum length
        "period3": b"ACG" * 500_000,  # overlapping matches with a distance below eight
        "skewed": bytes(rng.choice(np.arange(256, dtype=np.uint8), 500_000,
                                   p=np.array([2.0 ** -(1 + i // 4) for i in range(256)]) / sum(2.0 ** -(1 + i // 4) for i in range(256)))),
    }


PAYLOADS = payloads()
ENCODINGS = [
    ("stored", dict(level=0)),
    ("fast", dict(level=1)),
    ("default", dict(level=6)),
    ("best", dict(level=9)),
    ("fixed", dict(level=6, strategy=zlib.Z_FIXED)),
    ("huffman_only", dict(level=6, strategy=zlib.Z_HUFFMAN_ONLY)),
    ("rle", dict(level=6, strategy=zlib.Z_RLE)),
    ("small_blocks", dict(level=6, memlevel=1)),  # a new dynamic block every 256 symbols
    ("small_window", dict(level=6, wbits=16 + 9)),
]


@pytest.mark.parametrize("enc", [e[0] for e in ENCODINGS])
@pytest.mark.parametrize("name", list(PAYLOADS))
def test_gunzip_equals_zlib(tmp_path, name, enc):
    data = PAYLOADS[name]
    z = deflate(data, **dict(ENCODINGS)[enc])
    assert zlib.decompress(z, 31) == data
    p = str(tmp_path / "in.gz")
    with open(p, "wb") as f:
        f.write(z)
    for block in ((1 << 22), 65536, 4099) if len(data) > 100000 else ((1 << 22), 4099, 100, 7, 1):
        rc, got = gunzip(p, tmp_path, block)
        assert rc == len(data), (name, enc, block, rc)
        assert got == data, (name, enc, block)


def gz_header(flags=0, extra=b"", name=b"", comment=b"", hcrc=False):
    h = bytearray(b"\x1f\x8b\x08" + bytes([flags]) + b"\0\0\0\0\0\x03")
    if flags & 4:
        h += struct.pack("<H", len(extra)) + extra
    if flags & 8:
        h += name + b"\0"
    if flags & 16:
        h += comment + b"\0"
    if flags & 2:
        h += struct.pack("<H", zlib.crc32(bytes(h)) & 0xffff)
    return bytes(h)


def member(data, flags=0, level=6, **kw):
    raw = deflate(data, level=level, wbits=-15)
    return gz_header(flags, **kw) + raw + struct.pack("<II", zlib.crc32(data), len(data) & 0xffffffff)


def test_gunzip_header_fields_and_members(tmp_path):
    rng = np.random.default_rng(3)
    a, b, c = dna(rng, 70000), b"", dna(rng, 300)
    p = str(tmp_path / "in.gz")
    blob = (member(a, 4 | 8 | 16 | 2, extra=b"BC\x02\x00\x12\x34" * 50, name=b"draft.fa", comment=b"made by a test") + member(b) +
            member(c, 8, name=b"x" * 5000) + member(a, 1))
    assert gzip.decompress(blob) == a + b + c + a
    for tail in (b"", b"\0" * 700, b"trailing text that is no gzip member", b"\x1f"):
        with open(p, "wb") as f:
            f.write(blob + tail)
        for block in (1 << 20, 1000):
            rc, got = gunzip(p, tmp_path, block)
            assert rc == len(got) and got == a + b + c + a, (tail[:8], block, rc)
    # header fields longer than the decoder's input buffer (2 KB here): gzread takes any length
    long_blob = member(c, 4 | 8 | 16, extra=b"\x01" * 60000, name=b"n" * 9000, comment=b"c" * 70000) + member(a, 8, name=b"y" * 300000)
    assert gzip.decompress(long_blob) == c + a
    with open(p, "wb") as f:
        f.write(long_blob)
    for in_bytes in (2048, 4096, 0):
        rc, got = gunzip(p, tmp_path, 1 << 20, in_bytes)
        assert rc == len(got) and got == c + a, in_bytes
    with open(p, "wb") as f:
        f.write(long_blob[:len(long_blob) // 2 + 40000][:len(member(c, 4 | 8 | 16, extra=b"\x01" * 60000, name=b"n" * 9000, comment=b"c" * 70000)) + 200000])
    assert gunzip(p, tmp_path, 1 << 20, 2048)[0] == -3  # (the file ends inside the second member's name)
    # a file name that never ends (no NUL within tens of megabytes): an error, not the whole file pulled into memory
    with open(p, "wb") as f:
        f.write(member(c, 8, name=b"z" * 300)[:10] + b"z" * (40 << 20))
    for in_bytes in (2048, 0):
        assert gunzip(p, tmp_path, 1 << 20, in_bytes)[0] == -3
    # the second member starts, and the file ends inside its header / its data / its trailer
    full = member(a) + member(c)
    for cut in (len(member(a)) + 2, len(member(a)) + 9, len(full) - 9, len(full) - 8, len(full) - 1):
        with open(p, "wb") as f:
            f.write(full[:cut])
        with pytest.raises(Exception):
            gzip.decompress(full[:cut])
        rc, got = gunzip(p, tmp_path)
        assert rc == -3, cut
    # not gzip at all
    with open(p, "wb") as f:
        f.write(b">plain\nACGT\n")
    assert gunzip(p, tmp_path)[0] == -1
    with open(p, "wb") as f:
        f.write(b"")
    assert gunzip(p, tmp_path)[0] == -1


def test_gunzip_damaged_streams(tmp_path):
    """a damaged stream must never come back as a good one: it fails, or a member's checksum tells; where zlib still
    inflates it (a flipped bit in a header field nobody reads) the bytes are zlib's"""
    rng = np.random.default_rng(9)
    data = dna(rng, 60000) + b">c2\n" + bytes(rng.integers(65, 91, 3000, dtype=np.uint8)) + b"\n"
    p = str(tmp_path / "in.gz")
    for enc in ("default", "fixed", "stored", "small_blocks"):
        z = bytearray(deflate(data, **dict(ENCODINGS)[enc]))
        # truncations
        for cut in sorted(set(int(x) for x in rng.integers(1, len(z), 60)) | {1, 2, 9, 10, 11, len(z) - 1, len(z) - 8, len(z) - 9}):
            with open(p, "wb") as f:
                f.write(z[:cut])
            rc, got = gunzip(p, tmp_path, 5000)
            assert rc in (-1, -3), (enc, cut, rc)
            assert data.startswith(got)
        # flipped bits
        for it in range(150):
            y = bytearray(z)
            pos = int(rng.integers(0, len(y)))
            y[pos] ^= 1 << int(rng.integers(0, 8))
            with open(p, "wb") as f:
                f.write(y)
            try:
                want = gzip.decompress(bytes(y))
            except Exception:
                want = None
            rc, got = gunzip(p, tmp_path, 5000)
            if want is None:
                assert rc < 0, (enc, pos, rc)
            else:
                assert rc == len(want) and got == want, (enc, pos, rc)


def test_reader_backends_agree(tmp_path):
    """the streaming FASTA reader gives the same records, and the same verdict on damaged files, whichever inflater runs"""
    lib = H.hostsim_lib()
    from test_fasta_reader import dump, model
    rng = np.random.default_rng(21)
    parts = []
    for i in range(30):
        n = int(rng.integers(1, 600000))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), n))
        w = int(rng.integers(20, 200))
        parts.append(b">c%d len=%d\n" % (i, n) + b"\n".join(s[j:j + w] for j in range(0, n, w)) + b"\n")
    data = b"".join(parts)
    want = model(data)
    p = str(tmp_path / "in.fa.gz")
    variants = [deflate(data, 6), deflate(data, 1), deflate(data[:1 << 20], 6) + deflate(data[1 << 20:], 9), deflate(data, 0)]
    try:
        for z in variants:
            with open(p, "wb") as f:
                f.write(z)
            for backend in (0, 1):
                lib.hostsim_gzip_through_zlib(backend)
                assert dump(p, tmp_path) == want
                assert lib.hostsim_fasta_io_error(ctypes.c_char_p(p.encode())) == 0
        z = bytearray(variants[0])
        bad_crc = bytearray(z)
        bad_crc[-6] ^= 0x40
        bad_len = bytearray(z)
        bad_len[-2] ^= 0x01
        middle = bytearray(z)
        middle[len(z) // 2] ^= 0x10
        for broken in (z[:len(z) // 2], z[:-3], bad_crc, bad_len, middle):
            with open(p, "wb") as f:
                f.write(broken)
            for backend in (0, 1):
                lib.hostsim_gzip_through_zlib(backend)
                assert lib.hostsim_fasta_io_error(ctypes.c_char_p(p.encode())) == 1, backend
    finally:
        lib.hostsim_gzip_through_zlib(0)


class BitWriter:
    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def bits(self, value, count):  # LSB first (header fields, extra bits)
        self.acc |= value << self.n
        self.n += count
        while self.n >= 8:
            self.out.append(self.acc & 0xff)
            self.acc >>= 8
            self.n -= 8

    def code(self, code, length):  # a Huffman code: most significant bit first
        for i in range(length - 1, -1, -1):
            self.bits((code >> i) & 1, 1)

    def done(self):
        if self.n:
            self.out.append(self.acc & 0xff)
        return bytes(self.out)


def canonical(lengths):
    codes, code = {}, 0
    for ln in range(1, 16):
        for sym in sorted(s for s, l in lengths.items() if l == ln):
            codes[sym] = (code, ln)
            code += 1
        code <<= 1
    return codes


LEN_SYMS = {257: (3, 0), 258: (4, 0), 259: (5, 0), 260: (6, 0), 261: (7, 0), 262: (8, 0), 263: (9, 0), 264: (10, 0), 265: (11, 1), 266: (13, 1)}
DIST_SYMS = [(1, 0), (2, 0), (3, 0), (4, 0), (5, 1), (7, 1), (9, 2), (13, 2), (17, 3), (25, 3), (33, 4), (49, 4), (65, 5), (97, 5), (129, 6), (193, 6)]


def crafted_block(symbols, lit_lengths, dist_lengths, final=1):
    """a dynamic-Huffman block with exactly these code lengths; symbols = literals (ints) and (length symbol, extra,
    distance symbol, extra) tuples"""
    w = BitWriter()
    w.bits(final, 1)
    w.bits(2, 2)
    nlit, ndist = 286, 30
    w.bits(nlit - 257, 5)
    w.bits(ndist - 1, 5)
    w.bits(19 - 4, 4)
    pre = {s: 5 for s in range(16)}
    pre.update({16: 2, 17: 3, 18: 3})
    for s in (16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15):
        w.bits(pre[s], 3)
    pc = canonical(pre)
    for i in range(nlit):
        w.code(*pc[lit_lengths.get(i, 0)])
    for i in range(ndist):
        w.code(*pc[dist_lengths.get(i, 0)])
    lc, dc = canonical(lit_lengths), canonical(dist_lengths)
    for s in symbols:
        if isinstance(s, int):
            w.code(*lc[s])
        else:
            ls, lx, ds, dx = s
            w.code(*lc[ls])
            w.bits(lx, LEN_SYMS[ls][1])
            w.code(*dc[ds])
            w.bits(dx, DIST_SYMS[ds][1])
    if 256 in lc:
        w.code(*lc[256])
    return w.done()


def test_gunzip_crafted_code_lengths(tmp_path):
    """codes of every length from 1 to 15 in both alphabets (second-level tables of every size), every length and
    distance symbol of the set with its extra bits; zlib is the judge of what the stream says"""
    lit_lengths = {65: 1, 67: 2, 71: 3, 84: 4, 10: 5, 256: 6, 257: 7, 258: 8, 259: 9, 260: 10, 261: 11, 262: 12, 263: 13, 264: 14, 265: 15, 266: 15}
    dist_lengths = {i: i + 1 for i in range(15)}
    dist_lengths[15] = 15
    rng = np.random.default_rng(4)
    symbols = [int(x) for x in rng.choice([65, 67, 71, 84, 10], 300)]
    for rep in range(40):
        for ls in LEN_SYMS:
            ds = int(rng.integers(0, 16))
            symbols.append((ls, int(rng.integers(0, 1 << LEN_SYMS[ls][1])), ds, int(rng.integers(0, 1 << DIST_SYMS[ds][1]))))
            symbols += [int(x) for x in rng.choice([65, 67, 71, 84, 10], int(rng.integers(0, 6)))]
    raw = crafted_block(symbols, lit_lengths, dist_lengths)
    data = zlib.decompress(raw, -15)
    assert len(data) > 2000
    p = str(tmp_path / "in.gz")
    with open(p, "wb") as f:
        f.write(gz_header() + raw + struct.pack("<II", zlib.crc32(data), len(data)))
    for block in (1 << 16, 50, 1):
        rc, got = gunzip(p, tmp_path, block)
        assert rc == len(data) and got == data, block
    # the same with a match that reaches behind the start of the output: zlib and the decoder both refuse
    bad = crafted_block([65, 67, (257, 0, 4, 1)], lit_lengths, dist_lengths)
    with pytest.raises(zlib.error):
        zlib.decompress(bad, -15)
    with open(p, "wb") as f:
        f.write(gz_header() + bad + struct.pack("<II", 0, 0))
    assert gunzip(p, tmp_path)[0] == -3
    # code length sets zlib rejects: over-subscribed, incomplete (either alphabet), no end-of-block code
    no_eob = {k: v for k, v in lit_lengths.items() if k != 256}
    no_eob[66] = 6
    for lit, dist in (({**lit_lengths, 66: 1}, dist_lengths), ({k: v for k, v in lit_lengths.items() if k != 266}, dist_lengths),
                      (lit_lengths, {k: v for k, v in dist_lengths.items() if k != 15}), (no_eob, dist_lengths)):
        w = crafted_block([65, 67], lit, dist)
        with pytest.raises(zlib.error):
            zlib.decompress(w, -15)
        with open(p, "wb") as f:
            f.write(gz_header() + w + struct.pack("<II", 0, 0))
        assert gunzip(p, tmp_path)[0] == -3
    # a single distance code of one bit is an incomplete set zlib accepts
    one = crafted_block([65, 67, 71, 84, (257, 0, 0, 0), 10], lit_lengths, {0: 1})
    data = zlib.decompress(one, -15)
    with open(p, "wb") as f:
        f.write(gz_header() + one + struct.pack("<II", zlib.crc32(data), len(data)))
    assert gunzip(p, tmp_path) == (len(data), data)


@pytest.mark.parametrize("enc", ["default", "fast", "fixed", "stored", "small_blocks"])
def test_gunzip_input_buffer_seams(tmp_path, enc):
    """the compressed file read through buffers of a few KB: thousands of places where the decoder has to stop short of
    the buffered bytes and carry on after the refill, at every alignment"""
    rng = np.random.default_rng(12)
    data = dna(rng, 1_500_000) + PAYLOADS["text"][:200000] + PAYLOADS["random"][:50000]
    z = deflate(data, **dict(ENCODINGS)[enc])
    p = str(tmp_path / "in.gz")
    with open(p, "wb") as f:
        f.write(z)
    for in_bytes in (2048, 2049, 3001, 4096, 5003, 8192 + 7, 65536):
        rc, got = gunzip(p, tmp_path, 1 << 20, in_bytes)
        assert rc == len(data) and got == data, (enc, in_bytes, rc)
    for cut in (len(z) // 3, len(z) - 5):
        with open(p, "wb") as f:
            f.write(z[:cut])
        assert gunzip(p, tmp_path, 1 << 20, 2048)[0] == -3
