"""Host FASTA/FASTQ reader (ntedit_amd/host/fasta.cpp) against an independent model of the record
semantics the reference gets from kseq (lib/kseq.h:176-215 as used at ntedit.cpp:2219-2230)."""
import ctypes
import gzip
import os

import numpy as np
import pytest

import helpers as H

WS = b" \t\n\v\f\r"


def model(data):
    """kseq_read() loop -> [(header, sequence)] with header = name [+ " " + comment], both cut at
    an embedded NUL the way the reference's C-string handling does; stops at kseq's error returns"""
    pos, n = 0, len(data)
    last = 0
    out = []

    def getc():
        nonlocal pos
        if pos >= n:
            return -1
        pos += 1
        return data[pos - 1]

    def until_line(acc, strip=True):
        # ks_getuntil2(KS_SEP_LINE, append): -1 if nothing could be read at EOF
        nonlocal pos
        if pos >= n:
            return -1
        j = data.find(b"\n", pos)
        if j < 0:
            acc += data[pos:]
            pos = n
        else:
            acc += data[pos:j]
            pos = j + 1
        if strip and len(acc) > 1 and acc[-1] == 13:
            del acc[-1]
        return len(acc)

    def cstr(b):
        z = b.find(b"\0")
        return bytes(b if z < 0 else b[:z])

    while True:
        if last == 0:
            while True:
                c = getc()
                if c == -1 or c in b">@":
                    break
            if c == -1:
                break
            last = c
        # name: up to the first whitespace (or EOF)
        if pos >= n:
            break  # ks_getuntil(name) < 0
        j = pos
        while j < n and data[j] not in WS:
            j += 1
        name = bytearray(data[pos:j])
        delim = data[j] if j < n else -1
        pos = j + 1 if j < n else n
        comment = bytearray()
        if delim != -1 and delim != 10:
            until_line(comment)
        seq = bytearray()
        c = -1
        while True:
            c = getc()
            if c == -1 or c in b">+@":
                break
            if c == 10:
                continue
            seq.append(c)
            until_line(seq)
        if c == 62 or c == 64:
            last = c  # the first header character has been read
        hdr = cstr(name) + (b" " + cstr(comment) if len(comment) else b"")
        if c != 43:
            out.append((hdr, bytes(seq)))
            if c == -1:
                break
            continue
        # FASTQ quality
        while True:
            c = getc()
            if c == -1 or c == 10:
                break
        if c == -1:
            break  # -2: no quality string
        qual = bytearray()
        while until_line(qual) >= 0 and len(qual) < len(seq):
            pass
        last = 0
        if len(qual) != len(seq):
            break  # -2
        out.append((hdr, bytes(seq)))
    return out


def dump(path, tmp_path):
    lib = H.hostsim_lib()
    out = str(tmp_path / "dump.bin")
    rc = lib.hostsim_fasta_dump(ctypes.c_char_p(path.encode()), ctypes.c_char_p(out.encode()))
    assert rc >= 0
    recs = []
    with open(out, "rb") as f:
        while True:
            line = f.readline()
            if not line:
                break
            hl, sl = map(int, line.split())
            hdr = f.read(hl)
            assert f.read(1) == b"\n"
            seq = f.read(sl)
            assert f.read(1) == b"\n"
            recs.append((hdr, seq))
    assert len(recs) == rc
    return recs


CASES = [
    b">a\nACGT\nAC\n>b desc here\nGG\n",
    b">a\r\nACGT\r\nAC\r\n>b x\r\nGG\r\n",
    b"junk\n>a\tcomment with\ttabs\nAC\n\n\nGT\n>c\n\n>d\nA",
    b">a \r\nAC\n>b  two spaces\nG\n>c \nT\n",
    b">a\nAC>GT\nA@C\nA+C\n>b\nTT\n",
    b"@r1 x\nACGT\n+\nIIII\n@r2\nAC\nGT\n+r2\nII\nII\n>f\nAAA\n",
    b"@r1\nACGT\n+\nII\n",
    b"@r1\nACGT\n+",
    b">only_header",
    b">x\n",
    b"",
    b"\n\n",
    b">a\0b c\0d\nAC\0GT\n>e\nA\n",
    b">a\n\r\nAC\r\n\r\n",
]


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_reader_cases(tmp_path, ci):
    p = str(tmp_path / "in.fa")
    with open(p, "wb") as f:
        f.write(CASES[ci])
    assert dump(p, tmp_path) == model(CASES[ci])


def test_reader_gzip_and_large(tmp_path):
    rng = np.random.default_rng(5)
    parts = []
    for i in range(40):
        n = int(rng.integers(1, 400000))
        s = bytes(rng.choice(np.frombuffer(b"ACGTN", dtype=np.uint8), n))
        w = int(rng.integers(1, 200))
        parts.append(b">c%d len=%d\n" % (i, n) + b"\n".join(s[j:j + w] for j in range(0, n, w)) + b"\n")
    data = b"".join(parts)  # several MB: crosses the reader's buffer boundary many times
    p = str(tmp_path / "in.fa")
    with open(p, "wb") as f:
        f.write(data)
    want = model(data)
    assert dump(p, tmp_path) == want
    with gzip.open(p + ".gz", "wb", compresslevel=1) as f:
        f.write(data)
    assert dump(p + ".gz", tmp_path) == want


def test_reader_fuzz(tmp_path):
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b">@+\n\n\r \tACGTacgtN\0", dtype=np.uint8)
    p = str(tmp_path / "in.fa")
    for it in range(300):
        n = int(rng.integers(0, 120))
        data = bytes(rng.choice(alphabet, n))
        with open(p, "wb") as f:
            f.write(data)
        assert dump(p, tmp_path) == model(data), data


def map_dump(path, tmp_path, threads):
    """the mapped, multi-threaded reader (ntedit_amd/host/fasta_map.cpp); None when it refuses the file"""
    lib = H.hostsim_lib()
    out = str(tmp_path / "mdump.bin")
    rc = lib.hostsim_fasta_map_dump(ctypes.c_char_p(path.encode()), ctypes.c_char_p(out.encode()), ctypes.c_uint(threads))
    if rc == -1:
        return None
    assert rc >= 0
    recs = []
    with open(out, "rb") as f:
        while True:
            line = f.readline()
            if not line:
                break
            hl, sl = map(int, line.split())
            hdr = f.read(hl)
            assert f.read(1) == b"\n"
            seq = f.read(sl)
            assert f.read(1) == b"\n"
            recs.append((hdr, seq))
    assert len(recs) == rc
    return recs


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_mapped_reader_cases(tmp_path, ci):
    """whatever the mapped reader accepts it parses like kseq; CR line ends, FASTQ, NUL bytes, leading text and
    empty files are refused (the streaming reader takes those)"""
    p = str(tmp_path / "in.fa")
    with open(p, "wb") as f:
        f.write(CASES[ci])
    for threads in (1, 3):
        got = map_dump(p, tmp_path, threads)
        if got is not None:
            assert got == model(CASES[ci])
    accepted = map_dump(p, tmp_path, 1) is not None
    assert accepted == (ci in (0, 4, 8, 9)), ci  # 4: '>' '@' '+' inside a line are ordinary characters


def test_mapped_reader_large_and_refusals(tmp_path):
    rng = np.random.default_rng(6)
    parts = []
    for i in range(60):
        n = int(rng.integers(0, 300000))
        s = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), n))
        w = int(rng.integers(1, 200)) if i % 5 else max(1, n)
        hdr = b">c%d" % i if i % 3 else b">c%d  len=%d\tx y" % (i, n)
        body = b"\n".join(s[j:j + w] for j in range(0, n, w))
        parts.append(hdr + b"\n" + body + (b"\n\n" if i % 7 == 0 else b"\n"))
    data = b"".join(parts)[:-1]  # (no newline at the very end)
    p = str(tmp_path / "in.fa")
    with open(p, "wb") as f:
        f.write(data)
    want = model(data)
    assert dump(p, tmp_path) == want
    for threads in (1, 2, 7, 16):
        assert map_dump(p, tmp_path, threads) == want, threads
    with gzip.open(p + ".gz", "wb", compresslevel=1) as f:
        f.write(data)
    assert map_dump(p + ".gz", tmp_path, 4) is None
    for bad in (data.replace(b"\n", b"\r\n", 3), b"x" + data, data + b"\n@r\nAC\n+\nII\n", data[:1000] + b"\0" + data[1000:]):
        with open(p, "wb") as f:
            f.write(bad)
        assert map_dump(p, tmp_path, 4) is None


def test_mapped_reader_fuzz(tmp_path):
    rng = np.random.default_rng(12)
    alphabet = np.frombuffer(b">>\n\n\n \tACGTacgtN@+", dtype=np.uint8)
    p = str(tmp_path / "in.fa")
    n_acc = 0
    for it in range(400):
        n = int(rng.integers(0, 150))
        data = b">" + bytes(rng.choice(alphabet, n))
        with open(p, "wb") as f:
            f.write(data)
        got = map_dump(p, tmp_path, int(rng.integers(1, 5)))
        if got is not None:
            n_acc += 1
            assert got == model(data), data
    assert n_acc > 100


def bgzf_bytes(data, block=65280, rng=None, eof=True):
    """the BGZF container (SAM specification 4.1) around `data`, written here member by member; with `rng`, block
    sizes vary (1 byte .. 65280) and empty members are sprinkled in"""
    import struct
    import zlib
    out = []

    def member(chunk):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(chunk) + c.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        assert bsize < 65536
        out.append(b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize) +
                   body + struct.pack("<II", zlib.crc32(chunk) & 0xFFFFFFFF, len(chunk)))

    i = 0
    while i < len(data):
        n = block if rng is None else int(rng.integers(1, block + 1))
        member(data[i:i + n])
        i += n
        if rng is not None and rng.integers(0, 9) == 0:
            member(b"")
    if eof:
        member(b"")
    return b"".join(out)


def test_mapped_reader_bgzf(tmp_path):
    """bgzip-compressed drafts are inflated member by member on all threads and then read like a plain file; an
    ordinary gzip stream, a damaged member, a truncated file or FASTQ inside go to the streaming reader"""
    rng = np.random.default_rng(77)
    parts = []
    for i in range(40):
        n = int(rng.integers(0, 200000))
        s = bytes(rng.choice(np.frombuffer(b"ACGTNacgt", dtype=np.uint8), n))
        w = int(rng.integers(1, 120))
        parts.append((b">s%d some text\n" % i) + b"\n".join(s[j:j + w] for j in range(0, n, w)) + b"\n")
    data = b"".join(parts)
    want = model(data)
    p = str(tmp_path / "in.fa.gz")
    for variant in range(3):
        blob = bgzf_bytes(data, rng=rng if variant else None, eof=variant != 2)
        assert gzip.decompress(blob) == data  # (a valid multi-member gzip file)
        with open(p, "wb") as f:
            f.write(blob)
        assert dump(p, tmp_path) == want  # the streaming reader (zlib's gzread) sees the same text
        for threads in (1, 5, 16):
            assert map_dump(p, tmp_path, threads) == want, (variant, threads)
    blob = bgzf_bytes(data)
    bad = bytearray(blob)
    bad[len(bad) // 2] ^= 0x55  # a flipped bit inside some member's deflate data (or header)
    for broken in (bytes(bad), blob[:len(blob) // 2], blob + b"\x1f\x8b", gzip.compress(data, 1) + bgzf_bytes(b">x\nA\n")):
        with open(p, "wb") as f:
            f.write(broken)
        assert map_dump(p, tmp_path, 4) is None
    with open(p, "wb") as f:
        f.write(bgzf_bytes(b"@r1\nACGT\n+\nIIII\n"))
    assert map_dump(p, tmp_path, 4) is None
    assert dump(p, tmp_path) == [(b"r1", b"ACGT")]


def _as_driver(recs):
    """what the reference's caller keeps of kseq's record: contigSeq = seq->seq.s, a C string (ntedit.cpp:2230)"""
    return [(h, q if q.find(b"\0") < 0 else q[: q.find(b"\0")]) for h, q in recs]


def _abi_records(path, **kw):
    """the C-ABI ingest (ntedit_hip_fasta_load) that python -m ntedit_amd.run reads its draft with"""
    from ntedit_amd.run import read_fasta_fast
    return read_fasta_fast(path, **kw)


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_abi_ingest_cases(tmp_path, ci):
    """the multi-GPU driver must read a draft exactly like the `ntedit` binary does (same readers behind
    the C ABI): FASTQ records, CR line ends, NUL bytes, text in front of the first header, empty records"""
    p = str(tmp_path / "in.dat")  # (no .fa / .gz suffix to go by)
    with open(p, "wb") as f:
        f.write(CASES[ci])
    assert _abi_records(p) == _as_driver(model(CASES[ci]))
    with gzip.open(p + "z", "wb") as f:
        f.write(CASES[ci])
    assert _abi_records(p + "z") == _as_driver(model(CASES[ci]))
    with open(p + "b", "wb") as f:
        f.write(bgzf_bytes(CASES[ci], block=7))
    assert _abi_records(p + "b") == _as_driver(model(CASES[ci]))


def test_abi_ingest_min_len_fuzz_and_errors(tmp_path):
    from ntedit_amd import _lib
    rng = np.random.default_rng(12)
    alphabet = np.frombuffer(b">@+\n\n\r \tACGTacgtN\0", dtype=np.uint8)
    p = str(tmp_path / "in")
    for it in range(200):
        data = bytes(rng.choice(alphabet, int(rng.integers(0, 160))))
        with open(p, "wb") as f:
            f.write(data)
        z = int(rng.integers(0, 6))
        assert _abi_records(p, min_len=z) == [r for r in _as_driver(model(data)) if len(r[1]) >= z], data
    with pytest.raises(_lib.NtEditHipError):
        _abi_records(str(tmp_path / "missing.fa"))
    # a truncated gzip stream is an error, not a shorter draft
    big = b">a\n" + bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), 300000)) + b"\n>b\nACGT\n"
    with gzip.open(p + ".gz", "wb") as f:
        f.write(big)
    blob = open(p + ".gz", "rb").read()
    with open(p + ".cut", "wb") as f:
        f.write(blob[: len(blob) // 2])
    with pytest.raises(_lib.NtEditHipError):
        _abi_records(p + ".cut")
