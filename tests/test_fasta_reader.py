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
