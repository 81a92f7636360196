"""bench.py's input writers (not the product): the single-member gzip stream of `--e2e-gzip` and the BGZF container of
`--e2e-bgzf` must be valid files of their kind, or the end-to-end numbers measure an error path."""
import gzip
import os
import sys
import zlib

import numpy as np

import helpers as H

sys.path.insert(0, H.ROOT)
import bench  # noqa: E402


def test_crc32_combine():
    rng = np.random.default_rng(1)
    for la, lb in ((0, 0), (1, 0), (0, 5), (7, 1), (1000, 33), (65536, 100001)):
        a = bytes(rng.integers(0, 256, la, dtype=np.uint8))
        b = bytes(rng.integers(0, 256, lb, dtype=np.uint8))
        assert bench._crc32_combine(zlib.crc32(a), zlib.crc32(b), lb) == zlib.crc32(a + b)


def test_gzip_one_stream_is_one_valid_member(tmp_path):
    rng = np.random.default_rng(2)
    data = b">c\n" + bytes(rng.choice(np.frombuffer(b"ACGT\n", dtype=np.uint8), 3_000_000))
    src, dst = str(tmp_path / "d.fa"), str(tmp_path / "d.fa.gz")
    with open(src, "wb") as f:
        f.write(data)
    bench.gzip_one_stream(src, dst, 2)
    with gzip.open(dst, "rb") as f:
        assert f.read() == data
    with open(dst, "rb") as f:
        z = f.read()
    d = zlib.decompressobj(31)
    assert d.decompress(z) == data and d.eof and d.unused_data == b""  # ONE member, nothing behind it


def test_bgzf_writer(tmp_path):
    rng = np.random.default_rng(3)
    data = b">c\n" + bytes(rng.choice(np.frombuffer(b"ACGT\n", dtype=np.uint8), 500_000))
    src, dst = str(tmp_path / "d.fa"), str(tmp_path / "d.fa.gz")
    with open(src, "wb") as f:
        f.write(data)
    bench.bgzf_compress(src, dst, 2)
    with gzip.open(dst, "rb") as f:
        assert f.read() == data
    with open(dst, "rb") as f:
        head = f.read(18)
    assert head[:4] == b"\x1f\x8b\x08\x04" and head[12:14] == b"BC"  # a BGZF member: FEXTRA with the BC subfield
    assert os.path.getsize(dst) > 0
