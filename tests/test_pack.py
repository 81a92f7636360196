"""CPU tier: the packed form of a batch (include/ntedit_hip.h: 4-bit codes + a case bit per base) as
ntedit_hip_pack_bases writes it, against a numpy restatement of the layout and of the device-side unpacking."""
import ctypes

import numpy as np
import pytest

from ntedit_amd import _lib

LETTERS = np.frombuffer(b"ACGTRYSWKMBDHVNN", dtype=np.uint8)


def unpack(packed, n):
    """what k_unpack writes (nte_kernels.hip)"""
    codes_bytes = (n + 31) // 32 * 16
    codes = packed[:codes_bytes]
    cases = packed[codes_bytes:codes_bytes + (n + 127) // 128 * 16]
    nib = np.empty(codes_bytes * 2, dtype=np.uint8)
    nib[0::2] = codes & 15
    nib[1::2] = codes >> 4
    low = np.unpackbits(cases, bitorder="little")[:n].astype(np.uint8)
    return LETTERS[nib[:n]] | (low << 5)


def expected_bytes(raw):
    """the byte batch the device ends up with: accepted bases as they are, every other byte 'N' / 'n'"""
    a = np.frombuffer(raw, dtype=np.uint8)
    up = a & 0xDF
    accepted = np.isin(up, np.frombuffer(b"ACGTRYSWKMBDHV", dtype=np.uint8)) & (((a >= 65) & (a <= 90)) | ((a >= 97) & (a <= 122)))
    lower = (a >= 97) & (a <= 122)
    out = np.where(accepted, a, np.where(lower, ord("n"), ord("N"))).astype(np.uint8)
    return out


@pytest.mark.parametrize("n", [0, 1, 2, 15, 16, 17, 31, 127, 128, 129, 1000, (1 << 20) + 77, 3 * (1 << 20) + 5])
@pytest.mark.parametrize("threads", [1, 4])
def test_pack_bases_layout(n, threads):
    lib = _lib.load()
    rng = np.random.default_rng(n * 7 + threads)
    alphabet = np.frombuffer(b"ACGTacgtNnRYSWKMBDHVryswkmbdhv\nXx*.", dtype=np.uint8)
    raw = alphabet[rng.integers(0, len(alphabet), n)].tobytes()
    size = lib.ntedit_hip_packed_size(n)
    assert size == (n + 31) // 32 * 16 + (n + 127) // 128 * 16
    out = np.full(size + 16, 0xAB, dtype=np.uint8)
    rc = lib.ntedit_hip_pack_bases(ctypes.c_char_p(raw), n, out.ctypes.data_as(ctypes.c_void_p), threads)
    assert rc == 0
    assert (out[size:] == 0xAB).all()
    got = unpack(out[:size], n)
    assert (got == expected_bytes(raw)).all()


def test_pack_bases_refuses_what_it_cannot_carry():
    """U / u hash like T, a few other bytes pick up a seed through (c & 7) (nte_common.h, is_exotic): those batches go as bytes"""
    lib = _lib.load()
    base = b"ACGTN" * 300000
    size = lib.ntedit_hip_packed_size(len(base) + 1)
    out = np.zeros(size, dtype=np.uint8)
    for bad in (b"U", b"u", b"-", b"\x01", b"5"):
        for at in (0, 777777, len(base)):
            raw = base[:at] + bad + base[at:]
            assert lib.ntedit_hip_pack_bases(ctypes.c_char_p(raw), len(raw), out.ctypes.data_as(ctypes.c_void_p), 3) == 1, (bad, at)
    assert lib.ntedit_hip_pack_bases(ctypes.c_char_p(base), len(base), out.ctypes.data_as(ctypes.c_void_p), 3) == 0
    assert lib.ntedit_hip_pack_bases(None, 5, out.ctypes.data_as(ctypes.c_void_p), 1) < 0


def test_pack_bases_every_byte_value_both_forms():
    """every byte value, in the vector form (32 bases per step) and in the table loop (threads | 1 << 31): the same
    packed bytes, and the same verdict on what the packed form cannot carry (nte_common.h is_exotic: a byte without
    a code whose reverse-strand seed slot (c & 7) is not empty)"""
    lib = _lib.load()
    accepted = set(b"ACGTRYSWKMBDHVacgtryswkmbdhv")
    fill = b"ACGTNacgtn\nRYKM" * 9  # 144 bytes: the byte under test lands in a whole vector step and in the tail
    for c in range(256):
        exotic = c not in accepted and (c & 7) in (1, 3, 4, 5, 7)
        for at in (0, 37, 100, 143):
            raw = fill[:at] + bytes([c]) + fill[at + 1:]
            n = len(raw)
            size = lib.ntedit_hip_packed_size(n)
            outs = []
            for threads in (1, 1 | (1 << 31)):
                out = np.zeros(size, dtype=np.uint8)
                rc = lib.ntedit_hip_pack_bases(ctypes.c_char_p(raw), n, out.ctypes.data_as(ctypes.c_void_p), threads)
                assert rc == (1 if exotic else 0), (c, at, threads)
                outs.append(out)
            if not exotic:
                assert (outs[0] == outs[1]).all(), (c, at)
                assert (unpack(outs[0], n) == expected_bytes(raw)).all(), (c, at)
