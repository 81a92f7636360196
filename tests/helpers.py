"""Shared helpers for the test-suite: oracle / hostsim loaders, FASTA + .bf IO,
synthetic genome generators.  Test infrastructure only."""
import ctypes
import gzip
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_BUILD = os.path.join(ORACLE_DIR, "_build")
HOSTSIM_DIR = os.path.join(ROOT, "tests", "hostsim")
GOLDEN = os.path.join(ROOT, "tests", "golden")


class HipParams(ctypes.Structure):
    _fields_ = [
        ("min_contig_len", ctypes.c_uint32),
        ("max_insertions", ctypes.c_uint32),
        ("max_deletions", ctypes.c_uint32),
        ("edit_threshold", ctypes.c_float),
        ("missing_threshold", ctypes.c_float),
        ("edit_ratio", ctypes.c_float),
        ("missing_ratio", ctypes.c_float),
        ("use_ratio", ctypes.c_int32),
        ("jump", ctypes.c_uint32),
        ("mode", ctypes.c_int32),
        ("snv", ctypes.c_int32),
        ("mask", ctypes.c_int32),
        ("min_threshold", ctypes.c_uint32),
        ("max_threshold", ctypes.c_uint32),
        ("start_grid", ctypes.c_uint32),
        ("node_window", ctypes.c_uint32),
        ("screen_mode", ctypes.c_uint32),
        ("event_budget", ctypes.c_uint32),
    ]


def default_params(**kw):
    p = HipParams(100, 5, 5, 9.0, 5.0, 0.5, 0.5, 0, 3, 0, 0, 0, 1, 255, 0, 0, 0, 0)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)
    return ORACLE_BUILD


def build_hostsim():
    subprocess.run(["make", "-s", "-C", HOSTSIM_DIR], check=True)
    return os.path.join(HOSTSIM_DIR, "_build", "libhostsim.so")


_oracle = None


def oracle_lib():
    global _oracle
    if _oracle is None:
        build_oracle()
        lib = ctypes.CDLL(os.path.join(ORACLE_BUILD, "libntedit_oracle.so"))
        lib.ora_srol.restype = ctypes.c_uint64
        lib.ora_srol.argtypes = [ctypes.c_uint64]
        lib.ora_sror.restype = ctypes.c_uint64
        lib.ora_sror.argtypes = [ctypes.c_uint64]
        lib.ora_srol_n.restype = ctypes.c_uint64
        lib.ora_srol_n.argtypes = [ctypes.c_uint64, ctypes.c_uint]
        lib.ora_base_forward_hash.restype = ctypes.c_uint64
        lib.ora_base_forward_hash.argtypes = [ctypes.c_char_p, ctypes.c_uint]
        lib.ora_base_reverse_hash.restype = ctypes.c_uint64
        lib.ora_base_reverse_hash.argtypes = [ctypes.c_char_p, ctypes.c_uint]
        lib.ora_next_forward_hash.restype = ctypes.c_uint64
        lib.ora_next_forward_hash.argtypes = [ctypes.c_uint64, ctypes.c_uint, ctypes.c_ubyte, ctypes.c_ubyte]
        lib.ora_next_reverse_hash.restype = ctypes.c_uint64
        lib.ora_next_reverse_hash.argtypes = [ctypes.c_uint64, ctypes.c_uint, ctypes.c_ubyte, ctypes.c_ubyte]
        _oracle = lib
    return _oracle


class OraParams(ctypes.Structure):
    """ora_params of oracle/ntedit_oracle.h"""
    _fields_ = [("k", ctypes.c_uint), ("h", ctypes.c_uint), ("jump", ctypes.c_uint),
                ("min_contig_len", ctypes.c_uint), ("max_insertions", ctypes.c_uint),
                ("max_deletions", ctypes.c_uint), ("edit_threshold", ctypes.c_float),
                ("missing_threshold", ctypes.c_float), ("edit_ratio", ctypes.c_float),
                ("missing_ratio", ctypes.c_float), ("use_ratio", ctypes.c_int),
                ("insertion_cap", ctypes.c_uint), ("mode", ctypes.c_int), ("snv", ctypes.c_int),
                ("mask", ctypes.c_int), ("secbf", ctypes.c_int), ("min_threshold", ctypes.c_uint),
                ("max_threshold", ctypes.c_uint)]


def oracle_polish_flat(blob, offsets, lens, bits, hash_num, k, names=None, fa_path=None, tsv_path=None,
                       rep_bits=None, rep_hash_num=0, **params):
    """The oracle over an in-memory batch and in-memory filter(s) (default parameters unless overridden
    by name, 1 thread); returns the number of bases it processed."""
    import numpy as np
    lib = oracle_lib()
    lib.ora_polish_batch_flat.restype = ctypes.c_uint64
    p = OraParams()
    lib.ora_params_default(ctypes.byref(p))
    for name, v in params.items():
        setattr(p, name, v)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    rep_ptr, rep_n = None, 0
    if rep_bits is not None:
        rep_bits = np.ascontiguousarray(rep_bits, dtype=np.uint8)
        rep_ptr, rep_n = rep_bits.ctypes.data_as(ctypes.c_void_p), rep_bits.size
    arr = None
    if names is not None:
        arr = (ctypes.c_char_p * max(len(names), 1))(*names)
    return lib.ora_polish_batch_flat(
        ctypes.c_char_p(blob), offsets.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p), arr,
        ctypes.c_uint32(len(lens)), bits.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(bits.size),
        ctypes.c_uint(hash_num), ctypes.c_uint(k), rep_ptr, ctypes.c_uint64(rep_n), ctypes.c_uint(rep_hash_num),
        ctypes.byref(p), fa_path.encode() if fa_path else None, tsv_path.encode() if tsv_path else None)


def oracle_polish_flat_mt(blob, offsets, lens, bits, hash_num, k, threads):
    """Timing only: the oracle with contigs handed out to `threads` worker threads like the reference's
    OpenMP loop (default parameters); returns the number of bases polished."""
    import numpy as np
    lib = oracle_lib()
    lib.ora_polish_batch_flat_mt.restype = ctypes.c_uint64
    p = OraParams()
    lib.ora_params_default(ctypes.byref(p))
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    if isinstance(blob, np.ndarray):
        bptr = ctypes.c_void_p(np.ascontiguousarray(blob, dtype=np.uint8).ctypes.data)
    else:
        bptr = ctypes.cast(ctypes.c_char_p(blob), ctypes.c_void_p)
    return lib.ora_polish_batch_flat_mt(
        bptr, offsets.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p),
        ctypes.c_uint32(len(lens)), bits.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(bits.size),
        ctypes.c_uint(hash_num), ctypes.c_uint(k), ctypes.byref(p), ctypes.c_uint(threads))


def oracle_polish_flat_mt_files(blob, offsets, lens, names, bits, hash_num, k, threads, fa_path=None, tsv_path=None,
                                vcf_path=None, rep_bits=None, rep_hash_num=0, **params):
    """The oracle over an in-memory batch with contigs handed out to `threads` worker threads; the complete
    _edited.fa / _changes.tsv / VCF body are written in input order (= ora_polish_batch_flat's files).
    blob may be bytes or a numpy uint8 array (GB-sized batches: no copy).  Returns the bases polished."""
    lib = oracle_lib()
    lib.ora_polish_batch_flat_mt_files.restype = ctypes.c_uint64
    p = OraParams()
    lib.ora_params_default(ctypes.byref(p))
    for name, v in params.items():
        setattr(p, name, v)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    lens = np.ascontiguousarray(lens, dtype=np.uint32)
    bits = np.ascontiguousarray(bits, dtype=np.uint8)
    rep_ptr, rep_n = None, 0
    if rep_bits is not None:
        rep_bits = np.ascontiguousarray(rep_bits, dtype=np.uint8)
        rep_ptr, rep_n = rep_bits.ctypes.data_as(ctypes.c_void_p), rep_bits.size
    if isinstance(blob, np.ndarray):
        keep = np.ascontiguousarray(blob, dtype=np.uint8)
        bptr = ctypes.c_void_p(keep.ctypes.data)
    else:
        bptr = ctypes.cast(ctypes.c_char_p(blob), ctypes.c_void_p)
    arr = (ctypes.c_char_p * max(len(names), 1))(*names)
    enc = lambda x: x.encode() if x else None  # noqa: E731
    return lib.ora_polish_batch_flat_mt_files(
        bptr, offsets.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p), arr,
        ctypes.c_uint32(len(lens)), bits.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(bits.size),
        ctypes.c_uint(hash_num), ctypes.c_uint(k), rep_ptr, ctypes.c_uint64(rep_n), ctypes.c_uint(rep_hash_num),
        ctypes.byref(p), ctypes.c_uint(threads), enc(fa_path), enc(tsv_path), enc(vcf_path))


def usable_cpus():
    """CPUs this process may use (cgroup quota aware)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


_hostsim = None


def hostsim_lib():
    global _hostsim
    if _hostsim is None:
        _hostsim = ctypes.CDLL(build_hostsim())
    return _hostsim


def read_fasta(path):
    """kseq-like: returns [(header_text, sequence_bytes)]; header = name [+ ' ' + comment]."""
    op = gzip.open if path.endswith(".gz") else open
    recs = []
    name, chunks = None, []
    with op(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    recs.append((name, b"".join(chunks)))
                h = line[1:]
                parts = h.split(None, 1)
                nm = parts[0] if parts else b""
                rest = h[len(nm) + 1:] if len(h) > len(nm) else b""
                name = nm + (b" " + rest if rest else b"")
                chunks = []
            elif name is not None:
                chunks.append(line)
    if name is not None:
        recs.append((name, b"".join(chunks)))
    return recs


def write_fasta(path, recs, width=0):
    with open(path, "wb") as f:
        for name, seq in recs:
            f.write(b">" + name + b"\n")
            if width:
                for i in range(0, len(seq), width):
                    f.write(seq[i:i + width] + b"\n")
            else:
                f.write(seq + b"\n")


def load_bf(path):
    """returns dict(k, hash_num, bytes, counting, data=np.uint8 array)"""
    with open(path, "rb") as f:
        first = f.readline()
        assert first.startswith(b"[BTL"), first
        meta = {"counting": b"Counting" in first}
        while True:
            line = f.readline()
            if not line:
                raise ValueError("no [HeaderEnd]")
            if line.startswith(b"[HeaderEnd]"):
                break
            if b"=" in line:
                key, val = [x.strip() for x in line.split(b"=", 1)]
                meta[key.decode()] = val.decode().strip('"')
        data = np.frombuffer(f.read(), dtype=np.uint8)
    meta["k"] = int(meta["k"])
    meta["hash_num"] = int(meta["hash_num"])
    meta["bytes"] = int(meta["bytes"])
    assert data.size == meta["bytes"], (data.size, meta["bytes"])
    meta["data"] = data
    return meta


def pack_batch(recs, min_len=0):
    """Batch layout of include/ntedit_hip.h: contigs separated by '\\n'."""
    names, offs, lens, parts = [], [], [], []
    pos = 0
    for name, seq in recs:
        if len(seq) < min_len:
            continue
        names.append(name)
        offs.append(pos)
        lens.append(len(seq))
        parts.append(seq)
        parts.append(b"\n")
        pos += len(seq) + 1
    blob = b"".join(parts)
    return blob, np.array(offs, dtype=np.uint64), np.array(lens, dtype=np.uint32), names


def vcf_body(path):
    """VCF lines without the two header lines that depend on the day / the input path"""
    return [l for l in open(path).read().splitlines()
            if not l.startswith("##fileDate=") and not l.startswith("##reference=")]


def run_hostsim(recs, bf, params, out_prefix, rep=None, annot_path=None):
    lib = hostsim_lib()
    blob, offs, lens, names = pack_batch(recs, params.min_contig_len)
    n = len(names)
    name_arr = (ctypes.c_char_p * max(n, 1))(*names)
    nev = ctypes.c_uint64(0)
    nap = ctypes.c_uint64(0)
    rc = lib.hostsim_polish(
        ctypes.c_char_p(blob), ctypes.c_uint64(len(blob)),
        offs.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p),
        name_arr, ctypes.c_uint32(n),
        bf["data"].ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(bf["bytes"]),
        ctypes.c_uint32(bf["hash_num"]), ctypes.c_uint32(bf["k"]),
        rep["data"].ctypes.data_as(ctypes.c_void_p) if rep else None,
        ctypes.c_uint64(rep["bytes"] if rep else 0),
        ctypes.c_uint32(rep["hash_num"] if rep else 0),
        ctypes.byref(params),
        (out_prefix + "_edited.fa").encode(), (out_prefix + "_changes.tsv").encode(),
        ctypes.byref(nev), ctypes.byref(nap),
        ctypes.c_int(1 if bf.get("counting") else 0), ctypes.c_int(1 if (rep and rep.get("counting")) else 0),
        (out_prefix + "_variants.vcf").encode(), annot_path.encode() if annot_path else None)
    return rc, nev.value, nap.value


def oracle_args(params):
    a = ["-z", str(params.min_contig_len), "-i", str(params.max_insertions), "-d", str(params.max_deletions),
         "-j", str(params.jump), "-m", str(params.mode), "-a", str(params.mask),
         "-p", str(params.min_threshold), "-q", str(params.max_threshold), "-s", str(params.snv)]
    if params.use_ratio:
        a += ["-X", repr(float(params.missing_ratio)), "-Y", repr(float(params.edit_ratio))]
    else:
        a += ["-x", repr(float(params.missing_threshold)), "-y", repr(float(params.edit_threshold))]
    return a


def run_oracle(draft_path, bf_path, params, out_prefix, rep_path=None, annot_path=None):
    build_oracle()
    cmd = [os.path.join(ORACLE_BUILD, "ntedit_oracle"), "-f", draft_path, "-r", bf_path, "-b", out_prefix]
    if rep_path:
        cmd += ["-e", rep_path]
    if annot_path:
        cmd += ["-l", annot_path]
    cmd += oracle_args(params)
    subprocess.run(cmd, check=True)


def mkbf(fasta_paths, out, k=25, hashes=3, nbytes=1 << 20, counting=False):
    build_oracle()
    subprocess.run([os.path.join(ORACLE_BUILD, "mkbf"), "-k", str(k), "-g", str(hashes), "-s", str(nbytes),
                    "-o", out] + (["-C"] if counting else []) + list(fasta_paths), check=True)


# ---------------------------------------------------------------- synthetic data
def random_genome(rng, n):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=n)].tobytes()


def mutate(rng, seq, p_sub=1e-3, p_ins=1e-4, p_del=1e-4, max_indel=5):
    """truth -> draft with substitutions, insertions and deletions"""
    out = bytearray()
    i = 0
    n = len(seq)
    acgt = b"ACGT"
    while i < n:
        r = rng.random()
        if r < p_sub:
            c = seq[i]
            choices = [x for x in acgt if x != c]
            out.append(choices[int(rng.integers(0, 3))])
            i += 1
        elif r < p_sub + p_ins:
            l = int(min(max_indel, rng.geometric(0.6)))
            for _ in range(l):
                out.append(acgt[int(rng.integers(0, 4))])
        elif r < p_sub + p_ins + p_del:
            l = int(min(max_indel, rng.geometric(0.6)))
            i += l
        else:
            out.append(seq[i])
            i += 1
    return bytes(out)


def oracle_screen(blob, bf, min_threshold=1):
    lib = oracle_lib()
    out = np.zeros((len(blob) + 63) // 64, dtype=np.uint64)
    if bf.get("counting"):
        lib.ora_screen_counting_flat(ctypes.c_char_p(blob), ctypes.c_size_t(len(blob)),
                                     bf["data"].ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(bf["bytes"]),
                                     ctypes.c_uint(bf["hash_num"]), ctypes.c_uint(bf["k"]),
                                     ctypes.c_uint(min_threshold), out.ctypes.data_as(ctypes.c_void_p))
        return out
    lib.ora_screen_flat(ctypes.c_char_p(blob), ctypes.c_size_t(len(blob)),
                        bf["data"].ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(bf["bytes"]),
                        ctypes.c_uint(bf["hash_num"]), ctypes.c_uint(bf["k"]),
                        out.ctypes.data_as(ctypes.c_void_p))
    return out


def make_case(tmp, seed, n=60000, p_sub=2e-3, p_ins=3e-4, p_del=3e-4, k=25, hashes=3, bfbytes=1 << 17,
              contigs=3, flavor=""):
    """Writes truth.fa / t.bf / draft.fa (+ r.bf when 'sec' in flavor) under tmp.
    Returns dict(draft=..., bf=..., rep=... or None)."""
    os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(seed)
    truth = [random_genome(rng, n) for _ in range(contigs)]
    if "rep" in flavor:
        t = bytearray(truth[0])
        unit = random_genome(rng, int(rng.integers(1, 7)))
        pos = int(rng.integers(1000, n - 2000))
        rep = (unit * 400)[: int(rng.integers(60, 300))]
        t[pos:pos + len(rep)] = rep
        truth[0] = bytes(t)
    write_fasta(os.path.join(tmp, "truth.fa"), [(b"t%d" % i, s) for i, s in enumerate(truth)])
    if "cbf" in flavor:
        # counting filter: the genome 3x, its first third two more times, one stretch only once
        write_fasta(os.path.join(tmp, "part.fa"), [(b"p%d" % i, s[: len(s) // 3]) for i, s in enumerate(truth)])
        lo = [(b"l%d" % i, s[len(s) // 2: len(s) // 2 + 3000]) for i, s in enumerate(truth)]
        thin = [(b"t%d" % i, s[: len(s) // 2] ) for i, s in enumerate(truth)] + \
               [(b"u%d" % i, s[len(s) // 2 + 3000:]) for i, s in enumerate(truth)]
        write_fasta(os.path.join(tmp, "thin.fa"), thin)
        write_fasta(os.path.join(tmp, "lo.fa"), lo)
        srcs = [os.path.join(tmp, "thin.fa")] * 3 + [os.path.join(tmp, "part.fa")] * 2 + [os.path.join(tmp, "lo.fa")]
        mkbf(srcs, os.path.join(tmp, "t.bf"), k=k, hashes=hashes, nbytes=bfbytes * 8, counting=True)
    else:
        mkbf([os.path.join(tmp, "truth.fa")], os.path.join(tmp, "t.bf"), k=k, hashes=hashes, nbytes=bfbytes)
    draft = []
    for i, s in enumerate(truth):
        d = bytearray(mutate(rng, s, p_sub, p_ins, p_del))
        if "N" in flavor:
            for _ in range(3):
                q = int(rng.integers(0, len(d) - 50))
                d[q:q + int(rng.integers(1, 40))] = b"N"
        if "lower" in flavor:
            q = int(rng.integers(0, len(d) - 5000))
            d[q:q + 3000] = bytes(d[q:q + 3000]).lower()
        if "exotic" in flavor:
            # bytes that are not accepted bases but carry ntHash seeds (U hashes like T; '-', '5', ... pick up a
            # complement-slot seed through c & 7); placed close to errors so that indel sweeps hash across them
            for _ in range(60):
                q = int(rng.integers(0, len(d)))
                d[q] = b"Uu-*5!1="[int(rng.integers(0, 8))]
        if "iupac" in flavor:
            for _ in range(20):
                q = int(rng.integers(0, len(d)))
                d[q] = b"RYSWKMBDHV"[int(rng.integers(0, 10))]
        draft.append((b"c%d some comment" % i if i % 2 else b"c%d" % i, bytes(d)))
    draft.append((b"short", b"ACGT" * 10))
    write_fasta(os.path.join(tmp, "draft.fa"), draft, width=70)
    out = {"draft": os.path.join(tmp, "draft.fa"), "bf": os.path.join(tmp, "t.bf"), "rep": None,
           "truth": os.path.join(tmp, "truth.fa")}
    if "sec" in flavor:
        write_fasta(os.path.join(tmp, "rep.fa"), [(b"r", truth[0][1000:1600])])
        mkbf([os.path.join(tmp, "rep.fa")], os.path.join(tmp, "r.bf"), k=k, hashes=hashes, nbytes=1 << 14)
        out["rep"] = os.path.join(tmp, "r.bf")
    return out


def make_sweep_rich_case(tmp, seed=5, n=30000):
    """A counting-filter case in which position after position needs an indel sweep (bench.py --counting in
    small): a small filter, an error-rich draft, and the counters that are not zero rewritten to 1..4 by slot index,
    so that -p 2 cuts into the k-mers that are there.  Returns make_case's dict."""
    case = make_case(tmp, seed, n=n, contigs=2, flavor="cbf", bfbytes=1 << 14, p_sub=1e-2, p_ins=3e-3, p_del=3e-3)
    raw = open(case["bf"], "rb").read()
    cut = raw.index(b"[HeaderEnd]\n") + len(b"[HeaderEnd]\n")
    old = np.frombuffer(raw[cut:], dtype=np.uint8)
    idx = np.arange(old.size, dtype=np.uint64)
    val = (1 + ((idx * np.uint64(2654435761) + np.uint64(7)) >> np.uint64(13)) % np.uint64(4)).astype(np.uint8)
    with open(case["bf"], "wb") as f:
        f.write(raw[:cut] + np.where(old != 0, val, 0).astype(np.uint8).tobytes())
    return case


SWEEP_RICH_PARAMS = [dict(), dict(event_budget=8), dict(start_grid=16, max_insertions=2, max_deletions=3),
                     dict(jump=1, event_budget=100), dict(max_threshold=3)]


# (case kwargs, parameter kwargs) pairs shared by the hostsim (CPU) and GPU parity tests
PARITY_CONFIGS = [
    (dict(), dict()),
    (dict(flavor="N"), dict()),
    (dict(flavor="lower"), dict()),
    (dict(flavor="iupac"), dict()),
    (dict(), dict(mode=1)),
    (dict(), dict(mode=2, max_insertions=2, max_deletions=2)),
    (dict(), dict(mask=1)),
    (dict(), dict(use_ratio=1)),
    (dict(flavor="sec"), dict()),
    (dict(p_sub=2e-2, p_ins=3e-3, p_del=3e-3), dict()),
    (dict(flavor="rep", p_ins=2e-3), dict()),
    (dict(p_sub=1e-2), dict(node_window=164)),
    (dict(), dict(start_grid=2)),
    (dict(), dict(start_grid=1024)),
    (dict(), dict(max_insertions=0, max_deletions=0)),
    (dict(), dict(max_insertions=1, max_deletions=1)),
    (dict(k=32), dict(max_deletions=10)),
    (dict(), dict(jump=1)),
    (dict(), dict(jump=5)),
    (dict(bfbytes=1 << 15), dict()),
    (dict(), dict(edit_threshold=25.0, missing_threshold=25.0)),
    (dict(), dict(edit_threshold=25.0, missing_threshold=25.0, node_window=170, mode=1)),
    (dict(flavor="N lower iupac sec rep"), dict(mode=1, mask=1)),
    (dict(hashes=4, bfbytes=100003 * 8), dict()),
    (dict(hashes=1, bfbytes=1 << 18), dict()),
    (dict(hashes=6, bfbytes=(1 << 18) + 8), dict()),
    (dict(flavor="exotic", p_sub=5e-3, p_ins=2e-3, p_del=2e-3), dict(max_deletions=10)),
    (dict(flavor="exotic N", p_sub=5e-3, p_ins=2e-3, p_del=2e-3), dict(mode=2, max_insertions=3, max_deletions=6)),
    # SNV mode: every position re-assessed
    (dict(n=8000, contigs=2), dict(snv=1)),
    (dict(n=8000, contigs=2, flavor="N iupac lower"), dict(snv=1, mode=2, mask=1)),
    (dict(n=8000, contigs=2, flavor="cbf"), dict(snv=1, min_threshold=2)),
    # counting Bloom filters (KmerCountingBloomFilter8): -p / -q thresholds, coverage medians
    (dict(flavor="cbf"), dict()),
    (dict(flavor="cbf"), dict(min_threshold=2)),
    (dict(flavor="cbf N iupac"), dict(min_threshold=3, max_threshold=4)),
    (dict(flavor="cbf", hashes=2, bfbytes=50021), dict(min_threshold=2, mode=1)),
    (dict(flavor="cbf sec"), dict(min_threshold=2, max_threshold=5, mode=2, max_insertions=2, max_deletions=2)),
    # event budget: speculative events parked early, the applied ones re-run to completion
    (dict(p_sub=2e-2, p_ins=3e-3, p_del=3e-3), dict(start_grid=2, event_budget=4)),
    (dict(flavor="rep", p_ins=2e-3), dict(start_grid=16, event_budget=40)),
    (dict(flavor="N lower sec"), dict(mode=1, start_grid=2, event_budget=9)),
    # a contig of exactly k bases (the 40-base "short" record): never seeded by the reference (ntedit.cpp:527)
    (dict(n=8000, contigs=2, k=40), dict(snv=1, mask=1, min_contig_len=0)),
    (dict(n=20000, contigs=2, k=40, flavor="N"), dict(mask=1, min_contig_len=0)),
    # counting filter / -s 1 with many subset k-mers: the per-lane assessment keeps at most 32 counts in registers
    # (k=64, -j 3: 22 of them; k=128: 43 -> the position-by-position paths; found by the GPU fuzz, seeds 91919100436 / ...753)
    (dict(n=7000, contigs=1, k=64, hashes=2, flavor="cbf", p_sub=2e-3, p_del=3e-4, bfbytes=2048), dict(snv=1, min_threshold=2, max_threshold=4)),
    (dict(n=7000, contigs=1, k=128, hashes=2, flavor="cbf", p_sub=2e-3, p_del=3e-4, bfbytes=2028), dict(snv=1, jump=1, min_threshold=2, max_threshold=4, missing_threshold=1.5, edit_threshold=25.0, max_insertions=2)),
    (dict(n=7000, contigs=1, k=128, hashes=1, flavor="N cbf", p_sub=2e-3, p_ins=2e-3, bfbytes=1024), dict(snv=1, min_threshold=3, use_ratio=1, missing_ratio=0.1, edit_ratio=0.1, max_deletions=3)),
    (dict(n=9000, contigs=2, k=96, flavor="cbf"), dict(min_threshold=2)),
]


def make_tail_case(tmp, k=31):
    """Contigs whose only / first accepted k-mer is the last possible k-mer start: the reference's
    findFirstAcceptedKmer (ntedit.cpp:527, i + k < size) never seeds there.  Visible with -s 1 -a 1."""
    os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(99)
    truth = random_genome(rng, 3000)
    write_fasta(os.path.join(tmp, "truth.fa"), [(b"t", truth)])
    mkbf([os.path.join(tmp, "truth.fa")], os.path.join(tmp, "t.bf"), k=k, hashes=3, nbytes=1 << 14)
    g = lambda n: random_genome(rng, n)  # noqa: E731
    draft = [
        (b"exact", g(k)),                                  # one k-mer, never seeded
        (b"n_then_k", b"NNNNN" + g(k)),                    # no earlier accepted k-mer
        (b"short_n_k", g(k - 1) + b"N" + g(k)),            # prefix too short for a k-mer
        (b"long_n_k", truth[100:400] + b"N" + g(k)),       # earlier k-mers exist: the tail IS reached by rolling
        (b"k_plus_1", g(k + 1)),                           # two k-mers: the first is seeded, the second rolled into
        (b"n_k_plus_1", b"N" + g(k + 1)),
        (b"truth_tail", truth[500:900]),
    ]
    write_fasta(os.path.join(tmp, "draft.fa"), draft, width=60)
    return {"draft": os.path.join(tmp, "draft.fa"), "bf": os.path.join(tmp, "t.bf"), "rep": None}


def make_contig_end_case(tmp, k=25, seed=11):
    """Errors planted at every distance 0 .. 2k + 14 from a contig's end (substitutions, 1-3 base deletions and insertions,
    pairs of errors a few bases apart): the positions in front of a contig's end are where the character window of a
    failing position is cut short (nte_machine_sweeps.inc: win_rolls) or not available at all (the last k)."""
    os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(seed)
    truth = random_genome(rng, 300000)
    write_fasta(os.path.join(tmp, "truth.fa"), [(b"t", truth)])
    mkbf([os.path.join(tmp, "truth.fa")], os.path.join(tmp, "t.bf"), k=k, hashes=3, nbytes=1 << 20)
    draft = []
    n = 0
    for dist in range(0, 2 * k + 15):
        for kind in range(8):
            L = int(rng.integers(4 * k, 8 * k))
            st = int(rng.integers(0, len(truth) - L))
            d = bytearray(truth[st:st + L])
            at = L - 1 - dist

            def sub(q):
                d[q] = b"ACGT"[(b"ACGT".index(d[q]) + 1 + int(rng.integers(0, 3))) % 4]
            if kind == 0:
                sub(at)
            elif kind in (1, 2, 3):
                del d[at:at + kind]            # the draft lacks 1-3 bases: an insertion repairs it
            elif kind in (4, 5):
                d[at:at] = random_genome(rng, kind - 3)  # 1-2 extra bases: a deletion repairs it
            elif kind == 6:
                sub(at)
                if at >= 7:
                    sub(at - 7)
            else:
                sub(at)
                if at >= 3:
                    del d[at - 3:at - 2]
            draft.append((b"end%d" % n, bytes(d)))
            n += 1
    write_fasta(os.path.join(tmp, "draft.fa"), draft, width=0)
    return {"draft": os.path.join(tmp, "draft.fa"), "bf": os.path.join(tmp, "t.bf"), "rep": None}


def make_many_case(tmp, n_contigs=3000, mean_len=500, seed=7):
    """Fragmented-assembly shape: thousands of short contigs cut from one truth genome, with errors, some
    lowercase stretches and Ns (several renderer work units, events in most contigs)."""
    os.makedirs(tmp, exist_ok=True)
    rng = np.random.default_rng(seed)
    truth = random_genome(rng, 400000)
    write_fasta(os.path.join(tmp, "truth.fa"), [(b"t", truth)])
    mkbf([os.path.join(tmp, "truth.fa")], os.path.join(tmp, "t.bf"), k=25, hashes=3, nbytes=1 << 20)
    draft = []
    for i in range(n_contigs):
        L = int(rng.integers(mean_len // 3, mean_len * 2))
        st = int(rng.integers(0, len(truth) - L))
        d = bytearray(mutate(rng, truth[st:st + L], 4e-3, 6e-4, 6e-4))
        if i % 7 == 0 and len(d) > 120:
            d[40:90] = bytes(d[40:90]).lower()
        if i % 11 == 0 and len(d) > 200:
            d[150:153] = b"NNN"
        draft.append((b"frag%d" % i if i % 3 else b"frag%d note" % i, bytes(d)))
    write_fasta(os.path.join(tmp, "draft.fa"), draft, width=0)
    return {"draft": os.path.join(tmp, "draft.fa"), "bf": os.path.join(tmp, "t.bf"), "rep": None}


def tsv_from_edits(recs, pool, names, header_line):
    """_changes.tsv text rebuilt from ntedit_hip_edit records (numpy, dtype ntedit_amd._lib.EDIT_DTYPE) and the
    base pool: the inverse of the C ABI's edit-record accessor, used to check it against the file the renderer
    writes (row formats: ntedit.cpp:1013-1029 insertions, 1113-1150 substitutions, 1173-1183 deletions)."""
    out = [header_line]
    for e in recs:
        kind = int(e["kind"])
        name = names[int(e["contig"])]
        if kind == 4:  # SNV mode: position kept, VCF-only
            continue
        if kind == 1:
            row = [name, b"%d" % (int(e["draft_pos"]) + 1), bytes([int(e["draft_base"])]), bytes([int(e["new_base"])]),
                   b"%d" % int(e["support"])]
            for j in range(int(e["n_alt"])):
                row += [bytes([int(e["alt_base"][j])]), b"%d" % int(e["alt_support"][j])]
        else:
            bases = pool[int(e["bases_off"]):int(e["bases_off"]) + int(e["len"])]
            bases = bases.split(b"\0")[0]  # (the reference prints through c_str())
            row = [name, b"%d" % int(e["draft_pos"]), bytes([int(e["draft_base"])]), (b"+" if kind == 2 else b"-") + bases,
                   b"%d" % int(e["support"])]
        out.append(b"\t".join(row) + b"\n")
    return b"".join(out)


def load_edits_dump(path):
    """the test-only dump of hostsim_set_render_extras(edits_path): u64 count | records | u64 pool bytes | pool"""
    from ntedit_amd._lib import EDIT_DTYPE
    raw = open(path, "rb").read()
    dt = np.dtype(EDIT_DTYPE)
    n = int(np.frombuffer(raw[:8], dtype="<u8")[0])
    recs = np.frombuffer(raw[8:8 + n * dt.itemsize], dtype=dt)
    o = 8 + n * dt.itemsize
    npool = int(np.frombuffer(raw[o:o + 8], dtype="<u8")[0])
    return recs, raw[o + 8:o + 8 + npool]


# ---- the reference's demo fixture (demo/ecoli_ntedit_k25_changes.tsv), row by row ------------------------------------------
# With the proxy filter (k-mers of the genome reconstructed from draft + changes.tsv, 2^28 bytes, h = 3) 4,986 of the
# reference's 4,997 rows come out byte for byte.  The other 11 are listed here, one by one, with what differs -- the real
# filter was built from READS, so it lacks k-mers the proxy has (lower support) and holds k-mers of read errors / variants
# the proxy lacks (alternate bases, higher support); at two loci that changes WHICH of two equivalent edits is found
# first (the edited genome is the same base for base, test_oracle_demo.py checks all of it).
DEMO_ROWS_IDENTICAL = 4986
# same position, same original base, same edit; only the support / alternate columns differ: (position, ref tail, our tail)
DEMO_ROWS_SUPPORT_ONLY = [
    ("225886", "A", "G", ["9", "T", "9"], ["9"]),
    ("275229", "T", "C", ["6"], ["9"]),
    ("984601", "C", "T", ["9", "G", "3"], ["9"]),
    ("1159420", "G", "A", ["5"], ["4"]),
    ("1784783", "C", "-CATT", ["3"], ["8"]),
    ("2917590", "T", "A", ["9", "C", "4"], ["9"]),
    ("3385647", "C", "G", ["9", "T", "9"], ["9"]),
]
# another, equivalent edit at the same locus (reference rows -> our rows)
DEMO_ROWS_EQUIVALENT = [
    ([("1411220", "C", "G", "3"), ("1411223", "C", "+CAC", "8")], [("1411219", "A", "+GCA", "8")]),
    ([("3085097", "A", "T", "3"), ("3085098", "T", "-T", "8")], [("3085096", "A", "-A", "8")]),
]


def check_demo_rows(ref_lines, got_lines):
    """Every one of the reference's 4,997 demo rows is accounted for: identical, or one of the 11 listed above with
    exactly the listed difference.  Anything else -- one more differing row, one listed row changing -- fails."""
    assert got_lines[0] == ref_lines[0]  # header line, byte for byte
    ref, got = ref_lines[1:], got_lines[1:]
    assert len(ref) == 4997
    rs, gs = set(ref), set(got)
    assert len(rs) == len(ref) and len(gs) == len(got)
    assert len(rs & gs) == DEMO_ROWS_IDENTICAL, len(rs & gs)
    name = ref[0].split("\t")[0]
    want_ref, want_got = set(), set()
    for pos, orig, new, rtail, gtail in DEMO_ROWS_SUPPORT_ONLY:
        want_ref.add("\t".join([name, pos, orig, new] + rtail))
        want_got.add("\t".join([name, pos, orig, new] + gtail))
    for rrows, grows in DEMO_ROWS_EQUIVALENT:
        want_ref.update("\t".join((name,) + r) for r in rrows)
        want_got.update("\t".join((name,) + g) for g in grows)
    assert rs - gs == want_ref, sorted(rs - gs)
    assert gs - rs == want_got, sorted(gs - rs)
    # the rows both sides have come in the same order
    common = rs & gs
    assert [r for r in ref if r in common] == [g for g in got if g in common]
