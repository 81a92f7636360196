"""Multi-GPU layout of the hot path: one process per GPU (torch.distributed,
backend "nccl" = RCCL on ROCm, "gloo" on CPU for the tests).

The path shards embarrassingly: contigs are independent (ntedit.cpp:2213-2252
hands one contig to each OpenMP thread), so the only data-path collective is
ONE broadcast of the Bloom filter bit array from rank 0 at start-up; after that
ranks never talk until the host-side gather of per-shard outputs, which is
concatenated back in input order (= the reference at -t 1).

Work units are PIECES.  A contig is one piece unless it is longer than a GPU's
fair share; then it is cut into segments at event-free boundaries (see
include/ntedit_hip.h, ntedit_hip_segment): the reference's serial run over a
contig (kmerizeAndCorrect, ntedit.cpp:1747-2151) can be cut wherever it is in
its clean state, and the library verifies that from the edit records.  Pieces
are spread over the ranks by bases (greedy LPT); every rank renders its pieces
into <prefix>.shard<r>_* plus an index of byte counts, and rank 0 stitches the
files together by that index.  Nothing here depends on the GPU: the compute
backend is an object with screen() and polish() (ntedit_amd.run.HipBackend for
the real thing, the test-only host simulation in tests/dist_worker.py)."""
import json
import os

import numpy as np

SEG_NO_HEADER, SEG_NO_NEWLINE, SEG_SKIP = 1, 2, 4

CUT_WINDOW = 1 << 16   # a cut is looked for in this many bases behind its nominal position
MAX_SEG_BASES = 16_000_000
MIN_SEG_BASES = 1_000_000


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    rank, world, local = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


# ---------------------------------------------------------------------------------- partition
def lpt_assign(weights, world):
    """Greedy longest-processing-time assignment: returns owner[i] for every item.  Deterministic
    (ties by index), so every rank computes the same answer without talking."""
    order = sorted(range(len(weights)), key=lambda i: (-int(weights[i]), i))
    load = [0] * world
    owner = [0] * len(weights)
    for i in order:
        r = min(range(world), key=lambda q: (load[q], q))
        owner[i] = r
        load[r] += int(weights[i])
    return owner


def shard_contigs(lens, world, min_len=0):
    """Greedy LPT partition of WHOLE contigs by bases (not by count): returns a list of index arrays, one per
    rank, each in input order.  This is also the split of the `ntedit --shard I/N` command line
    (ntedit_amd/host/main.cpp); run_sharded() below additionally cuts contigs that are larger than a share."""
    lens = np.asarray(lens, dtype=np.int64)
    idx = [i for i in range(len(lens)) if lens[i] >= min_len]
    owner = lpt_assign([lens[i] for i in idx], world)
    parts = [[] for _ in range(world)]
    for i, r in zip(idx, owner):
        parts[r].append(i)
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


def halo_bases(k, max_insertions, max_deletions):
    """Look-ahead room behind a cut.  While its k-mer head is at h the serial machine reads at most the
    next k k-mers (step 2, ntedit.cpp:1826-1873), shifted by the indel candidates (tryIndels / tryDeletion,
    ntedit.cpp:1451-1744): bases below h + 2k + max_del + max_ins + a few.  An event that ends in front of the
    cut therefore never sees the end of a buffer that reaches this far beyond it."""
    return 3 * int(k) + int(max_insertions) + int(max_deletions) + 64


def lead_bases(k):
    """k-mers in the filter the serial run gets in front of a cut (to come back to its clean state)"""
    return int(k) + 32


_ACGT = np.zeros(256, dtype=bool)
for _c in b"ACGTacgt":
    _ACGT[_c] = True


def refine_cut(seq, nominal, k, halo, screen_fn, window=CUT_WINDOW):
    """The first position c >= nominal + lead such that every k-mer that starts in [c - lead, c + halo - k] is
    made of ACGT and IS in the filter; None if [nominal, nominal + window) has no such stretch (or the contig
    ends there).  screen_fn(bytes) -> uint64 bitmap with bit i set iff the k-mer at i is accepted and absent
    (ntedit_hip_screen).  A function of the draft and the filter only: every rank finds the same cut."""
    lead = lead_bases(k)
    n = len(seq)
    lo = int(nominal)
    hi = min(n, lo + int(window) + k - 1)
    if hi - lo < lead + halo + k:
        return None
    win = seq[lo:hi]
    arr = np.frombuffer(win, dtype=np.uint8)
    n_kmers = len(arr) - k + 1
    bad = np.concatenate(([0], np.cumsum(~_ACGT[arr], dtype=np.int64)))
    valid = (bad[k:k + n_kmers] - bad[:n_kmers]) == 0
    bitmap = np.ascontiguousarray(screen_fn(bytes(win)), dtype=np.uint64)
    absent = np.unpackbits(bitmap.view(np.uint8), bitorder="little")[:n_kmers].astype(bool)
    good = valid & ~absent
    need = lead + halo - k + 1
    if n_kmers < need:
        return None
    cs = np.concatenate(([0], np.cumsum(good, dtype=np.int64)))
    full = np.nonzero(cs[need:] - cs[:n_kmers - need + 1] == need)[0]
    if full.size == 0:
        return None
    c = lo + int(full[0]) + lead
    if c + halo >= n:  # (the rest of the contig is shorter than the look-ahead room: not worth a cut)
        return None
    return c


class Piece:
    """bases [start, end) of contig `contig`; seg = its index among the contig's n_seg pieces"""
    __slots__ = ("contig", "seg", "n_seg", "start", "end", "owner")

    def __init__(self, contig, seg, n_seg, start, end):
        self.contig, self.seg, self.n_seg, self.start, self.end, self.owner = contig, seg, n_seg, start, end, 0

    def __repr__(self):
        return "Piece(c%d %d/%d [%d,%d) r%d)" % (self.contig, self.seg, self.n_seg, self.start, self.end, self.owner)


def auto_seg_bases(total, world):
    """pieces of at most an eighth of a rank's share (greedy LPT then balances the ranks to a few percent), within
    [MIN_SEG_BASES, MAX_SEG_BASES]; one rank: no cutting.  A cut costs one 64 KB screening call at planning time
    and ~150 bases of look-ahead halo, i.e. nothing."""
    if world <= 1:
        return 0
    return int(max(MIN_SEG_BASES, min(MAX_SEG_BASES, -(-total // (8 * world)))))


def plan_pieces(records, world, min_len, k, halo, screen_fn, seg_bases=None, refine=True):
    """records: [(header, seq)] known on every rank.  Returns the pieces of all contigs >= min_len in input
    order, owners assigned.  Contigs longer than 1.5 * seg_bases are cut into ceil(len / seg_bases) segments at
    refined, event-free positions; a nominal cut that cannot be refined is dropped (its two segments stay
    together).  refine=False (tests) keeps the nominal cuts: the library's verification then has to catch the
    bad ones."""
    lens = [len(s) for _, s in records]
    total = sum(l for l in lens if l >= min_len)
    if seg_bases is None:
        seg_bases = auto_seg_bases(total, world)
    pieces = []
    for ci, (_, seq) in enumerate(records):
        L = lens[ci]
        if L < min_len:
            continue
        cuts = []
        if seg_bases and L > seg_bases + seg_bases // 2:
            n = -(-L // seg_bases)
            for j in range(1, n):
                nominal = (L * j) // n
                if cuts and nominal <= cuts[-1]:
                    continue
                c = refine_cut(seq, nominal, k, halo, screen_fn) if refine else (
                    nominal if nominal + halo < L else None)
                if c is not None and (not cuts or c > cuts[-1]) and c < L:
                    cuts.append(c)
        bounds = [0] + cuts + [L]
        for j in range(len(bounds) - 1):
            pieces.append(Piece(ci, j, len(bounds) - 1, bounds[j], bounds[j + 1]))
    owner = lpt_assign([p.end - p.start for p in pieces], world)
    for p, r in zip(pieces, owner):
        p.owner = r
    return pieces


def piece_entry(records, p_first, p_last, halo):
    """(name, bases, (pos_offset, halo, flags)) of the batch entry that covers pieces p_first..p_last of one
    contig (a joined re-run covers more than one piece)"""
    hdr, seq = records[p_first.contig]
    last = p_last.seg == p_last.n_seg - 1
    h = 0 if last else halo
    flags = (SEG_NO_HEADER if p_first.seg > 0 else 0) | (0 if last else SEG_NO_NEWLINE)
    a, b = p_first.start, min(len(seq), p_last.end + h)
    # (a Draft's sequence: described now, read straight into the rank's batch buffer later -- ntedit_amd.run.LazySeq)
    return bytes(hdr), (seq.lazy(a, b) if hasattr(seq, "lazy") else seq[a:b]), (p_first.start, h, flags)


# ---------------------------------------------------------------------------------- the filter
def broadcast_filter_tensor(t, src=0):
    """The path's single collective: broadcast the filter bit array (uint8 tensor, on the
    GPU for nccl/RCCL, on the CPU for gloo) from rank `src` to every rank."""
    import torch.distributed as dist
    if dist.is_initialized() and dist.get_world_size() > 1:
        if t.is_cuda and dist.get_backend() == "gloo":
            # (rehearsals of the N > 1 driver on a box with fewer GPUs than ranks: gloo moves host memory)
            h = t.cpu()
            dist.broadcast(h, src=src)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src)
    return t


def _keep(polisher, slot, buf):
    if not hasattr(polisher, "_filter_keepalive"):
        polisher._filter_keepalive = {}
    polisher._filter_keepalive[slot] = buf


def shared_filter(polisher, nbytes, hash_num, k, slot=0):
    """Allocate the filter bit array as a torch tensor on this rank's GPU and let the library
    adopt it (ntedit_hip_set_filter_device).  Every rank calls this with the same geometry;
    rank `src` then fills it (filter_insert / copy_) and broadcast_filter() ships it.  This is the BUILD side
    (like btllib's constructor, the size is rounded up to whole 64-bit words)."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    nbytes = (int(nbytes) + 7) // 8 * 8
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(dev)
    polisher.set_filter_device(buf.data_ptr(), nbytes, hash_num, k, slot=slot)
    _keep(polisher, slot, buf)
    return buf


def broadcast_filter(buf, src=0):
    """The path's single data-path collective: one broadcast of the filter bit array
    (RCCL over xGMI).  `buf` is the tensor returned by shared_filter()."""
    import torch
    torch.cuda.synchronize()
    broadcast_filter_tensor(buf, src)
    torch.cuda.synchronize()
    return buf


def read_bf_header(path):
    """(meta dict, offset of the array) of a btllib .bf file"""
    meta = {}
    with open(path, "rb") as f:
        first = f.readline()
        if not first.startswith(b"[BTL") or b"BloomFilter" not in first:
            raise ValueError("%s: not a btllib Bloom filter file" % path)
        meta["counting"] = b"Counting" in first
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: header without [HeaderEnd]" % path)
            if line.startswith(b"[HeaderEnd]"):
                break
            if b"=" in line:
                key, val = [x.strip() for x in line.split(b"=", 1)]
                meta[key.decode()] = val.decode().strip('"')
        off = f.tell()
    for key in ("bytes", "hash_num", "k"):
        meta[key] = int(meta[key])
    return meta, off


def load_and_broadcast_filter(polisher, path, src=0, slot=0):
    """Rank `src` reads a btllib .bf file; every rank ends up with the same filter in HBM after ONE broadcast.
    The modulus of the slot arithmetic is the header's `bytes` exactly (btllib takes it as it is); only the
    allocation is padded to whole 64-bit words."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank() if dist.is_initialized() else 0
    dev = torch.device("cuda", torch.cuda.current_device())
    hdr = torch.zeros(4, dtype=torch.int64, device=dev)
    data = None
    if rank == src:
        meta, off = read_bf_header(path)
        nbytes = meta["bytes"]
        data = np.memmap(path, dtype=np.uint8, mode="r", offset=off, shape=(nbytes,))
        hdr[0], hdr[1], hdr[2], hdr[3] = meta["k"], meta["hash_num"], nbytes, int(meta["counting"])
    broadcast_filter_tensor(hdr, src)
    k, h, nbytes, counting = [int(x) for x in hdr.tolist()]
    pad = (nbytes + 7) // 8 * 8
    buf = torch.zeros(pad, dtype=torch.uint8, device=dev)
    if rank == src:
        # pinned staging in 256 MiB pieces: the file is read once, by one rank
        step = 256 << 20
        stage = torch.empty(min(step, nbytes), dtype=torch.uint8).pin_memory()
        for o in range(0, nbytes, step):
            n = min(step, nbytes - o)
            stage[:n].numpy()[:] = data[o:o + n]
            buf[o:o + n].copy_(stage[:n], non_blocking=False)
    broadcast_filter(buf, src)
    polisher.set_filter_device(buf.data_ptr(), nbytes, h, k, slot=slot, counting=bool(counting))
    _keep(polisher, slot, buf)
    return buf


# ---------------------------------------------------------------------------------- the sharded run
def shard_paths(out_prefix, rank):
    base = "%s.shard%d" % (out_prefix, rank)
    return base + "_edited.fa", base + "_changes.tsv", base + "_variants.vcf", base + ".index.json"


LAST_PHASES = {}  # host seconds of the last run_sharded on this rank: plan / polish (backend calls) / gather


def run_sharded(records, backend, out_prefix, min_len, rank, world, k, halo, write_headers, barrier=None,
                seg_bases=None, refine=True, all_gather=None):
    """records: [(header, seq)] known on every rank.  backend.screen(bytes) -> absent bitmap;
    backend.polish(entries, fa, tsv, vcf, append) -> (bad, sizes): polishes the batch of entries
    [(name, bases, (pos_offset, halo, flags))], renders those whose cut verifies (appending to the three files),
    returns the indices of the entries whose cut does NOT verify (nothing written for them) and the (n, 3) byte
    counts.  write_headers(prefix) creates <prefix>_edited.fa (empty), _changes.tsv and _variants.vcf with their
    header lines.  The shard files are gathered into <out_prefix>_* in input order: by every rank at once when
    all_gather (obj -> [obj of rank 0, 1, ...]) is given -- the per-piece byte counts travel, a few KB; each rank then
    copies ITS pieces to their places in the final files (gather_parallel) --, else by rank 0 (merge_shards).
    Returns this rank's pieces."""
    import time
    t_plan = time.perf_counter()
    pieces = plan_pieces(records, world, min_len, k, halo, backend.screen, seg_bases, refine)
    t_plan = time.perf_counter() - t_plan
    t_polish = time.perf_counter()
    by_contig = {}
    for p in pieces:
        by_contig.setdefault(p.contig, []).append(p)
    mine = [p for p in pieces if p.owner == rank]
    fa, tsv, vcf, idx_path = shard_paths(out_prefix, rank)
    for path in (fa, tsv, vcf):
        open(path, "wb").close()
    index = []  # [contig, first seg, last seg, fa bytes, tsv bytes, vcf bytes] in file order
    todo = [(p, p) for p in mine]
    while todo:
        entries = [piece_entry(records, a, b, halo) for a, b in todo]
        bad, sizes = backend.polish(entries, fa, tsv, vcf, True)
        bad = set(int(i) for i in bad)
        nxt = []
        for i, (a, b) in enumerate(todo):
            if i in bad:
                # The serial run was not clean at this cut (an edit chain ran into it): polish the segment
                # again together with its successor, whoever owns that one -- the gather drops the successor's
                # own output.  (b is never a contig's last piece here: those have no halo to violate.)
                nxt.append((a, by_contig[a.contig][b.seg + 1]))
            else:
                index.append([a.contig, a.seg, b.seg] + [int(x) for x in sizes[i]])
        todo = nxt
    with open(idx_path, "w") as f:
        json.dump(index, f)
    t_polish = time.perf_counter() - t_polish
    t_gather = time.perf_counter()
    if all_gather is not None and world > 1:
        gather_parallel(pieces, all_gather(index), rank, world, out_prefix, write_headers, barrier)
    else:
        if barrier:
            barrier()
        if rank == 0:
            merge_shards(pieces, world, out_prefix, write_headers)
    LAST_PHASES.clear()
    LAST_PHASES.update(plan=round(t_plan, 4), polish=round(t_polish, 4), gather=round(time.perf_counter() - t_gather, 4))
    return mine


def _copy_range(src, dst, n, bufsize=16 << 20):
    while n > 0:
        chunk = src.read(min(n, bufsize))
        if not chunk:
            raise IOError("shard file shorter than its index says")
        dst.write(chunk)
        n -= len(chunk)


def _where_from_indexes(indexes):
    """{(contig, first seg): (rank, last seg, offsets in the rank's shard files, byte counts)} from every rank's index;
    a joined entry (first seg < last seg) supersedes the single pieces it covers"""
    where = {}
    for r, index in enumerate(indexes):
        off = [0, 0, 0]
        for ci, s0, s1, nf, nt, nv in index:
            key = (ci, s0)
            if key not in where or where[key][1] < s1:
                where[key] = (r, s1, tuple(off), (nf, nt, nv))
            off[0] += nf
            off[1] += nt
            off[2] += nv
    return where


def _copy_at(src_fd, src_off, dst_fd, dst_off, n, bufsize=32 << 20):
    """n bytes from one file to their place in another (in the kernel where it can: copy_file_range)"""
    while n > 0:
        step = min(n, bufsize)
        done = 0
        if hasattr(os, "copy_file_range"):
            try:
                done = os.copy_file_range(src_fd, dst_fd, step, src_off, dst_off)
            except OSError:
                done = 0
        if done <= 0:
            chunk = os.pread(src_fd, step, src_off)
            if not chunk:
                raise IOError("shard file shorter than its index says")
            done = os.pwrite(dst_fd, chunk, dst_off)
        src_off += done
        dst_off += done
        n -= done


def gather_parallel(pieces, indexes, rank, world, out_prefix, write_headers, barrier):
    """The gather without a gatherer (VERDICT r5 "next" 5b): every rank knows every rank's index (all-gathered: a few KB),
    so every rank computes the same layout of the three final files -- header lines, then the pieces in input order,
    joined entries standing for the pieces they cover -- and copies the pieces IT rendered from its shard files to
    their offsets, all ranks at once.  Rank 0 writes the header lines and sizes the files first; nobody reads
    another rank's bytes.  Same bytes as merge_shards (which stays for `ntedit --shard` runs: tests compare them)."""
    where = _where_from_indexes(indexes)
    suffixes = ("_edited.fa", "_changes.tsv", "_variants.vcf")
    # layout: [(rank, source offsets, sizes)] in output order
    order = []
    i = 0
    while i < len(pieces):
        p = pieces[i]
        key = (p.contig, p.seg)
        if key not in where:
            raise RuntimeError("no shard rendered %r" % (p,))
        r, s1, off, sz = where[key]
        order.append((r, off, sz))
        i += s1 - p.seg + 1
    if rank == 0:
        write_headers(out_prefix)
        head = [os.path.getsize(out_prefix + s) for s in suffixes]
        for s in range(3):
            with open(out_prefix + suffixes[s], "r+b") as f:
                f.truncate(head[s] + sum(sz[s] for _, _, sz in order))
    if barrier:
        barrier()
    total = [sum(sz[s] for _, _, sz in order) for s in range(3)]
    head = [os.path.getsize(out_prefix + suffixes[s]) - total[s] for s in range(3)]
    srcs = [os.open(path, os.O_RDONLY) for path in shard_paths(out_prefix, rank)[:3]]
    dsts = [os.open(out_prefix + s, os.O_WRONLY) for s in suffixes]
    try:
        at = list(head)
        for r, off, sz in order:
            if r == rank:
                for s in range(3):
                    _copy_at(srcs[s], off[s], dsts[s], at[s], sz[s])
            for s in range(3):
                at[s] += sz[s]
    finally:
        for fd in srcs + dsts:
            os.close(fd)
    if barrier:
        barrier()
    for path in shard_paths(out_prefix, rank):
        if os.path.exists(path):
            os.remove(path)


def merge_shards(pieces, world, out_prefix, write_headers):
    """Host-side gather: every rank's index lists, in file order, the pieces it rendered with their byte counts in
    the three shard files; the final files are the pieces in input order.  Pieces are keyed by (contig ordinal,
    segment), never by header text (contig names need not be unique).  A joined entry (first seg < last seg)
    supersedes the single pieces it covers."""
    where = {}
    for r in range(world):
        _, _, _, idx_path = shard_paths(out_prefix, r)
        with open(idx_path) as f:
            index = json.load(f)
        off = [0, 0, 0]
        for ci, s0, s1, nf, nt, nv in index:
            key = (ci, s0)
            if key not in where or where[key][1] < s1:
                where[key] = (r, s1, tuple(off), (nf, nt, nv))
            off[0] += nf
            off[1] += nt
            off[2] += nv
    write_headers(out_prefix)
    outs = [open(out_prefix + s, "ab") for s in ("_edited.fa", "_changes.tsv", "_variants.vcf")]
    ins = [[open(path, "rb") for path in shard_paths(out_prefix, r)[:3]] for r in range(world)]
    try:
        i = 0
        while i < len(pieces):
            p = pieces[i]
            key = (p.contig, p.seg)
            if key not in where:
                raise RuntimeError("no shard rendered %r" % (p,))
            r, s1, off, sz = where[key]
            for s in range(3):
                ins[r][s].seek(off[s])
                _copy_range(ins[r][s], outs[s], sz[s])
            i += s1 - p.seg + 1  # (a joined entry also stands for the pieces behind it)
    finally:
        for f in outs:
            f.close()
        for fs in ins:
            for f in fs:
                f.close()
    for r in range(world):
        for path in shard_paths(out_prefix, r):
            if os.path.exists(path):
                os.remove(path)
