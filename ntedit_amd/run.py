"""python -m ntedit_amd.run -f draft.fa[.gz] -r solid.bf [-e repeat.bf] [-b prefix] [ntedit flags]

The multi-GPU driver of the hot path: one process per GPU, launched with

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        -m ntedit_amd.run -f draft.fa -r solid.bf -b out

(or plainly, without a launcher, on one GPU).  Rank 0 reads the Bloom filter file(s) once and ships the bit
arrays to every GPU with ONE RCCL broadcast each (dist.load_and_broadcast_filter); every rank maps the draft and
reads its INDEX (headers, lengths: Draft / ntedit_hip_fasta_open), computes the same partition of the contigs by bases --
contigs larger than a GPU's share are cut at event-free boundaries (dist.plan_pieces) --, reads the bases of ITS pieces
straight into its page-locked batch, polishes them through the C ABI (ntedit_amd.Polisher = libntedit_hip.so), writes
<prefix>.shard<r>_*, and -- after the per-piece byte counts have been all-gathered, a few KB -- copies its pieces to
their offsets in <prefix>_edited.fa / _changes.tsv / _variants.vcf itself (dist.gather_parallel): no rank holds the
draft, no rank reads another rank's output.  Byte-identical to the single-GPU `ntedit` binary's output (and to the
reference at -t 1).

What the reference does instead: readAndCorrect's OpenMP loop (ntedit.cpp:2213-2252), contigs handed to threads one
at a time, output in completion order."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

from . import dist as ndist
from .polisher import Polisher, default_params, pack_batch


def read_fasta_fast(path, min_len=0, threads=0):
    """[(header, sequence bytes)] of the records with >= min_len bases, read by the SAME code the `ntedit` binary
    ingests with (ntedit_hip_fasta_load: kseq's record semantics, lib/kseq.h:176-215 as used at
    ntedit.cpp:2223-2230 -- gzip / BGZF told apart by their magic bytes, FASTQ, CR line ends, NUL bytes, text in
    front of the first header).  Raises on an unreadable, corrupt or truncated file instead of returning a shorter
    draft."""
    from . import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    err = ctypes.create_string_buffer(512)
    rc = lib.ntedit_hip_fasta_load(os.fsencode(path), int(min_len), int(threads), ctypes.byref(h), err, len(err))
    if rc != 0:
        raise _lib.NtEditHipError("%s (%d)" % (err.value.decode(errors="replace") or "cannot read the draft", rc))
    try:
        nbytes = ctypes.c_uint64()
        base = lib.ntedit_hip_fasta_blob(h, ctypes.byref(nbytes))
        recs = []
        hp, hl, off, ln = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        for i in range(lib.ntedit_hip_fasta_count(h)):
            lib.ntedit_hip_fasta_record(h, i, ctypes.byref(hp), ctypes.byref(hl), ctypes.byref(off), ctypes.byref(ln))
            recs.append((ctypes.string_at(hp.value, hl.value) if hl.value else b"",
                         ctypes.string_at(base + off.value, ln.value) if ln.value else b""))
        return recs
    finally:
        lib.ntedit_hip_fasta_free(h)


class LazySlice:
    """bases [start, end) of a record of a Draft: read when somebody needs them (straight into a page-locked batch
    buffer with read_into(), or as bytes)"""
    __slots__ = ("draft", "rec", "start", "end")

    def __init__(self, draft, rec, start, end):
        self.draft, self.rec, self.start, self.end = draft, rec, int(start), int(end)

    def __len__(self):
        return self.end - self.start

    def read_into(self, address):
        self.draft.read_into(self.rec, self.start, self.end, address)

    def __bytes__(self):
        return self.draft.read(self.rec, self.start, self.end)


class LazySeq:
    """one record's sequence, sliceable: seq[a:b] reads those bases (the 64 KB windows cuts are looked for in),
    seq.lazy(a, b) only describes them"""
    __slots__ = ("draft", "rec", "n")

    def __init__(self, draft, rec, n):
        self.draft, self.rec, self.n = draft, rec, int(n)

    def __len__(self):
        return self.n

    def __getitem__(self, key):
        if not isinstance(key, slice):
            raise TypeError("LazySeq takes slices")
        a, b, step = key.indices(self.n)
        if step != 1:
            raise ValueError("LazySeq slices are contiguous")
        return self.draft.read(self.rec, a, max(a, b))

    def lazy(self, a, b):
        return LazySlice(self.draft, self.rec, a, b)


class Draft:
    """The draft as a multi-GPU rank needs it: the INDEX of the file -- headers and lengths, found by
    the library's mapped multi-threaded reader, ntedit_hip_fasta_open -- and its bases on demand.  Every rank plans the same
    partition from the lengths and then reads ITS OWN pieces (and the windows of its cuts) out of the mapped file;
    nobody holds the draft.  records() is the [(header, sequence)] list dist.run_sharded takes, with LazySeq
    sequences.  Inputs the mapped reader does not take (single-stream gzip, FASTQ, ...) are loaded whole by the
    library; nothing changes for the caller."""

    def __init__(self, path, min_len=0, threads=0):
        from . import _lib
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        err = ctypes.create_string_buffer(512)
        rc = self._lib.ntedit_hip_fasta_open(os.fsencode(path), int(min_len), int(threads), ctypes.byref(self._h), err, len(err))
        if rc != 0:
            raise _lib.NtEditHipError("%s (%d)" % (err.value.decode(errors="replace") or "cannot read the draft", rc))
        self.headers, self.lens = [], []
        hp, hl, off, ln = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        for i in range(self._lib.ntedit_hip_fasta_count(self._h)):
            self._lib.ntedit_hip_fasta_record(self._h, i, ctypes.byref(hp), ctypes.byref(hl), ctypes.byref(off), ctypes.byref(ln))
            self.headers.append(ctypes.string_at(hp.value, hl.value) if hl.value else b"")
            self.lens.append(int(ln.value))
        self.bytes_read = 0

    def records(self):
        return [(h, LazySeq(self, i, n)) for i, (h, n) in enumerate(zip(self.headers, self.lens))]

    def read_into(self, rec, start, end, address):
        if self._lib.ntedit_hip_fasta_read(self._h, rec, int(start), int(end - start), ctypes.c_void_p(address)) != 0:
            raise ValueError("draft record %d: bases [%d, %d) are out of range" % (rec, start, end))
        self.bytes_read += end - start

    def read(self, rec, start, end):
        buf = ctypes.create_string_buffer(max(1, end - start))
        self.read_into(rec, start, end, ctypes.addressof(buf))
        return buf.raw[:end - start]

    def close(self):
        if self._h:
            self._lib.ntedit_hip_fasta_free(self._h)
            self._h = ctypes.c_void_p()


class HipBackend:
    """dist.run_sharded's compute backend on the real thing: the HIP library through the C ABI"""

    def __init__(self, polisher, annot=None):
        self.pol = polisher
        self.annot = annot
        self.ms_gpu = 0.0
        self.bases = 0
        self.n_rerun = 0

    def screen(self, blob):
        return self.pol.screen(blob)

    def _pinned_batch(self, entries):
        """the batch layout of pack_batch, written straight into page-locked memory (ntedit_hip_host_alloc): the batch then
        crosses PCIe asynchronously, in pieces, under the screening -- from ordinary memory the runtime stages it through
        its own buffers first (a rank's 375 Mbp: ~35 ms against ~10)"""
        import numpy as np
        total = sum(len(e[1]) + 1 for e in entries)
        lib = self.pol._lib
        if total > getattr(self, "_pin_cap", 0):
            if getattr(self, "_pin_ptr", None):
                lib.ntedit_hip_host_free(ctypes.c_void_p(self._pin_ptr))
            self._pin_ptr = lib.ntedit_hip_host_alloc(total + total // 8 + 4096)
            self._pin_cap = total + total // 8 + 4096 if self._pin_ptr else 0
        if not getattr(self, "_pin_ptr", None):
            return None
        buf = np.ctypeslib.as_array((ctypes.c_uint8 * total).from_address(self._pin_ptr))
        offs, lens, pos = [], [], 0
        for _, seq, _ in entries:
            n = len(seq)
            if hasattr(seq, "read_into"):
                seq.read_into(self._pin_ptr + pos)  # (a piece of a Draft: mapped file -> page-locked batch, one copy)
            else:
                ctypes.memmove(self._pin_ptr + pos, seq, n)
            buf[pos + n] = 10
            offs.append(pos)
            lens.append(n)
            pos += n + 1
        return buf, np.array(offs, dtype=np.uint64), np.array(lens, dtype=np.uint32)

    def close(self):
        if getattr(self, "_pin_ptr", None):
            self.pol._lib.ntedit_hip_host_free(ctypes.c_void_p(self._pin_ptr))
            self._pin_ptr, self._pin_cap = None, 0

    def polish(self, entries, fa, tsv, vcf, append):
        names = [e[0] for e in entries]
        pinned = self._pinned_batch(entries)
        if pinned is not None:
            blob, offs, lens = pinned
        else:
            blob, offs, lens, _ = pack_batch([(e[0], bytes(e[1])) for e in entries], 0)
        res = self.pol.polish_batch(blob, offs, lens)
        # the renderer's own predicate, asked before anything is written (a refused entry would leave the shard
        # files half-written): a cut that is not event-free is polished again joined with its successor
        ok = res.cuts_ok(lens, [e[2] for e in entries])
        segs, bad = [], []
        for i, (_, _, (off, halo, flags)) in enumerate(entries):
            if not ok[i]:
                bad.append(i)
                flags |= ndist.SEG_SKIP
            segs.append((off, halo, flags))
        sizes = res.write(blob, offs, lens, names, fa, tsv, append=append, vcf_path=vcf, annot=self.annot,
                          segments=segs, want_sizes=True)
        st = res.stats()
        self.ms_gpu += st.ms_total
        self.bases += int(lens.sum())
        self.n_rerun += len(bad)
        res.free()
        return bad, sizes


def parse(argv=None):
    ap = argparse.ArgumentParser(prog="python -m ntedit_amd.run", description=__doc__,
                                 formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-f", dest="draft", required=True)
    ap.add_argument("-r", dest="bf", required=True)
    ap.add_argument("-e", dest="bfrep")
    ap.add_argument("-b", dest="prefix")
    ap.add_argument("-l", dest="annot")
    ap.add_argument("-z", dest="min_contig_len", type=int, default=100)
    ap.add_argument("-i", dest="max_insertions", type=int, default=5)
    ap.add_argument("-d", dest="max_deletions", type=int, default=5)
    ap.add_argument("-x", dest="missing_threshold", type=float, default=5.0)
    ap.add_argument("-y", dest="edit_threshold", type=float, default=9.0)
    ap.add_argument("-X", dest="missing_ratio", type=float)
    ap.add_argument("-Y", dest="edit_ratio", type=float)
    ap.add_argument("-j", dest="jump", type=int, default=3)
    ap.add_argument("-m", dest="mode", type=int, default=0)
    ap.add_argument("-s", dest="snv", type=int, default=0)
    ap.add_argument("-a", dest="mask", type=int, default=0)
    ap.add_argument("-p", dest="min_threshold", type=int, default=1)
    ap.add_argument("-q", dest="max_threshold", type=int, default=255)
    ap.add_argument("-t", dest="threads", type=int, default=0, help="host threads rendering the output")
    ap.add_argument("-k", dest="k_ignored", type=int, help="ignored: k comes from the filter")
    ap.add_argument("--seg-bases", type=int, default=None,
                    help="cut contigs longer than 1.5x this (default: an eighth of a GPU's share, 1-32 Mbp)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--report", action="store_true")
    return ap.parse_args(argv)


def main(argv=None):
    import ctypes
    import torch
    import torch.distributed as dist
    args = parse(argv)
    rank, world, local = ndist.env_rank()
    if not torch.cuda.is_available():
        sys.stderr.write("ntEdit v2.1.1: error: no HIP device (this build has no CPU path)\n")
        return 1
    if (args.backend or "nccl") != "nccl":
        # (rehearsal: `--backend gloo` lets N ranks share the GPUs that are there -- RCCL refuses two ranks on one device --
        # so the N > 1 path of this driver can be run on a one-GPU box; its timings then say nothing about N GPUs)
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if world > 1:
        ndist.init_process_group(args.backend or "nccl")
    elif args.backend:
        # a one-rank group still exercises the collective path (the round-end GPU test does this)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        dist.init_process_group(backend=args.backend, rank=0, world_size=1)
    pol = Polisher(local)
    pol._lib.ntedit_hip_bind_near_device(local)  # host threads and buffers on the socket this rank's GPU hangs off
    t0 = time.perf_counter()
    ndist.load_and_broadcast_filter(pol, args.bf, 0, 0)
    if args.bfrep:
        ndist.load_and_broadcast_filter(pol, args.bfrep, 0, 1)
    t_filter = time.perf_counter() - t0
    k, h, nbytes, counting = pol.filter_info(0)
    p = default_params(min_contig_len=args.min_contig_len, max_insertions=args.max_insertions,
                       max_deletions=args.max_deletions, missing_threshold=args.missing_threshold,
                       edit_threshold=args.edit_threshold, jump=args.jump, mode=args.mode, snv=args.snv,
                       mask=args.mask, min_threshold=args.min_threshold if counting else 1,
                       max_threshold=args.max_threshold)
    if args.missing_ratio is not None or args.edit_ratio is not None:
        p.use_ratio = 1
        if args.missing_ratio is not None:
            p.missing_ratio = args.missing_ratio
        if args.edit_ratio is not None:
            p.edit_ratio = args.edit_ratio
    warn = ctypes.create_string_buffer(1024)
    pol._lib.ntedit_hip_params_clamp(ctypes.byref(p), warn, 1024)  # ntedit.cpp:2478-2493
    if warn.value and rank == 0:
        sys.stderr.write(warn.value.decode())
    pol.set_params(p)
    if args.threads:
        pol._lib.ntedit_hip_set_host_threads(args.threads)
    prefix = args.prefix
    if not prefix:  # ntedit.cpp:2496-2502
        prefix = "%s_k%d_z%d_r%s_i%d_d%d_m%d" % (os.path.basename(args.draft), k, p.min_contig_len,
                                                  os.path.basename(args.bf), p.max_insertions, p.max_deletions, p.mode)
    annot = ctypes.c_void_p()
    if args.annot:
        if pol._lib.ntedit_hip_annot_load(args.annot.encode(), ctypes.byref(annot)):
            sys.stderr.write("Unable to open file\n")
            annot = ctypes.c_void_p()

    # the draft's index (headers, lengths); a rank reads the bases of its own pieces only (Draft)
    t0 = time.perf_counter()
    draft = Draft(args.draft)
    records = draft.records()
    t_read = time.perf_counter() - t0

    # start-up, like the filter load: this rank's buffers for its share of the draft + one internal warm-up batch, so that
    # the one and only polish call of a rank costs what a warm one does (ntedit_hip_reserve)
    t0 = time.perf_counter()
    total = sum(len(r[1]) for r in records)
    pol.reserve(int(total / max(1, world) * 1.15) + (1 << 20), 2 * len(records) + 64)
    t_reserve = time.perf_counter() - t0

    def write_headers(pre):
        open(pre + "_edited.fa", "wb").close()
        pol.write_tsv_header(pre + "_changes.tsv")
        pol._lib.ntedit_hip_write_vcf_header((pre + "_variants.vcf").encode(), args.draft.encode())

    backend = HipBackend(pol, annot if annot.value else None)
    halo = ndist.halo_bases(k, 0 if p.snv else p.max_insertions, 0 if p.snv else p.max_deletions)
    barrier = dist.barrier if (dist.is_initialized() and world > 1) else None

    def all_gather(obj):
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out
    t0 = time.perf_counter()
    mine = ndist.run_sharded(records, backend, prefix, p.min_contig_len, rank, world, k, halo, write_headers,
                             barrier=barrier, seg_bases=args.seg_bases,
                             all_gather=all_gather if barrier else None)
    if barrier:
        barrier()
    t_run = time.perf_counter() - t0
    if args.report:
        n_seg = sum(1 for q in mine if q.n_seg > 1)
        sys.stdout.write('{"rank": %d, "world": %d, "pieces": %d, "segments": %d, "bases": %d, "reruns": %d, '
                         '"gpu_ms": %.3f, "filter_s": %.3f, "index_s": %.3f, "draft_bases_total": %d, "draft_bytes_read": %d, '
                         '"reserve_s": %.3f, "run_s": %.3f, "phases_s": %s}\n' %
                         (rank, world, len(mine), n_seg, backend.bases, backend.n_rerun, backend.ms_gpu, t_filter,
                          t_read, total, draft.bytes_read, t_reserve, t_run, json.dumps(ndist.LAST_PHASES)))
        sys.stdout.flush()
    if annot.value:
        pol._lib.ntedit_hip_annot_free(annot)
    backend.close()
    draft.close()
    pol.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
