"""Host-side mirror of the reference's per-contig polishing call.

The reference exposes its hot path as
    kmerizeAndCorrect(hdr, seq, len, bloom, bloomrep, dfout, rfout, vfout, clinvar)
(ntedit.cpp:1747) driven by readAndCorrect (ntedit.cpp:2154).  `Polisher`
keeps that shape: load the filter(s) once, set the opt:: parameters, then hand
it batches of contigs; it returns / writes `_edited.fa` and `_changes.tsv`.
All compute happens in libntedit_hip.so on the GPU.
"""
import ctypes
import numpy as np

from . import _lib
from ._lib import NtEditHipError, Params, Stats, Segment, WriteOptions, EDIT_DTYPE

PRIMARY, SECONDARY = 0, 1


def default_params(**kw):
    lib = _lib.load()
    p = Params()
    lib.ntedit_hip_params_default(ctypes.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def pack_batch(records, min_contig_len=0):
    """records: iterable of (header_bytes, sequence_bytes).  Returns the batch layout of
    include/ntedit_hip.h: (blob, offsets u64, lens u32, names) with one '\\n' behind every contig.
    Contigs shorter than min_contig_len are dropped (ntedit.cpp:2242)."""
    names, offs, lens, parts = [], [], [], []
    pos = 0
    for name, seq in records:
        if len(seq) < min_contig_len:
            continue
        names.append(bytes(name))
        offs.append(pos)
        lens.append(len(seq))
        parts.append(bytes(seq))
        parts.append(b"\n")
        pos += len(seq) + 1
    return b"".join(parts), np.array(offs, dtype=np.uint64), np.array(lens, dtype=np.uint32), names


class Result:
    def __init__(self, lib, handle, owner=None):
        self._lib, self._h = lib, handle
        self._owner = owner  # keeps the Polisher (and its context) alive as long as the result is

    def stats(self):
        s = Stats()
        self._lib.ntedit_hip_result_stats(self._h, ctypes.byref(s))
        return s

    @staticmethod
    def _blob_ptr(blob):
        """(keep-alive object, void*) of a host batch: numpy array (no copy), bytearray (no copy) or bytes"""
        if isinstance(blob, np.ndarray):
            buf = np.ascontiguousarray(blob, dtype=np.uint8)
            return buf, ctypes.c_void_p(buf.ctypes.data)
        if isinstance(blob, bytearray):
            arr = (ctypes.c_char * max(len(blob), 1)).from_buffer(blob)
            return arr, ctypes.cast(arr, ctypes.c_void_p)
        buf = blob if isinstance(blob, bytes) else bytes(blob)
        return buf, ctypes.cast(ctypes.c_char_p(buf), ctypes.c_void_p)

    @staticmethod
    def _segment_array(segments, n):
        """segments: None or a sequence of (pos_offset, halo, flags) per entry"""
        if segments is None:
            return None
        arr = (Segment * max(n, 1))()
        for i, (off, halo, flags) in enumerate(segments):
            arr[i].pos_offset, arr[i].halo, arr[i].flags = int(off), int(halo), int(flags)
        return arr

    def write(self, blob, offsets, lens, names, fa_path, tsv_path, append=False, vcf_path=None, snv=False, annot=None,
              segments=None, want_sizes=False):
        """Render the batch (ntedit_hip_write_outputs_ex).  SNV mode follows the parameters the batch was polished
        with (`snv` is ignored).  segments: per-entry (pos_offset, halo, flags) for contigs cut into segments.
        want_sizes: return an (n, 3) uint64 array with the bytes every entry appended to fa / tsv / vcf."""
        n = len(names)
        arr = (ctypes.c_char_p * max(n, 1))(*names)
        buf, ptr = self._blob_ptr(blob)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        seg = self._segment_array(segments, n)
        sizes = np.zeros((max(n, 1), 3), dtype=np.uint64) if want_sizes else None
        wo = WriteOptions()
        wo.fa_path = fa_path.encode() if fa_path else None
        wo.tsv_path = tsv_path.encode() if tsv_path else None
        wo.vcf_path = vcf_path.encode() if vcf_path else None
        wo.append = 1 if append else 0
        wo.annot = annot
        wo.segments = ctypes.cast(seg, ctypes.c_void_p) if seg is not None else None
        wo.out_sizes = sizes.ctypes.data if sizes is not None else None
        rc = self._lib.ntedit_hip_write_outputs_ex(
            self._h, ptr, offsets.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p), arr, n,
            ctypes.byref(wo))
        if rc:
            raise NtEditHipError("write_outputs failed (%d)%s" % (
                rc, ": a segment's cut is not event-free" if rc == _lib.E_SEGMENT else ""))
        return sizes[:n] if sizes is not None else None

    def cover_ends(self, n_contigs):
        """per entry: where the serial run of its last applied event ended (0 = no applied event)"""
        out = np.zeros(max(n_contigs, 1), dtype=np.uint32)
        rc = self._lib.ntedit_hip_result_cover_ends(self._h, n_contigs, out.ctypes.data_as(ctypes.c_void_p))
        if rc:
            raise NtEditHipError("result_cover_ends failed (%d)" % rc)
        return out[:n_contigs]

    def cuts_ok(self, lens, segments):
        """per entry: will write() accept it with this (pos_offset, halo, flags)?  The renderer's own predicate
        (ntedit_hip_result_cuts_ok); an entry that fails is polished again joined with its successor."""
        n = len(lens)
        seg = self._segment_array(segments, n)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        ok = np.zeros(max(n, 1), dtype=np.uint8)
        rc = self._lib.ntedit_hip_result_cuts_ok(self._h, n, lens.ctypes.data_as(ctypes.c_void_p),
                                                 ctypes.cast(seg, ctypes.c_void_p), ok.ctypes.data_as(ctypes.c_void_p))
        if rc:
            raise NtEditHipError("result_cuts_ok failed (%d)" % rc)
        return ok[:n].astype(bool)

    def edits(self, blob, offsets, lens, segments=None):
        """(records, pool): every _changes.tsv row as a numpy record (dtype _lib.EDIT_DTYPE) plus the byte pool
        the inserted / deleted bases live in"""
        n = len(lens)
        buf, ptr = self._blob_ptr(blob)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        seg = self._segment_array(segments, n)
        ed, cnt, pool = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_void_p()
        rc = self._lib.ntedit_hip_result_edits(
            self._h, ptr, offsets.ctypes.data_as(ctypes.c_void_p), lens.ctypes.data_as(ctypes.c_void_p), n,
            ctypes.cast(seg, ctypes.c_void_p) if seg is not None else None, ctypes.byref(ed), ctypes.byref(cnt),
            ctypes.byref(pool))
        if rc:
            raise NtEditHipError("result_edits failed (%d)" % rc)
        dt = np.dtype(EDIT_DTYPE)
        if cnt.value == 0:
            return np.zeros(0, dtype=dt), b""
        recs = np.frombuffer(ctypes.string_at(ed.value, cnt.value * dt.itemsize), dtype=dt).copy()
        need = int((recs["bases_off"].astype(np.int64) + recs["len"]).max())
        return recs, ctypes.string_at(pool.value, need)

    def free(self):
        if self._h:
            self._lib.ntedit_hip_result_free(self._h)
            self._h = None

    def __del__(self):
        self.free()


class Polisher:
    def __init__(self, device=0):
        self._lib = _lib.load()
        h = ctypes.c_void_p()
        rc = self._lib.ntedit_hip_create(device, ctypes.byref(h))
        if rc:
            raise NtEditHipError("ntedit_hip_create(device=%d) failed (%d): no usable HIP device" % (device, rc))
        self._h = h
        self.params = default_params()

    def _check(self, rc, what):
        if rc:
            msg = self._lib.ntedit_hip_last_error(self._h)
            raise NtEditHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def close(self):
        # results hold page-locked buffers that belong to the context: free them first
        if self._h:
            self._lib.ntedit_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- filters
    def load_filter_file(self, path, slot=PRIMARY):
        self._check(self._lib.ntedit_hip_load_filter_file(self._h, slot, path.encode()), "load_filter_file")

    def set_filter(self, bits, hash_num, k, slot=PRIMARY, counting=False):
        bits = np.ascontiguousarray(bits, dtype=np.uint8)
        self._check(self._lib.ntedit_hip_set_filter(self._h, slot, bits.ctypes.data_as(ctypes.c_void_p),
                                                    bits.size, hash_num, k, int(counting)), "set_filter")

    def set_filter_device(self, device_ptr, nbytes, hash_num, k, slot=PRIMARY, counting=False):
        self._check(self._lib.ntedit_hip_set_filter_device(self._h, slot, ctypes.c_void_p(device_ptr), nbytes,
                                                           hash_num, k, int(counting)), "set_filter_device")

    def filter_alloc(self, nbytes, hash_num, k, slot=PRIMARY):
        self._check(self._lib.ntedit_hip_filter_alloc(self._h, slot, nbytes, hash_num, k), "filter_alloc")

    def filter_insert(self, bases, slot=PRIMARY, device_ptr=None, n=None):
        if device_ptr is not None:
            rc = self._lib.ntedit_hip_filter_insert(self._h, slot, ctypes.c_void_p(device_ptr), n, 1)
        else:
            rc = self._lib.ntedit_hip_filter_insert(self._h, slot, ctypes.cast(ctypes.c_char_p(bases), ctypes.c_void_p),
                                                    len(bases), 0)
        self._check(rc, "filter_insert")

    def filter_download(self, slot=PRIMARY):
        k, h, nb, cnt = self.filter_info(slot)
        out = np.empty(nb, dtype=np.uint8)
        self._check(self._lib.ntedit_hip_filter_download(self._h, slot, out.ctypes.data_as(ctypes.c_void_p)),
                    "filter_download")
        return out

    def filter_occupancy(self, slot=PRIMARY):
        """(occupied, slots): set bits / non-zero counters and the filter's size in slots"""
        occ, slots = ctypes.c_uint64(), ctypes.c_uint64()
        self._check(self._lib.ntedit_hip_filter_occupancy(self._h, slot, ctypes.byref(occ), ctypes.byref(slots)),
                    "filter_occupancy")
        return occ.value, slots.value

    def filter_save_file(self, path, slot=PRIMARY):
        self._check(self._lib.ntedit_hip_filter_save_file(self._h, slot, path.encode()), "filter_save_file")

    def filter_info(self, slot=PRIMARY):
        k, h, nb, cnt = ctypes.c_uint32(), ctypes.c_uint32(), ctypes.c_uint64(), ctypes.c_int()
        self._check(self._lib.ntedit_hip_filter_info(self._h, slot, ctypes.byref(k), ctypes.byref(h),
                                                     ctypes.byref(nb), ctypes.byref(cnt)), "filter_info")
        return k.value, h.value, nb.value, bool(cnt.value)

    def filter_device_ptr(self, slot=PRIMARY):
        return self._lib.ntedit_hip_filter_device_ptr(self._h, slot)

    # ---- parameters
    def set_params(self, params):
        self.params = params
        self._check(self._lib.ntedit_hip_set_params(self._h, ctypes.byref(params)), "set_params")

    # ---- hot path
    def screen(self, blob):
        """absent bitmap (np.uint64 words) for a host batch"""
        n = len(blob)
        out = np.zeros((n + 63) // 64, dtype=np.uint64)
        self._check(self._lib.ntedit_hip_screen(self._h, ctypes.cast(ctypes.c_char_p(blob), ctypes.c_void_p), n, 0,
                                                out.ctypes.data_as(ctypes.c_void_p)), "screen")
        return out

    def screen_device(self, bases_ptr, n, bitmap_ptr):
        self._check(self._lib.ntedit_hip_screen(self._h, ctypes.c_void_p(bases_ptr), n, 1,
                                                ctypes.c_void_p(bitmap_ptr)), "screen")
        return self._lib.ntedit_hip_last_kernel_ms(self._h)

    def pack_bases(self, blob, out=None, threads=0):
        """The packed form of a host batch (4-bit codes + case bits, include/ntedit_hip.h), or None when the batch holds a
        byte that form cannot carry.  out: a uint8 buffer of packed_size(len(blob)) bytes (e.g. page-locked)."""
        n = len(blob)
        size = int(self._lib.ntedit_hip_packed_size(n))
        if out is None:
            out = np.empty(size, dtype=np.uint8)
        assert out.nbytes >= size
        keep, ptr = Result._blob_ptr(blob)
        rc = self._lib.ntedit_hip_pack_bases(ptr, n, out.ctypes.data_as(ctypes.c_void_p), threads)
        del keep
        if rc < 0:
            raise NtEditHipError("pack_bases failed (%d)" % rc)
        return None if rc else out

    def packed_size(self, n):
        return int(self._lib.ntedit_hip_packed_size(n))

    def polish_batch(self, blob, offsets, lens, device_ptr=None, n=None, packed=None):
        """packed: the batch's packed form (pack_bases); it crosses PCIe instead of `blob`, whose length is still the batch's"""
        res = ctypes.c_void_p()
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        lens = np.ascontiguousarray(lens, dtype=np.uint32)
        if packed is not None:
            rc = self._lib.ntedit_hip_polish_batch(self._h, packed.ctypes.data_as(ctypes.c_void_p), len(blob),
                                                   offsets.ctypes.data_as(ctypes.c_void_p),
                                                   lens.ctypes.data_as(ctypes.c_void_p), len(lens), 2,
                                                   ctypes.byref(res))
        elif device_ptr is not None:
            rc = self._lib.ntedit_hip_polish_batch(self._h, ctypes.c_void_p(device_ptr), n,
                                                   offsets.ctypes.data_as(ctypes.c_void_p),
                                                   lens.ctypes.data_as(ctypes.c_void_p), len(lens), 1,
                                                   ctypes.byref(res))
        else:
            keep, ptr = Result._blob_ptr(blob)  # bytes, bytearray or a numpy array (e.g. a page-locked buffer)
            rc = self._lib.ntedit_hip_polish_batch(self._h, ptr, len(blob), offsets.ctypes.data_as(ctypes.c_void_p),
                                                   lens.ctypes.data_as(ctypes.c_void_p), len(lens), 0,
                                                   ctypes.byref(res))
            del keep
        self._check(rc, "polish_batch")
        return Result(self._lib, res, self)

    def write_tsv_header(self, tsv_path):
        """header line of _changes.tsv for the loaded primary filter / current parameters"""
        k, _, _, counting = self.filter_info(PRIMARY)
        rc = self._lib.ntedit_hip_write_tsv_header(tsv_path.encode(), k, self.params.jump, int(counting))
        if rc:
            raise NtEditHipError("cannot write %s" % tsv_path)

    def polish_records(self, records, out_prefix, draft_name="", annot_path=None):
        """readAndCorrect at -t 1 for an in-memory list of (header, sequence): writes
        <prefix>_edited.fa, <prefix>_changes.tsv and <prefix>_variants.vcf; returns Stats."""
        blob, offs, lens, names = pack_batch(records, self.params.min_contig_len)
        tsv = out_prefix + "_changes.tsv"
        fa = out_prefix + "_edited.fa"
        vcf = out_prefix + "_variants.vcf"
        self.write_tsv_header(tsv)
        self._lib.ntedit_hip_write_vcf_header(vcf.encode(), draft_name.encode())
        open(fa, "wb").close()
        annot = ctypes.c_void_p()
        if annot_path:
            if self._lib.ntedit_hip_annot_load(annot_path.encode(), ctypes.byref(annot)):
                raise NtEditHipError("cannot read %s" % annot_path)
        # (send_packed: the batch crosses PCIe as 4-bit codes + case bits when it can; the renderer keeps the bytes)
        packed = self.pack_bases(blob) if getattr(self, "send_packed", False) and len(blob) else None
        res = self.polish_batch(blob, offs, lens, packed=packed)
        res.write(blob, offs, lens, names, fa, tsv, append=True, vcf_path=vcf, snv=bool(self.params.snv), annot=annot)
        if annot_path:
            self._lib.ntedit_hip_annot_free(annot)
        st = res.stats()
        res.free()
        return st

    def reserve(self, max_batch_bytes, max_contigs=0, events_hint=0, on_device=0):
        """ntedit_hip_reserve: buffers for batches of up to max_batch_bytes + one internal warm-up batch, so that the
        first polish_batch of this context costs what a warm one does (on_device: 0 host, 1 device, 2 packed batches)"""
        self._check(self._lib.ntedit_hip_reserve(self._h, int(max_batch_bytes), int(max_contigs), int(events_hint),
                                                 int(on_device)), "reserve")

    def device_tables(self):
        """ntedit_hip_device_tables: the reference's candidate tables as the device code holds them, as text"""
        buf = ctypes.create_string_buffer(1 << 16)
        n = ctypes.c_uint64()
        self._check(self._lib.ntedit_hip_device_tables(self._h, buf, ctypes.c_uint64(len(buf)), ctypes.byref(n)), "device_tables")
        return buf.raw[:n.value].decode()

    def set_tuning(self, key, value):
        """Test / tuning knobs (include/ntedit_hip.h: none of them can change a result)."""
        self._check(self._lib.ntedit_hip_set_tuning(self._h, key.encode(), int(value)), "set_tuning")

    def gather_bench(self, nbytes, n_probes):
        pps, ms = ctypes.c_double(), ctypes.c_float()
        self._check(self._lib.ntedit_hip_gather_bench(self._h, nbytes, n_probes, ctypes.byref(pps), ctypes.byref(ms)),
                    "gather_bench")
        return pps.value, ms.value
