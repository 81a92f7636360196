"""ctypes binding of include/ntedit_hip.h (libntedit_hip.so, built in-tree).

There is deliberately no fallback: if the HIP library is missing or no GPU is
visible, every compute call raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NTEDIT_HIP_LIB") or os.path.join(_HERE, "libntedit_hip.so")  # (override: A/B builds)


class NtEditHipError(RuntimeError):
    pass


class Params(ctypes.Structure):
    """ntedit_hip_params (the reference's opt:: block, ntedit.cpp:99-133)."""
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


class Stats(ctypes.Structure):
    _fields_ = [
        ("bases", ctypes.c_uint64),
        ("absent_kmers", ctypes.c_uint64),
        ("events", ctypes.c_uint64),
        ("events_deferred", ctypes.c_uint64),
        ("events_applied", ctypes.c_uint64),
        ("substitutions", ctypes.c_uint64),
        ("insertions", ctypes.c_uint64),
        ("deletions", ctypes.c_uint64),
        ("ms_screen", ctypes.c_float),
        ("ms_extract", ctypes.c_float),
        ("ms_machine", ctypes.c_float),
        ("ms_total", ctypes.c_float),
        ("screen_launches", ctypes.c_uint32),
        ("screen_binned", ctypes.c_uint32),
        ("ms_partition", ctypes.c_float),
        ("ms_probe", ctypes.c_float),
        ("events_skipped", ctypes.c_uint32),
        ("screen_chunks_direct", ctypes.c_uint32),
        ("screen_overflow_records", ctypes.c_uint64),
    ]


class Segment(ctypes.Structure):
    """ntedit_hip_segment: a batch entry that is one segment of a contig cut for multi-GPU sharding"""
    _fields_ = [("pos_offset", ctypes.c_uint32), ("halo", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("reserved", ctypes.c_uint32)]


SEG_NO_HEADER, SEG_NO_NEWLINE, SEG_SKIP = 1, 2, 4
E_SEGMENT = -7
EDIT_SUB, EDIT_INS, EDIT_DEL, EDIT_SNV_KEPT = 1, 2, 3, 4


class Edit(ctypes.Structure):
    """ntedit_hip_edit: one _changes.tsv row as a POD record"""
    _fields_ = [("contig", ctypes.c_uint32), ("draft_pos", ctypes.c_uint32), ("bases_off", ctypes.c_uint32),
                ("len", ctypes.c_uint16), ("support", ctypes.c_uint16), ("kind", ctypes.c_uint8),
                ("draft_base", ctypes.c_uint8), ("new_base", ctypes.c_uint8), ("n_alt", ctypes.c_uint8),
                ("alt_base", ctypes.c_uint8 * 3), ("alt_support", ctypes.c_uint8 * 3), ("reserved", ctypes.c_uint8 * 2)]


# the same record as a numpy dtype (28 bytes)
EDIT_DTYPE = [("contig", "<u4"), ("draft_pos", "<u4"), ("bases_off", "<u4"), ("len", "<u2"), ("support", "<u2"),
              ("kind", "u1"), ("draft_base", "u1"), ("new_base", "u1"), ("n_alt", "u1"), ("alt_base", "u1", (3,)),
              ("alt_support", "u1", (3,)), ("reserved", "u1", (2,))]


class WriteOptions(ctypes.Structure):
    """ntedit_hip_write_options"""
    _fields_ = [("fa_path", ctypes.c_char_p), ("tsv_path", ctypes.c_char_p), ("vcf_path", ctypes.c_char_p),
                ("append", ctypes.c_int), ("annot", ctypes.c_void_p), ("segments", ctypes.c_void_p),
                ("out_sizes", ctypes.c_void_p)]


# every symbol include/ntedit_hip.h declares
EXPORTS = [
    "ntedit_hip_params_default", "ntedit_hip_params_clamp", "ntedit_hip_create", "ntedit_hip_destroy",
    "ntedit_hip_last_error", "ntedit_hip_set_filter", "ntedit_hip_set_filter_device",
    "ntedit_hip_load_filter_file", "ntedit_hip_filter_info", "ntedit_hip_filter_device_ptr",
    "ntedit_hip_filter_alloc", "ntedit_hip_filter_insert", "ntedit_hip_filter_download",
    "ntedit_hip_filter_save_file", "ntedit_hip_set_params", "ntedit_hip_screen", "ntedit_hip_polish_batch",
    "ntedit_hip_result_free", "ntedit_hip_result_stats", "ntedit_hip_write_outputs",
    "ntedit_hip_write_tsv_header", "ntedit_hip_last_kernel_ms", "ntedit_hip_gather_bench",
    "ntedit_hip_annot_load", "ntedit_hip_annot_free", "ntedit_hip_write_vcf_header", "ntedit_hip_write_outputs_vcf",
    "ntedit_hip_set_host_threads", "ntedit_hip_filter_occupancy",
    "ntedit_hip_write_outputs_ex", "ntedit_hip_result_cover_ends", "ntedit_hip_result_edits",
    "ntedit_hip_host_alloc", "ntedit_hip_host_free", "ntedit_hip_bind_near_device", "ntedit_hip_set_tuning", "ntedit_hip_build_id", "ntedit_hip_device_tables", "ntedit_hip_packed_size", "ntedit_hip_pack_bases",
    "ntedit_hip_fasta_load", "ntedit_hip_fasta_open", "ntedit_hip_fasta_read", "ntedit_hip_fasta_count", "ntedit_hip_fasta_blob", "ntedit_hip_fasta_record",
    "ntedit_hip_fasta_free", "ntedit_hip_result_cuts_ok", "ntedit_hip_reserve",
]

_lib = None


def load():
    """Load libntedit_hip.so; raises NtEditHipError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NtEditHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(make -C ntedit_amd/csrc). There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp, u64, u32, ci = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    lib.ntedit_hip_params_default.argtypes = [ctypes.POINTER(Params)]
    lib.ntedit_hip_params_default.restype = None
    lib.ntedit_hip_params_clamp.argtypes = [ctypes.POINTER(Params), ctypes.c_char_p, ctypes.c_size_t]
    lib.ntedit_hip_params_clamp.restype = None
    lib.ntedit_hip_create.argtypes = [ci, ctypes.POINTER(vp)]
    lib.ntedit_hip_destroy.argtypes = [vp]
    lib.ntedit_hip_destroy.restype = None
    lib.ntedit_hip_last_error.argtypes = [vp]
    lib.ntedit_hip_last_error.restype = ctypes.c_char_p
    lib.ntedit_hip_set_filter.argtypes = [vp, ci, vp, u64, u32, u32, ci]
    lib.ntedit_hip_set_filter_device.argtypes = [vp, ci, vp, u64, u32, u32, ci]
    lib.ntedit_hip_load_filter_file.argtypes = [vp, ci, ctypes.c_char_p]
    lib.ntedit_hip_filter_info.argtypes = [vp, ci, ctypes.POINTER(u32), ctypes.POINTER(u32),
                                           ctypes.POINTER(u64), ctypes.POINTER(ci)]
    lib.ntedit_hip_filter_device_ptr.argtypes = [vp, ci]
    lib.ntedit_hip_filter_device_ptr.restype = vp
    lib.ntedit_hip_filter_alloc.argtypes = [vp, ci, u64, u32, u32]
    lib.ntedit_hip_filter_insert.argtypes = [vp, ci, vp, u64, ci]
    lib.ntedit_hip_filter_download.argtypes = [vp, ci, vp]
    lib.ntedit_hip_filter_save_file.argtypes = [vp, ci, ctypes.c_char_p]
    lib.ntedit_hip_set_params.argtypes = [vp, ctypes.POINTER(Params)]
    lib.ntedit_hip_screen.argtypes = [vp, vp, u64, ci, vp]
    lib.ntedit_hip_polish_batch.argtypes = [vp, vp, u64, vp, vp, u32, ci, ctypes.POINTER(vp)]
    lib.ntedit_hip_result_free.argtypes = [vp]
    lib.ntedit_hip_result_free.restype = None
    lib.ntedit_hip_result_stats.argtypes = [vp, ctypes.POINTER(Stats)]
    lib.ntedit_hip_write_outputs.argtypes = [vp, vp, vp, vp, ctypes.POINTER(ctypes.c_char_p), u32,
                                             ctypes.c_char_p, ctypes.c_char_p, ci]
    lib.ntedit_hip_write_tsv_header.argtypes = [ctypes.c_char_p, u32, u32, ci]
    lib.ntedit_hip_annot_load.argtypes = [ctypes.c_char_p, ctypes.POINTER(vp)]
    lib.ntedit_hip_annot_free.argtypes = [vp]
    lib.ntedit_hip_annot_free.restype = None
    lib.ntedit_hip_write_vcf_header.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
    lib.ntedit_hip_write_outputs_vcf.argtypes = [vp, vp, vp, vp, ctypes.POINTER(ctypes.c_char_p), u32,
                                                 ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ci, ci, vp]
    lib.ntedit_hip_write_outputs_ex.argtypes = [vp, vp, vp, vp, ctypes.POINTER(ctypes.c_char_p), u32,
                                                ctypes.POINTER(WriteOptions)]
    lib.ntedit_hip_result_cover_ends.argtypes = [vp, u32, vp]
    lib.ntedit_hip_result_edits.argtypes = [vp, vp, vp, vp, u32, vp, ctypes.POINTER(vp), ctypes.POINTER(u64),
                                            ctypes.POINTER(vp)]
    lib.ntedit_hip_host_alloc.argtypes = [ctypes.c_size_t]
    lib.ntedit_hip_host_alloc.restype = vp
    lib.ntedit_hip_host_free.argtypes = [vp]
    lib.ntedit_hip_host_free.restype = None
    lib.ntedit_hip_bind_near_device.argtypes = [ci]
    lib.ntedit_hip_filter_occupancy.argtypes = [vp, ci, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.ntedit_hip_set_host_threads.argtypes = [ctypes.c_uint]
    lib.ntedit_hip_set_host_threads.restype = None
    lib.ntedit_hip_last_kernel_ms.argtypes = [vp]
    lib.ntedit_hip_last_kernel_ms.restype = ctypes.c_float
    lib.ntedit_hip_gather_bench.argtypes = [vp, u64, u64, ctypes.POINTER(ctypes.c_double),
                                            ctypes.POINTER(ctypes.c_float)]
    lib.ntedit_hip_set_tuning.argtypes = [vp, ctypes.c_char_p, u64]
    lib.ntedit_hip_packed_size.argtypes = [u64]
    lib.ntedit_hip_packed_size.restype = u64
    lib.ntedit_hip_pack_bases.argtypes = [vp, u64, vp, ctypes.c_uint]
    lib.ntedit_hip_build_id.argtypes = []
    lib.ntedit_hip_build_id.restype = ctypes.c_char_p
    lib.ntedit_hip_reserve.argtypes = [vp, u64, u32, u64, ci]
    lib.ntedit_hip_result_cuts_ok.argtypes = [vp, u32, vp, vp, vp]
    lib.ntedit_hip_fasta_load.argtypes = [ctypes.c_char_p, u64, ctypes.c_uint, ctypes.POINTER(vp), ctypes.c_char_p,
                                          ctypes.c_size_t]
    lib.ntedit_hip_fasta_open.argtypes = lib.ntedit_hip_fasta_load.argtypes
    lib.ntedit_hip_fasta_read.argtypes = [vp, u64, u64, u64, vp]
    lib.ntedit_hip_fasta_count.argtypes = [vp]
    lib.ntedit_hip_fasta_count.restype = u64
    lib.ntedit_hip_fasta_blob.argtypes = [vp, ctypes.POINTER(u64)]
    lib.ntedit_hip_fasta_blob.restype = vp
    lib.ntedit_hip_fasta_record.argtypes = [vp, u64, ctypes.POINTER(vp), ctypes.POINTER(u64), ctypes.POINTER(u64),
                                            ctypes.POINTER(u64)]
    lib.ntedit_hip_fasta_free.argtypes = [vp]
    lib.ntedit_hip_fasta_free.restype = None
    _lib = lib
    return lib
