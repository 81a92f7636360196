/*
 * ntedit_hip.h -- C ABI of the MI355X-native ntEdit hot path.
 *
 * The reference has no plugin/FFI seam; its hot path is one C++ call,
 *     kmerizeAndCorrect(hdr, seq, len, bloom, bloomrep, dfout, rfout, vfout, clinvar)
 * (ntedit.cpp:1747-1757) made per contig from readAndCorrect's OpenMP loop
 * (ntedit.cpp:2242-2245), with its parameters in the opt:: globals
 * (ntedit.cpp:99-133).  This header is the boundary a maintainer would bind
 * in its place: plain C, plain pointers and sizes, no C++/torch types.
 * INTEGRATION.md shows the reference-side stub.
 *
 * All functions return 0 on success or a negative NTEDIT_E_* code;
 * ntedit_hip_last_error() gives a message.  The host maps a failure to the
 * reference's convention (`ntEdit: error: ...` on stderr, exit(EXIT_FAILURE),
 * ntedit.cpp:476-483,2442-2445).  Nothing here falls back to the CPU: without
 * a HIP device every compute entry point fails with NTEDIT_E_DEVICE.
 */
#ifndef NTEDIT_HIP_H
#define NTEDIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NTEDIT_OK 0
#define NTEDIT_E_ARG (-1)      /* bad argument / unsupported option        */
#define NTEDIT_E_DEVICE (-2)   /* HIP runtime error or no device           */
#define NTEDIT_E_NOFILTER (-3) /* primary Bloom filter not set             */
#define NTEDIT_E_OVERFLOW (-4) /* internal capacity exceeded after retries */
#define NTEDIT_E_IO (-5)       /* file could not be read / written         */
#define NTEDIT_E_UNSUPPORTED (-6) /* operation not available (e.g. GPU build of a counting filter) */
#define NTEDIT_E_SEGMENT (-7)  /* a contig segment's cut is not event-free (see ntedit_hip_segment)      */
#define NTEDIT_E_INTERNAL (-8) /* an invariant of the library does not hold (the renderer: an event reached across the margin a
                                  contig's parts are cut with); the output files are incomplete -- a bug to report, not an I/O error */

#define NTEDIT_FILTER_PRIMARY 0   /* -r  (ntedit.cpp:2438) */
#define NTEDIT_FILTER_SECONDARY 1 /* -e  (ntedit.cpp:2570) */

typedef struct ntedit_hip_ctx ntedit_hip_ctx;
typedef struct ntedit_hip_result ntedit_hip_result;
typedef struct ntedit_hip_annot ntedit_hip_annot; /* -l annotation map, see ntedit_hip_annot_load() */

/* The opt:: parameter block (ntedit.cpp:99-133).  k and h are taken from the
 * primary filter (ntedit.cpp:2439,2448), not from -k. */
typedef struct ntedit_hip_params
{
	uint32_t min_contig_len;   /* -z (host-side drop, ntedit.cpp:2242) */
	uint32_t max_insertions;   /* -i */
	uint32_t max_deletions;    /* -d */
	float edit_threshold;      /* -y */
	float missing_threshold;   /* -x */
	float edit_ratio;          /* -Y */
	float missing_ratio;       /* -X */
	int32_t use_ratio;         /* set when -X or -Y was given */
	uint32_t jump;             /* -j */
	int32_t mode;              /* -m */
	int32_t snv;               /* -s */
	int32_t mask;              /* -a */
	uint32_t min_threshold;    /* -p (counting filters only; forced to 1 for plain filters) */
	uint32_t max_threshold;    /* -q (counting filters only) */
	/* tuning, not part of the reference surface (0 = default) */
	uint32_t start_grid;       /* extra event start every N positions in an absent run (power of 2) */
	uint32_t node_window;      /* rope nodes kept live per event thread */
	uint32_t screen_mode;      /* 0 auto, 1 direct gather kernel, 2 L2-partitioned (binned) pipeline */
	uint32_t event_budget;     /* positions a speculative event may walk before it is parked and, if it
	                              turns out to be applied, re-run to completion (0 = default 2048) */
} ntedit_hip_params;

/* defaults of ntedit.cpp:99-133 */
void ntedit_hip_params_default(ntedit_hip_params* p);
/* main()'s clamping after the filter is known (ntedit.cpp:2411-2413,2478-2493).
 * warn (may be NULL, cap bytes) receives the reference's warning texts. */
void ntedit_hip_params_clamp(ntedit_hip_params* p, char* warn, size_t cap);

int ntedit_hip_create(int device, ntedit_hip_ctx** out);
void ntedit_hip_destroy(ntedit_hip_ctx* ctx);
const char* ntedit_hip_last_error(const ntedit_hip_ctx* ctx);

/* ---- Bloom filters (replaces BFWrapper, ntedit.cpp:350-401) -------------
 * bits: the btllib bit array (LSB-first within a byte), nbytes a multiple of 8.
 * set_filter copies host memory to HBM; set_filter_device adopts a device
 * pointer owned by the caller (e.g. a buffer that was just RCCL-broadcast). */
int ntedit_hip_set_filter(
    ntedit_hip_ctx* ctx,
    int slot,
    const uint8_t* bits,
    uint64_t nbytes,
    uint32_t hash_num,
    uint32_t k,
    int counting);
int ntedit_hip_set_filter_device(
    ntedit_hip_ctx* ctx,
    int slot,
    void* device_bits,
    uint64_t nbytes,
    uint32_t hash_num,
    uint32_t k,
    int counting);
/* reads a btllib-format .bf file (header + raw array) straight into HBM */
int ntedit_hip_load_filter_file(ntedit_hip_ctx* ctx, int slot, const char* path);
/* filter geometry as loaded: k, hash_num, bytes, counting */
int ntedit_hip_filter_info(
    const ntedit_hip_ctx* ctx,
    int slot,
    uint32_t* k,
    uint32_t* hash_num,
    uint64_t* nbytes,
    int* counting);
void* ntedit_hip_filter_device_ptr(const ntedit_hip_ctx* ctx, int slot);

/* Build side (fixtures / benchmarks; mirrors src/ntedit_make_genome_bf.cpp:143-157):
 * allocate a zeroed filter in HBM, insert every all-ACGT k-mer of a sequence,
 * download / save it. */
int ntedit_hip_filter_alloc(ntedit_hip_ctx* ctx, int slot, uint64_t nbytes, uint32_t hash_num, uint32_t k);
int ntedit_hip_filter_insert(ntedit_hip_ctx* ctx, int slot, const char* bases, uint64_t n, int on_device);
/* occupied = set bits (plain) / non-zero counters (counting), slots = bits / counters:
 * btllib's get_fpr() is (occupied / slots)^hash_num (printed at ntedit_make_genome_bf.cpp:159) */
int ntedit_hip_filter_occupancy(ntedit_hip_ctx* ctx, int slot, uint64_t* occupied, uint64_t* slots);
int ntedit_hip_filter_download(const ntedit_hip_ctx* ctx, int slot, uint8_t* bits);
int ntedit_hip_filter_save_file(const ntedit_hip_ctx* ctx, int slot, const char* path);

int ntedit_hip_set_params(ntedit_hip_ctx* ctx, const ntedit_hip_params* p);

/* ---- hot path ------------------------------------------------------------
 * Batch layout: `bases` holds the contigs of the batch; contig i occupies
 * bases[offsets[i] .. offsets[i]+lens[i]) and every contig is followed by at
 * least one byte that is not an accepted base (the host driver uses '\n').
 * n = total bytes.  on_device: NTEDIT_HIP_BASES_HOST (0) `bases` is host memory, NTEDIT_HIP_BASES_DEVICE (1) it is
 * already in HBM, NTEDIT_HIP_BASES_PACKED (2; ntedit_hip_polish_batch only) it is host memory in the packed form below. */
#define NTEDIT_HIP_BASES_HOST 0
#define NTEDIT_HIP_BASES_DEVICE 1
#define NTEDIT_HIP_BASES_PACKED 2

/* The packed form of a batch: what crosses PCIe when the producer of the batch (a FASTA parser touches every byte
 * anyway) hands it over as 4-bit character codes instead of bytes -- 5 bits per base instead of 8 on a link that is
 * slower than the screening (3 GB of draft: ~110 ms at 27 GB/s against ~95 ms of screening).  Layout, for n bytes:
 *   codes   ceil(n / 32) * 16 bytes: two codes per byte, even position in the low nibble; 0..13 = the accepted bases
 *           A C G T R Y S W K M B D H V (either case), 15 = anything else (N, separators, ...)
 *   case    ceil(n / 128) * 16 bytes, right behind the codes: bit i (LSB first) = byte i is a lower-case letter
 * The device unpacks it into the byte batch every kernel reads (code 15 comes back as 'N' / 'n': to the hot path every
 * non-accepted byte is the same, ntedit.cpp:493-499), so results do not depend on the form.  Bytes that are NOT the
 * same to it -- U / u and the handful of other bytes whose ntHash seed is not zero (nte_common.h, is_exotic) -- cannot
 * be packed: ntedit_hip_pack_bases() then returns 1 and the caller hands the batch over as bytes.
 * `bases` stays the batch for ntedit_hip_write_outputs() either way (the renderer copies draft bytes).
 * threads: 0 = the ntedit_hip_set_host_threads() setting.  Returns 0, 1 (not packable) or NTEDIT_E_ARG. */
uint64_t ntedit_hip_packed_size(uint64_t n);
int ntedit_hip_pack_bases(const char* bases, uint64_t n, void* packed, unsigned threads);

/* Page-locked host memory for batches (optional): a batch handed over from such a buffer crosses PCIe
 * asynchronously, in pieces, while the pieces already in HBM are being screened.  Any other host memory
 * works too (staged by the runtime).  NULL when there is no device / no memory. */
void* ntedit_hip_host_alloc(size_t bytes);
void ntedit_hip_host_free(void* p);

/* Host placement (optional; the reference has no counterpart: its threads compute where the scheduler puts them).  Binds
 * the calling thread -- and the threads and first-touched pages it creates from then on -- to the CPUs of the NUMA node
 * the device hangs off (PCI bus id -> /sys/bus/pci/devices/<id>/numa_node).  A batch that crosses the socket
 * interconnect before it crosses PCIe costs the `ntedit` binary 15 % end to end on a two-socket host.  Returns the
 * node, or -1 when nothing was done (unknown topology, single node, NTEDIT_HIP_NO_BIND set). */
int ntedit_hip_bind_near_device(int device);

/* Start-up costs out of the first batch (optional; the reference's counterpart is what main() does before its
 * "reading/processing" stamp, ntedit.cpp:2589: loading the filters).  Call after the filter(s) and parameters are set:
 * sizes every grow-only device and page-locked buffer for batches of up to max_batch_bytes bytes / max_contigs contigs
 * (events_hint = expected event starts per batch, 0 = one per 400 bases; on_device = how the batches will arrive,
 * NTEDIT_HIP_BASES_*), and runs one small internal batch through the current configuration so that kernel code,
 * kernel attributes and scratch memory are in place.  Fresh device memory maps at ~40 GB/s (the 84 GB of screening
 * records of a 3 Gbp batch: two seconds); without this call the first ntedit_hip_polish_batch pays that, with it the
 * first call costs what a warm one does.  Never changes a result; may be called again when the configuration changes.
 * WHAT IT COSTS NOT TO CALL IT (measured, one MI355X): a context's first 3 Gbp batch takes ~2 s more (the record buffers
 * being mapped) and even with buffers already there its event machine runs 54-57 ms instead of 32 (kernel code objects,
 * scratch memory and workspaces set up inside the call: `[configs3]` against `[nonpow2]` in profiles/r6_gpu_tests*.log);
 * a 250 Mbp batch 22.9 ms instead of 13.4.  A caller that binds this ABI and times its first batch should call it. */
int ntedit_hip_reserve(ntedit_hip_ctx* ctx, uint64_t max_batch_bytes, uint32_t max_contigs, uint64_t events_hint, int on_device);

/* step 1 only (ntedit.cpp:1798-1807): bit i of bitmap (ceil(n/64) words,
 * host memory, or device memory when on_device) is set iff the k-mer starting
 * at byte i consists of accepted bases only and is NOT in the primary filter. */
int ntedit_hip_screen(
    ntedit_hip_ctx* ctx,
    const char* bases,
    uint64_t n,
    int on_device,
    uint64_t* bitmap);

/* steps 1-5 + makeEdit for every contig of the batch.  The result holds the
 * edit records; render it with ntedit_hip_write_outputs(). */
int ntedit_hip_polish_batch(
    ntedit_hip_ctx* ctx,
    const char* bases,
    uint64_t n,
    const uint64_t* offsets,
    const uint32_t* lens,
    uint32_t n_contigs,
    int on_device,
    ntedit_hip_result** out);
void ntedit_hip_result_free(ntedit_hip_result* r); /* (also legal after the context was destroyed) */

typedef struct ntedit_hip_stats
{
	uint64_t bases;          /* bytes screened                               */
	uint64_t absent_kmers;   /* set bits in the screening bitmap             */
	uint64_t events;         /* event threads launched                       */
	uint64_t events_deferred; /* events re-run by the sweep-only second launch */
	uint64_t events_applied; /* events that survive the serial-order filter  */
	uint64_t substitutions, insertions, deletions; /* rope/record counts     */
	float ms_screen;         /* HIP-event time of the screening launches (sum) */
	float ms_extract;        /* the run-map kernel (k_assess: -s 1, counting filters); 0 when it does not run */
	float ms_machine;        /* event machine launches (sum); overlaps screening when pipelined */
	float ms_total;          /* first kernel start -> edit records in host memory */
	uint32_t screen_launches; /* launches of the dominant screening kernel in this batch: k_bin_probe (binned
	                             screening, one per record chunk) or k_screen (direct; pipeline chunks / H2D pieces) */
	uint32_t screen_binned;   /* 1: binned pipeline (k_wc_scatter_b, k_bin_probe), 0: k_screen */
	float ms_partition;       /* binned: HIP-event time of the partition kernels (count + scan + scatter), sum */
	float ms_probe;           /* binned: HIP-event time of the k_bin_probe launches, sum */
	uint32_t events_skipped;  /* events not run because they start inside their cluster primary's run */
	uint32_t screen_chunks_direct; /* binned: record chunks whose overflow list ran out and that k_screen screened again
	                                  (a draft of very few distinct k-mers; 0 on anything like a genome) */
	uint64_t screen_overflow_records; /* binned: entries of the overflow list handed out (blocks of 256 per partition
	                                     wavefront): probes of repeated k-mers that did not fit their slice's run */
} ntedit_hip_stats;
int ntedit_hip_result_stats(const ntedit_hip_result* r, ntedit_hip_stats* s);

/* Host-side rendering of a result (replaces writeEditsToFile, ntedit.cpp:925-1213,
 * for _edited.fa and _changes.tsv; write_outputs_vcf() adds _variants.vcf).
 * bases/offsets/lens: the same batch, in HOST memory.  names[i] is the FASTA
 * header text (name + " " + comment, ntedit.cpp:2224-2229).  Files are opened
 * in append mode when append != 0; the TSV header is written by
 * ntedit_hip_write_tsv_header(). */
int ntedit_hip_write_outputs(
    const ntedit_hip_result* r,
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    const char* const* names,
    uint32_t n_contigs,
    const char* fa_path,
    const char* tsv_path,
    int append);
int ntedit_hip_write_tsv_header(const char* tsv_path, uint32_t k, uint32_t jump, int counting);

/* _variants.vcf (ntedit.cpp:951-977, 986-1162, 1184-1208; header 2192-2211) and the -l
 * annotation map (vcf_entry_to_map, ntedit.cpp:2261-2274; plain or gzipped input).
 * write_outputs_vcf = write_outputs + the VCF body.  SNV mode (unedited positions with supported
 * alternatives are VCF-only records, ntedit.cpp:1428-1443) follows the -s flag the batch was POLISHED
 * with; the `snv` argument is kept for source compatibility and ignored.
 * annot may be NULL (every annotation reads "NA"). */
int ntedit_hip_annot_load(const char* vcf_path, ntedit_hip_annot** out);
void ntedit_hip_annot_free(ntedit_hip_annot* a);
int ntedit_hip_write_vcf_header(const char* vcf_path, const char* draft_filename);
int ntedit_hip_write_outputs_vcf(
    const ntedit_hip_result* r,
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    const char* const* names,
    uint32_t n_contigs,
    const char* fa_path,
    const char* tsv_path,
    const char* vcf_path,
    int append,
    int snv,
    const ntedit_hip_annot* annot);

/* ---- contigs cut into segments (multi-GPU sharding of contigs larger than a GPU's share) -----------
 * The reference polishes a contig serially (kmerizeAndCorrect, ntedit.cpp:1747-2151; its only limit is
 * the 32-bit position, ntedit.cpp:1773-1774), so a single chromosome keeps one OpenMP thread busy while
 * the others idle.  Here a batch entry may be a SEGMENT [a, c) of a contig, handed over with `halo`
 * extra draft bases [c, c + halo) behind it as look-ahead room.  The serial run can be cut at c exactly
 * when it is in its clean state there (both rope cursors in the open position node, window = k untouched
 * draft bases): from then on its state is a function of the draft alone, and the next segment, polished
 * anywhere else, starts from the same state.  The caller places c inside a run of k-mers that are all in
 * the filter (ntedit_amd/dist.py: refine_cut); the library VERIFIES the cut from the edit records --
 * every applied event ended at or before c and nothing behind c was touched -- and refuses to render
 * the entry otherwise (NTEDIT_E_SEGMENT; ntedit_hip_result_cuts_ok() lets the caller check first and
 * re-run the segment joined with its successor).  Concatenating the segments' output is then byte-identical to
 * the unsplit contig's. */
#define NTEDIT_SEG_NO_HEADER 1u  /* not the first segment of its contig: no ">name" line                  */
#define NTEDIT_SEG_NO_NEWLINE 2u /* not the last segment: the sequence line stays open                    */
#define NTEDIT_SEG_SKIP 4u       /* write nothing for this entry (it was superseded by a joined re-run)    */
typedef struct ntedit_hip_segment
{
	uint32_t pos_offset; /* contig position of the entry's first base: added to every reported position */
	uint32_t halo;       /* trailing bases of the entry that belong to the next segment (not rendered) */
	uint32_t flags;      /* NTEDIT_SEG_* */
	uint32_t reserved;
} ntedit_hip_segment;

/* where the serial run of every entry's last applied event ended (entry-relative position; 0 = the entry
 * has no applied event).  A segment with a halo is valid iff cover_end <= lens[i] - halo. */
int ntedit_hip_result_cover_ends(const ntedit_hip_result* r, uint32_t n_contigs, uint32_t* cover_ends);
/* ok[i] = 1 iff write_outputs_ex() will accept entry i with segments[i] (the renderer's own predicate, evaluated
 * from the edit records without rendering: the last applied event ended at or before the cut, the rope was not
 * terminated and ends in the open position node, which starts in front of the cut).  Check BEFORE writing: an entry
 * that fails is polished again joined with its successor. */
int ntedit_hip_result_cuts_ok(const ntedit_hip_result* r, uint32_t n_contigs, const uint32_t* lens, const ntedit_hip_segment* segments, uint8_t* ok);

/* ---- edit records (the reference's per-contig rope + substitution queue, sRec / seqNode,
 * ntedit.cpp:599-620, flattened the way writeEditsToFile walks them, ntedit.cpp:936-1212) -------------
 * One record per _changes.tsv row, in file order; in SNV mode (-s 1) positions that keep their base but
 * have supported alternatives (VCF-only records) come as NTEDIT_EDIT_SNV_KEPT. */
#define NTEDIT_EDIT_SUB 1
#define NTEDIT_EDIT_INS 2
#define NTEDIT_EDIT_DEL 3
#define NTEDIT_EDIT_SNV_KEPT 4
typedef struct ntedit_hip_edit
{
	uint32_t contig;     /* entry index in the batch                                                      */
	uint32_t draft_pos;  /* 0-based draft position: SUB the base; INS the base the insertion precedes;
	                        DEL the first deleted base.  TSV column 2 = draft_pos + 1 (SUB) / draft_pos     */
	uint32_t bases_off;  /* INS / DEL: offset of the inserted / deleted bases in the pool                   */
	uint16_t len;        /* INS / DEL: number of bases; SUB: 1                                              */
	uint16_t support;    /* k-mers supporting the edit (TSV column 5)                                       */
	uint8_t kind;        /* NTEDIT_EDIT_*                                                                   */
	uint8_t draft_base;  /* TSV "OriginalBase"                                                              */
	uint8_t new_base;    /* SUB: the replacement                                                            */
	uint8_t n_alt;       /* SUB: alternate bases with support > 0                                           */
	uint8_t alt_base[3];
	uint8_t alt_support[3];
	uint8_t reserved[2];
} ntedit_hip_edit;
/* Builds (once, cached in the result) and returns the records of the whole batch.  bases/offsets/lens:
 * the batch in HOST memory, segments: NULL or the descriptors the batch will be rendered with.  The
 * pointers stay valid until ntedit_hip_result_free(). */
int ntedit_hip_result_edits(
    ntedit_hip_result* r,
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    uint32_t n_contigs,
    const ntedit_hip_segment* segments,
    const ntedit_hip_edit** edits,
    uint64_t* n_edits,
    const char** base_pool);

/* write_outputs with everything optional in one block.  SNV mode is taken from the parameters the batch
 * was polished with. */
typedef struct ntedit_hip_write_options
{
	const char* fa_path;  /* NULL: skip that stream */
	const char* tsv_path;
	const char* vcf_path;
	int append;
	const ntedit_hip_annot* annot;      /* -l map or NULL */
	const ntedit_hip_segment* segments; /* NULL, or one descriptor per entry */
	uint64_t* out_sizes;                /* NULL, or 3 * n_contigs: bytes every entry appended to fa / tsv / vcf
	                                       (the index the multi-GPU gather merges by) */
} ntedit_hip_write_options;
int ntedit_hip_write_outputs_ex(
    const ntedit_hip_result* r,
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    const char* const* names,
    uint32_t n_contigs,
    const ntedit_hip_write_options* opt);

/* Host threads used by write_outputs() to render contigs concurrently (the reference's -t;
 * output order and bytes do not depend on it).  0 = default (up to 8).  Process-wide.
 * A result may be rendered and freed on another thread than the one that runs
 * polish_batch() on the same context. */
void ntedit_hip_set_host_threads(unsigned n);

/* timings of the last screen()/filter_insert() call (HIP events, ms) */
float ntedit_hip_last_kernel_ms(const ntedit_hip_ctx* ctx);

/* random 1-byte gather micro-benchmark over a filter-sized buffer: the
 * "HBM random-read roofline" denominator of SURVEY.md 8(d).  Returns probes/s. */
int ntedit_hip_gather_bench(ntedit_hip_ctx* ctx, uint64_t nbytes, uint64_t n_probes, double* probes_per_s, float* ms);

/* ---- draft ingest (replaces kseq as used by readAndCorrect, ntedit.cpp:2213-2234; lib/kseq.h:176-215) ------------
 * Reads a whole FASTA / FASTQ draft (plain, gzip or BGZF -- told apart by their magic bytes) with the host binary's
 * readers and keeps the records with >= min_len bases (-z, ntedit.cpp:2242) as ONE buffer in the batch layout of
 * ntedit_hip_polish_batch: every sequence followed by '\n'.  header = name [+ " " + comment] (ntedit.cpp:2224-2229).
 * err (may be NULL) receives a message when the file cannot be opened or turns out to be corrupt / truncated
 * half-way (NTEDIT_E_IO; kseq would stop silently).  threads 0 = default. */
typedef struct ntedit_hip_fasta ntedit_hip_fasta;
int ntedit_hip_fasta_load(const char* path, uint64_t min_len, unsigned threads, ntedit_hip_fasta** out, char* err, size_t errcap);
/* The same records WITHOUT their bases: the file is mapped and indexed (headers, lengths), sequences are read on demand with
 * ntedit_hip_fasta_read -- what a rank of a multi-GPU run needs to plan the partition and then read its own share
 * (python -m ntedit_amd.run).  ntedit_hip_fasta_record() then reports offset = ~0 and ntedit_hip_fasta_blob() nothing.  Inputs
 * the mapped reader does not take (single-stream gzip, FASTQ, ...) are loaded whole; the calls behave the same. */
int ntedit_hip_fasta_open(const char* path, uint64_t min_len, unsigned threads, ntedit_hip_fasta** out, char* err, size_t errcap);
int ntedit_hip_fasta_read(const ntedit_hip_fasta* f, uint64_t i, uint64_t start, uint64_t n, char* dst);
uint64_t ntedit_hip_fasta_count(const ntedit_hip_fasta* f);
const char* ntedit_hip_fasta_blob(const ntedit_hip_fasta* f, uint64_t* nbytes);
int ntedit_hip_fasta_record(const ntedit_hip_fasta* f, uint64_t i, const char** header, uint64_t* header_len, uint64_t* offset, uint64_t* len);
void ntedit_hip_fasta_free(ntedit_hip_fasta* f);

/* Test and tuning knobs (not part of the reference surface).  NONE of them can change a result: they pick between
 * implementations that are bit-identical by construction (and tested to be), split work differently, or print
 * timings.  Keys:
 *   screening   "screen_mode" (overrides params.screen_mode), "bin_chunk" (k-mer starts per record chunk of the
 *               partitioned screening), "bin_cap_percent" (record-run capacity in percent of the expectation: forces the
 *               overflow list), "bin_ovf_cap" (entries of the overflow list: forces a list that runs out, i.e. record chunks
 *               screened again by the direct kernel), "bin_fallback" (1: the direct kernel, like "screen_mode" 1), "bin_scatter"
 *               (1: the barrier-free partition kernel, kept as the second implementation the tests compare),
 *               "force_xcc" (x + 1: the probe stage behaves as if every wavefront ran on XCD x), "bin_timing",
 *               "candmap" (1: with -s 1 on a plain filter the first probes of every position's substitution candidates go
 *               through the partitioned pipeline before k_assess; exact, measured slower, off), "h2d_fixed_schedule" (1: a
 *               host batch is screened in round 3's fixed chunk schedule instead of chunks sized by arrival)
 *   batches     "chunk_bytes" (pipeline chunk size), "h2d_piece" (bytes per host-to-device piece)
 *   machine     "inline_tries", "no_rounds", "force_rounds", "no_early_copy", "lanes" (runs of failing positions one
 *               position per lane: 0 off, 1 in the clean state, 2 also behind substitutions), "defer_fail" (failing positions after which the thread-per-event launch hands an event
 *               over; "defer_fail_snv": the same with -s 1, measured slower, 0), "snv_wave" (1: the events of -s 1 go to the
 *               wavefront-per-event launch; measured slower), "defer_run" (hand-over
 *               threshold of the thread-per-event launch), "assess" (the run map: 0 never, 1 always; default: with -s 1
 *               and counting filters), "machine_cfg" (0: the general instantiation of the machine kernels)
 * (The measured-and-rejected variants of round 3 -- record chunks partitioned while the previous one is probed, slices
 * probed in parts, uncached records, event rounds in pieces, a batch polished in pipeline chunks as it arrives -- are
 * gone from the library; DESIGN.md 8 keeps their numbers, the history their code.)
 * The library reads two environment variables only: NTEDIT_HIP_DEBUG (diagnostics on stderr) and
 * NTEDIT_HIP_NO_BIND (see ntedit_hip_bind_near_device). */
int ntedit_hip_set_tuning(ntedit_hip_ctx* ctx, const char* key, uint64_t value);

/* The reference's candidate tables -- num_tries, polish_bases_array / snv_bases_array, multi_possible_bases (ntedit.cpp:172,
 * 176-199, 203-348) -- as the device code holds them (one GPU thread runs the machine's own candidate_bases /
 * insertion_candidate), as text: "num_tries 0 1 5 21 85 341", "polish A TCG", ..., "snv N ATCG", "multi A A AA AC ...".
 * tests/ compare its SHA-256 per section with the hashes of the same text extracted from the reference's source
 * (tests/golden/reference_tables.json, tests/tools/reference_tables.py).  *len = bytes needed (without the 0). */
int ntedit_hip_device_tables(ntedit_hip_ctx* ctx, char* out, uint64_t cap, uint64_t* len);

/* Identifies what the library's kernels were built from (a hash of the device-side sources, set by the Makefile):
 * bench.py stamps the counter records it keeps under profiles/ with it and quotes them only for the same build.
 * No counterpart in the reference. */
const char* ntedit_hip_build_id(void);

#ifdef __cplusplus
}
#endif
#endif
