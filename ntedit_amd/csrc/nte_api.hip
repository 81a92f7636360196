// nte_api.hip -- implementation of the C ABI in include/ntedit_hip.h.
// Host side of the library: context / buffer management on one MI355X,
// kernel launches on the context's HIP stream, HIP-event timing, and the glue
// to the host renderer.  No CPU compute path exists here: every entry point
// that needs the GPU fails with NTEDIT_E_DEVICE when there is none.
#include "nte_kernels.hip"

#include "../../include/ntedit_hip.h"
#include "../host/bfio.h"
#include "../host/params.h"
#include "../host/render.h"
#include "../host/resolve.h"

#include <atomic>
#include <cctype>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <sched.h>
#include <string>
#include <vector>

using namespace nte;

namespace {

struct DevBuf
{
	void* p = nullptr;
	size_t cap = 0;
	bool uncached = false; // allocate as memory the L2 does not keep (hipDeviceMallocUncached)
};

struct DevFilter
{
	u8* data = nullptr;
	u64 nbytes = 0;
	u32 hash_num = 0, k = 0;
	bool counting = false, owned = false, set = false;
};

} // namespace

// page-locked host buffer (D2H at PCIe speed, no zero-fill); recycled through the context
static std::atomic<unsigned> g_host_threads(0); // ntedit_hip_set_host_threads()

// contexts that are alive: a result that outlives its context (a garbage-collected binding may free them in
// any order) must not hand its buffers back to a pool that is gone
static std::mutex g_live_mu;
static std::vector<const void*> g_live_ctx;

static bool
ctx_alive(const void* c)
{
	std::lock_guard<std::mutex> lk(g_live_mu);
	for (const void* p : g_live_ctx) {
		if (p == c) {
			return true;
		}
	}
	return false;
}

struct PinBuf
{
	void* p = nullptr;
	size_t cap = 0;
};

struct ntedit_hip_ctx
{
	int device = 0;
	hipStream_t stream = nullptr;
	hipStream_t stream2 = nullptr; // event extraction + machine of the chunk pipeline
	std::vector<hipEvent_t> chunk_ev; // 2 per chunk on `stream` (screen begin / end)
	std::vector<hipEvent_t> h2d_ev;   // one per host-to-device piece of a host-resident batch
	std::vector<hipEvent_t> bin_ev;   // 3 per record chunk of the binned screening (start, partitioned, probed)
	u32 bin_chunks_last = 0;          // chunks of the last binned screening (0: the direct kernel ran)
	u64 h2d_piece_bytes = 0;          // > 0: the batch is arriving from the host in pieces of this size on stream2
	u64 h2d_pieces = 0;               //      (h2d_ev[i] = piece i is in HBM); the binned screening waits chunk by chunk
	DevFilter filt[2];
	ntedit_hip_params hp;
	DevParams dp;
	bool dp_valid = false;
	u64* d_tab = nullptr;
	u32 tab_k = 0;
	std::string err;
	float last_ms = 0.f;
	hipEvent_t ev[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
	hipEvent_t ev_assess[2] = { nullptr, nullptr };
	DevBuf packed; // a batch as it crossed PCIe in the packed form (NTEDIT_HIP_BASES_PACKED), before k_unpack
	DevBuf runmap; // the absent bitmap minus the positions that cannot do anything (k_assess)
	DevBuf seq, bitmap, block_counts, block_offsets, events, first_chunk, arena, counters, deferred;
	DevBuf ws_nodes, ws_ov_pos, ws_ov_chr, ws_prev, ws_lps, ws_win;
	DevBuf offs, lens;
	DevBuf bin_records[2], bin_fill[2], bin_ctl[2], bin_ovf[2], bin_lost; // (two sets: chunk j + 1 is partitioned while chunk j is probed)
	hipStream_t stream3 = nullptr;    // the probe stage of the binned screening when it overlaps the partition stage
	hipStream_t stream_copy = nullptr; // H2D pieces of a host batch that is polished in pipeline chunks (stream2 runs the event machine then)
	bool bin_fallback = false;        // an overflow list overflowed: this context screens with the direct kernel from now on
	struct Tuning                     // ntedit_hip_set_tuning(): test / tuning knobs, none of which can change a result
	{
		u32 screen_mode = 0;     // overrides params.screen_mode when not 0
		u64 bin_chunk = 0;       // k-mer starts per record chunk of the binned screening (tests: several chunks)
		u32 bin_cap_percent = 0; // run capacity in percent of the expected records (tests: force the overflow list)
		u32 force_xcc = 0;       // x + 1: every probe wavefront pretends to run on XCD x (tests)
		u32 bin_timing = 0;      // per-stage times of the binned screening on stderr
		u64 chunk_bytes = 0;     // pipeline chunk size (tests: many chunks)
		u64 h2d_piece = ~0ULL;   // bytes per host-to-device piece (~0: default)
		u32 inline_tries = ~0u;  // candidates of an indel sweep the deferring launch tries itself (~0: default)
		u32 assess = ~0u;        // k_assess before the event machine: 0 never, 1 always, ~0: with -s 1 and with counting filters
		u32 machine_cfg = ~0u;   // 0: always the general instantiation of the machine kernels (~0: the most specific one)
		u32 lanes = ~0u;         // DevParams::lanes (~0: default)
		u32 defer_run = ~0u;     // DevParams::defer_run (~0: default)
		u32 screen_lds_pad = 0;  // LDS pad of the direct screening kernel (occupancy experiments)
		u32 no_rounds = 0, no_early_copy = 0, no_lds_ws = 0;
		u32 force_rounds = 0;     // event rounds whatever the number of events (tests: small inputs)
		u32 machine_pieces = 0;   // a round's list in this many pieces, the sweeps of piece i next to pass 1 of piece i + 1 (0 = automatic, 1 = off)
		u32 probe_parts_log2 = 0; // the probe stage walks every slice 2^n times, one part of it per walk
		u32 records_uncached = 0; // the screening records in memory the L2 does not keep (experiments)
		u32 bin_scatter = 0;      // partition kernel: 0 barrier-phased (k_wc_scatter_b), 1 barrier-free (k_wc_scatter)
		u32 bin_overlap = 0;      // partition chunk j + 1 while chunk j is probed (two record buffers, a second stream)
		u32 h2d_chunks = 0;       // a large batch in host memory is polished in this many pipeline chunks (0, 1: one)
	} tune;
	DevBuf ev_cover, ev_before, ev_flags, ev_list, ev_bmax; // event rounds
	u32 cu_count = 256;
	size_t lds_per_block = 160 * 1024;
	std::vector<PinBuf> pin_pool;
	std::mutex pin_mu;
};

struct ntedit_hip_result
{
	ntedit_hip_ctx* owner = nullptr;
	PinBuf arena_buf;      // arena copy (Items)
	size_t arena_items = 0;
	PinBuf first_buf;      // per-event first chunk, then compacted in place to ev_first
	size_t n_ev_first = 0;
	ntedit_hip_stats st;
	nte_host::RenderStats rst;
	int snv = 0; // -s of the parameters the batch was polished with
	// ntedit_hip_result_edits(): built on first use
	bool edits_built = false;
	std::vector<ntedit_hip_edit> edits;
	std::string edit_pool;
};

namespace {

int
fail(ntedit_hip_ctx* c, int code, const char* fmt, ...)
{
	char buf[512];
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(buf, sizeof buf, fmt, ap);
	va_end(ap);
	if (c) {
		c->err = buf;
	}
	return code;
}

#define HIP_TRY(ctx, expr)                                                                       \
	do {                                                                                         \
		hipError_t e_ = (expr);                                                                  \
		if (e_ != hipSuccess) {                                                                  \
			return fail((ctx), NTEDIT_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));         \
		}                                                                                        \
	} while (0)

// take a pinned buffer of at least `bytes` from the pool (or allocate one)
int
pin_take(ntedit_hip_ctx* c, size_t bytes, PinBuf* out);
void
pin_give(ntedit_hip_ctx* c, PinBuf& b);

int
ensure(ntedit_hip_ctx* c, DevBuf& b, size_t bytes)
{
	if (bytes <= b.cap) {
		return 0;
	}
	if (b.p) {
		HIP_TRY(c, hipFree(b.p));
		b.p = nullptr;
		b.cap = 0;
	}
	size_t want = bytes + bytes / 8 + 256;
	if (b.uncached) {
		HIP_TRY(c, hipExtMallocWithFlags(&b.p, want, hipDeviceMallocUncached));
	} else {
		HIP_TRY(c, hipMalloc(&b.p, want));
	}
	b.cap = want;
	return 0;
}

int
pin_take(ntedit_hip_ctx* c, size_t bytes, PinBuf* out)
{
	std::lock_guard<std::mutex> lk(c->pin_mu); // results are freed from other threads
	int best = -1;
	for (size_t i = 0; i < c->pin_pool.size(); i++) {
		if (c->pin_pool[i].cap >= bytes && (best < 0 || c->pin_pool[i].cap < c->pin_pool[best].cap)) {
			best = (int)i;
		}
	}
	if (best >= 0) {
		*out = c->pin_pool[best];
		c->pin_pool.erase(c->pin_pool.begin() + best);
		return 0;
	}
	// drop the oldest pooled buffer if the pool is getting large
	if (c->pin_pool.size() >= 4) {
		(void)hipHostFree(c->pin_pool.front().p);
		c->pin_pool.erase(c->pin_pool.begin());
	}
	size_t want = bytes + bytes / 4 + 4096;
	void* p = nullptr;
	HIP_TRY(c, hipHostMalloc(&p, want, hipHostMallocDefault));
	out->p = p;
	out->cap = want;
	return 0;
}

void
pin_give(ntedit_hip_ctx* c, PinBuf& b)
{
	if (!b.p) {
		return;
	}
	if (c && ctx_alive(c)) {
		std::lock_guard<std::mutex> lk(c->pin_mu);
		c->pin_pool.push_back(b);
	} else {
		(void)hipHostFree(b.p);
	}
	b.p = nullptr;
	b.cap = 0;
}

void
release(DevBuf& b)
{
	if (b.p) {
		(void)hipFree(b.p);
	}
	b.p = nullptr;
	b.cap = 0;
}

Filter
dev_filter(const DevFilter& f)
{
	Filter r;
	r.data = f.data;
	filter_set_size(r, f.counting ? f.nbytes : f.nbytes * 8); // counters vs bits
	r.hash_num = f.hash_num;
	r.counting = f.counting ? 1 : 0;
	return r;
}

int
refresh_params(ntedit_hip_ctx* c)
{
	const DevFilter& f = c->filt[0];
	if (!f.set) {
		return fail(c, NTEDIT_E_NOFILTER, "primary Bloom filter not set");
	}
	const DevFilter& r = c->filt[1];
	if (r.set) {
		if (r.k != f.k) {
			// ntedit.cpp:2581-2585
			return fail(
			    c,
			    NTEDIT_E_ARG,
			    "secondary Bloom filter k size (%u) is different than main Bloom filter k size (%u)",
			    r.k,
			    f.k);
		}
	}
	int rc = nte_host::make_dev_params(c->hp, f.k, f.hash_num, r.set, &c->dp, f.counting);
	if (rc) {
		return fail(c, rc, "unsupported parameter combination (k=%u h=%u)", f.k, f.hash_num);
	}
#ifdef NTE_ABLATION
	if (const char* e = getenv("NTEDIT_HIP_MACHINE_DEBUG")) {
		c->dp.debug_stop = (u32)atoi(e); // timing ablations; results are NOT valid (ablation build only)
	}
#endif
	if (c->tune.inline_tries != ~0u) { // tuning / tests (any value gives the same results)
		c->dp.inline_tries = c->tune.inline_tries;
	}
	if (c->tune.lanes != ~0u) {
		c->dp.lanes = c->tune.lanes;
	}
	if (c->tune.defer_run != ~0u) {
		c->dp.defer_run = c->tune.defer_run;
	}
	if (!c->d_tab) {
		HIP_TRY(c, hipMalloc((void**)&c->d_tab, TAB_WORDS * sizeof(u64)));
	}
	if (c->tab_k != f.k) {
		u64 tab[TAB_WORDS];
		build_seed_tables(f.k, tab);
		HIP_TRY(c, hipMemcpy(c->d_tab, tab, sizeof tab, hipMemcpyHostToDevice));
		c->tab_k = f.k;
	}
	c->dp_valid = true;
	return 0;
}

bool binned_applicable(const ntedit_hip_ctx* c, const Filter& f, u64 n, u32* slice_log2, u32* n_slices);
int bin_records_lost(ntedit_hip_ctx* c, bool* lost);
int run_screen_binned(ntedit_hip_ctx* c, const u8* d_seq, u64 n, const Filter& f, u64* d_bitmap, u64 n_words, u32 slog, u32 n_slices, hipStream_t stream = nullptr, u64 pos_begin = 0, u64 pos_end = ~0ULL);

// Launches k_screen over tiles [first_tile, first_tile + n_tiles) of the batch on `stream`.
// lds_pad > 0 lowers the kernel's occupancy (its speed does not depend on it: it is bound by
// the L2-miss path from 2 workgroups per CU upwards) so that k_machine can share the CUs.
template<bool INSERT>
int
launch_screen_tiles(
    ntedit_hip_ctx* c,
    hipStream_t stream,
    const u8* d_seq,
    u64 n,
    const Filter& f,
    u64* d_bitmap,
    u64 n_words,
    u64 first_tile,
    u64 n_tiles,
    size_t lds_pad)
{
	if (n_tiles == 0) {
		return 0;
	}
	if (n_tiles > 0x7FFFFFFFull) {
		return fail(c, NTEDIT_E_ARG, "batch too large");
	}
	dim3 grid((unsigned)n_tiles), block(SCREEN_TPB);
	const bool pow2 = f.mask != 0;
	if (c->tune.screen_lds_pad) { // tuning
		lds_pad = c->tune.screen_lds_pad;
	}
#define NTE_LAUNCH(H)                                                                            \
	do {                                                                                         \
		if (pow2) {                                                                              \
			hipLaunchKernelGGL(                                                                  \
			    (k_screen<H, true, INSERT>), grid, block, lds_pad, stream, d_seq, n, f, c->dp,   \
			    c->d_tab, d_bitmap, n_words, first_tile);                                        \
		} else {                                                                                 \
			hipLaunchKernelGGL(                                                                  \
			    (k_screen<H, false, INSERT>), grid, block, lds_pad, stream, d_seq, n, f, c->dp,  \
			    c->d_tab, d_bitmap, n_words, first_tile);                                        \
		}                                                                                        \
	} while (0)
	switch (f.hash_num) {
	case 1:
		NTE_LAUNCH(1);
		break;
	case 2:
		NTE_LAUNCH(2);
		break;
	case 3:
		NTE_LAUNCH(3);
		break;
	case 4:
		NTE_LAUNCH(4);
		break;
	case 5:
		NTE_LAUNCH(5);
		break;
	default:
		NTE_LAUNCH(0);
		break;
	}
#undef NTE_LAUNCH
	HIP_TRY(c, hipGetLastError());
	return 0;
}

template<bool INSERT>
int
launch_screen(ntedit_hip_ctx* c, const u8* d_seq, u64 n, const Filter& f, u64* d_bitmap, u64 n_words)
{
	const u64 blocks = (n + SCREEN_TILE - 1) / SCREEN_TILE;
	if (blocks == 0) {
		return 0;
	}
	if (!INSERT) {
		u32 slog = 0, n_slices = 0;
		if (binned_applicable(c, f, n, &slog, &n_slices)) {
			return run_screen_binned(c, d_seq, n, f, d_bitmap, n_words, slog, n_slices);
		}
	}
	if (!INSERT) {
		c->bin_chunks_last = 0;
	}
	return launch_screen_tiles<INSERT>(c, c->stream, d_seq, n, f, d_bitmap, n_words, 0, blocks, 0);
}

// ---- L2-partitioned ("binned") screening; see nte_kernels.hip / nte_bin_wc.inc
// screen_mode 1 forces the direct gather kernel, 2 the binned pipeline (tests run it on small inputs); 0 picks:
// the binned pipeline pays when the filter is far larger than the L2s (the direct kernel then runs at the
// fabric's ~51 G requests/s) and the batch is large enough to fill the persistent partition kernel.
bool
binned_applicable(const ntedit_hip_ctx* c, const Filter& f, u64 n, u32* slice_log2, u32* n_slices)
{
	u32 mode = c->tune.screen_mode ? c->tune.screen_mode : c->hp.screen_mode;
	// (-s 1 has no screening probes at all: k_screen only marks the k-mers of accepted bases)
	if (mode == 1 || c->bin_fallback || f.hash_num == 0 || f.hash_num > 5 || c->hp.snv) {
		return false;
	}
	// slices of 2 MiB of filter (2^24 bit slots, 2^21 counters) while it has at most WC_MAX_SLICES of them, 4 / 8 MiB beyond
	u32 slog = f.counting ? 21 : 24;
	u64 ns = (f.bits + (1ULL << slog) - 1) >> slog;
	while (ns > (u64)WC_MAX_SLICES) {
		slog++;
		ns = (f.bits + (1ULL << slog) - 1) >> slog;
	}
	if (slog > (f.counting ? 23u : 26u) || n >= (1ULL << (63 - slog)) || wc_lds_bytes(c->dp.k) + 1024 > c->lds_per_block) {
		return false; // slices beyond 8 MiB do not stay in an XCD's 4 MiB L2 long enough to matter
	}
	*slice_log2 = slog;
	*n_slices = (u32)ns;
	if (mode == 2) {
		return true;
	}
	return f.bits >= (f.counting ? 1ULL << 27 : 1ULL << 30) && n >= (1ULL << 26); // auto: filters >= 128 MiB, batches >= 64 Mbases
}

// The binned screening of this context lost probe records (an overflow list overflowed: a draft made of very few
// distinct k-mers): its bitmap is void.  From then on the context screens with the direct kernel; the caller runs
// the screening again.  (Streams must be idle.)
int
bin_records_lost(ntedit_hip_ctx* c, bool* lost)
{
	*lost = false;
	if (!c->bin_lost.p || !c->bin_chunks_last) {
		return 0;
	}
	u32 v = 0;
	HIP_TRY(c, hipMemcpy(&v, c->bin_lost.p, 4, hipMemcpyDeviceToHost));
	if (v) {
		HIP_TRY(c, hipMemset(c->bin_lost.p, 0, 4));
		c->bin_fallback = true;
		*lost = true;
	}
	return 0;
}

// geometry of one record chunk: runs of `cap` records, one per (slice, partition workgroup) pair
#ifndef NTE_MACHINE_PIECES
#define NTE_MACHINE_PIECES 1
#endif

struct WcPlan
{
	u32 n_wg;
	u32 cap;
	u64 n_wtiles;
	u64 record_bytes;
	bool short_runs; // the runs cannot hold what the pairs expect (WC_MAX_RUN): the chunk has to be smaller
};

WcPlan
plan_wc(const ntedit_hip_ctx* c, u64 kmers, u32 hash_num, u32 n_slices)
{
	WcPlan w;
	w.n_wtiles = (kmers + WC_WTILE - 1) / WC_WTILE;
	const u64 per_wg_tiles = WC_WAVES; // a workgroup below that would idle wavefronts
	u64 n_wg = (w.n_wtiles + per_wg_tiles - 1) / per_wg_tiles;
	if (n_wg > c->cu_count) {
		n_wg = c->cu_count; // persistent workgroups, one per CU (the rings take nearly all of a CU's LDS)
	}
	if (n_wg == 0) {
		n_wg = 1;
	}
	w.n_wg = (u32)n_wg;
	// records a pair can expect: the k-mer starts of its workgroup's wavefront tiles, h probes each, spread evenly
	// over the slices (the hash values of distinct k-mers are uniform); 6 sigma + a group on top.  What exceeds the
	// run (repeated k-mers: their probes all meet the same slices) goes to the overflow list.
	const u64 tiles_per_wg = (w.n_wtiles + (u64)w.n_wg * WC_WAVES - 1) / ((u64)w.n_wg * WC_WAVES) * WC_WAVES;
	const double mean = (double)tiles_per_wg * WC_WTILE * hash_num / (double)n_slices;
	double cap = mean + 6.0 * sqrt(mean) + 64.0;
	if (c->tune.bin_cap_percent) {
		cap = mean * c->tune.bin_cap_percent / 100.0 + 8.0; // tests: force the overflow path
	}
	u64 capi = ((u64)cap + WC_GROUP - 1) / WC_GROUP * WC_GROUP;
	const u64 max_run = c->tune.bin_scatter == 1 ? WC_MAX_RUN : WCB_MAX_RUN;
	w.short_runs = !c->tune.bin_cap_percent && capi > max_run;
	if (capi > max_run) {
		capi = max_run; // (what does not fit goes through the overflow list)
	}
	w.cap = (u32)capi;
	w.record_bytes = (u64)n_slices * w.n_wg * w.cap * 8;
	return w;
}

template<int H, bool POW2>
int
launch_wc(ntedit_hip_ctx* c, hipStream_t stream, const WcArgs& w)
{
	// (per device: the attribute belongs to the function ON the current device)
	if (c->tune.bin_scatter == 1) {
		const size_t lds = wc_lds_bytes(w.b.p.k);
		HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wc_scatter<H, POW2>),
		                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
		hipLaunchKernelGGL((k_wc_scatter<H, POW2>), dim3(w.n_wg), dim3(WC_TPB), lds, stream, w);
	} else {
		HIP_TRY(c, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wc_scatter_b<H, POW2>),
		                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)WCB_LDS_BYTES));
		hipLaunchKernelGGL((k_wc_scatter_b<H, POW2>), dim3(w.n_wg), dim3(WCB_TPB), WCB_LDS_BYTES, stream, w);
	}
	return 0;
}

int
run_wc_partition(ntedit_hip_ctx* c, hipStream_t stream, const WcArgs& w)
{
	const bool pow2 = w.b.f.mask != 0;
	switch (w.b.f.hash_num) {
	case 1:
		return pow2 ? launch_wc<1, true>(c, stream, w) : launch_wc<1, false>(c, stream, w);
	case 2:
		return pow2 ? launch_wc<2, true>(c, stream, w) : launch_wc<2, false>(c, stream, w);
	case 3:
		return pow2 ? launch_wc<3, true>(c, stream, w) : launch_wc<3, false>(c, stream, w);
	case 4:
		return pow2 ? launch_wc<4, true>(c, stream, w) : launch_wc<4, false>(c, stream, w);
	default:
		return pow2 ? launch_wc<5, true>(c, stream, w) : launch_wc<5, false>(c, stream, w);
	}
}

int
run_screen_binned(ntedit_hip_ctx* c, const u8* d_seq, u64 n, const Filter& f, u64* d_bitmap, u64 n_words, u32 slog, u32 n_slices, hipStream_t stream, u64 pos_begin, u64 pos_end)
{
	if (!stream) {
		stream = c->stream;
	}
	if (pos_end > n) {
		pos_end = n;
	}
	const u64 span = pos_end - pos_begin;
	// Record chunks: the whole range at once when its records take no more than 40 % of the HBM that is free right
	// now (3 Gbp at h = 3: 74.5 GB), else as few equal chunks as that allows.  (Partitioning chunk j + 1 on this stream
	// while chunk j is probed on another -- "bin_overlap", two record buffers, the partition kernel held to 96 VGPRs so
	// that probe wavefronts fit next to it -- was built and measured on the 3 Gbp workload: 113.7 ms in four chunks
	// against 109-115 ms for chunks one after the other and 103 ms for ONE chunk; the sum of the stages' own times.  Both
	// stages lean on the L2: the probe stage saturates it, the partition stage sends 4.5e9 16-byte stores through it.)
	const u64 unit = (u64)WC_WTILE * WC_WAVES; // chunk sizes: whole workgroup rounds (also a multiple of 16 bytes)
	const bool overlap = c->tune.bin_overlap && span >= (1ULL << 28);
	u64 parts = overlap ? 4 : 1;
	const bool arriving = c->h2d_piece_bytes && span >= 4 * c->h2d_piece_bytes && !overlap;
	{
		size_t free_b = 0, total_b = 0;
		u64 room = ~0ULL;
		if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
			room = (u64)(free_b + c->bin_records[0].cap + c->bin_records[1].cap) / 5 * (overlap ? 1 : 2);
		}
		for (;;) {
			const WcPlan pl = plan_wc(c, (span + parts - 1) / parts, f.hash_num, n_slices);
			if (parts >= 4096 || (pl.record_bytes <= room && !pl.short_runs)) {
				break;
			}
			parts++;
		}
		if (plan_wc(c, (span + parts - 1) / parts, f.hash_num, n_slices).record_bytes > room) {
			return fail(c, NTEDIT_E_DEVICE, "not enough device memory for the screening records");
		}
	}
	u64 chunk = ((span + parts - 1) / parts + unit - 1) / unit * unit;
	if (c->tune.bin_chunk) { // tests: force several chunks
		const u64 v = c->tune.bin_chunk / unit * unit;
		if (v >= unit && v < chunk) {
			chunk = v;
		}
	}
	const bool two = overlap && chunk < span; // two buffer sets, two streams
	const WcPlan plan0 = plan_wc(c, span < chunk ? span : chunk, f.hash_num, n_slices);
	const u32 ovf_cap = 1u << 22; // 64 MiB of overflow entries per chunk; beyond that the direct kernel takes over
	const u32 parts_log2 = c->tune.probe_parts_log2;
	const size_t ctl_words = CTL_WORK + ((size_t)n_slices << parts_log2) + 1;
	int rc;
	if (!c->bin_lost.p) {
		if ((rc = ensure(c, c->bin_lost, 4))) {
			return rc;
		}
		HIP_TRY(c, hipMemsetAsync(c->bin_lost.p, 0, 4, stream));
	}
	for (int q = 0; q < (two ? 2 : 1); q++) {
		if ((rc = ensure(c, c->bin_records[q], plan0.record_bytes)) ||
		    (rc = ensure(c, c->bin_fill[q], (size_t)n_slices * plan0.n_wg * 4)) ||
		    (rc = ensure(c, c->bin_ctl[q], (ctl_words + 8) * 4)) ||
		    (rc = ensure(c, c->bin_ovf[q], (size_t)ovf_cap * sizeof(WcOvf)))) {
			return rc;
		}
	}
	if (two && !c->stream3) {
		HIP_TRY(c, hipStreamCreate(&c->stream3));
	}
	hipStream_t pstream = two ? c->stream3 : stream;
	{
		// the probe stage ORs into the bitmap: clear the words of the range first
		const u64 w0 = pos_begin / 64, w1 = (pos_end + 63) / 64 < n_words ? (pos_end + 63) / 64 : n_words;
		HIP_TRY(c, hipMemsetAsync(d_bitmap + w0, 0, (w1 - w0) * 8, stream));
	}
	// chunk boundaries.  A batch that is still crossing PCIe is screened in chunks that grow -- one copy piece, three,
	// eight, the rest: every chunk waits for its own pieces only, the first one for 128 MB instead of the whole batch,
	// and the copy (55 GB/s) stays ahead of the screening (32 GB/s) from there on.
	std::vector<u64> cuts;
	{
		u64 at = pos_begin, step = chunk;
		if (arriving) {
			step = c->h2d_piece_bytes / unit * unit;
			step = step < unit ? unit : step;
		}
		while (at < pos_end) {
			u64 len = step < chunk ? step : chunk;
			if (arriving && pos_end - at < len + len / 2) {
				len = pos_end - at < chunk ? pos_end - at : chunk; // (no small rest)
			}
			at = at + len < pos_end ? at + len : pos_end;
			cuts.push_back(at);
			if (arriving) {
				step = step * 3 < chunk ? step * 3 : chunk;
			}
		}
	}
	const u64 n_chunks = cuts.size();
	// per chunk: [0] partition begins, [1] partitioned, [2] probe begins, [3] probed (timed; [1] and [3] also order the streams)
	while (c->bin_ev.size() < 4 * (size_t)n_chunks) {
		hipEvent_t e;
		HIP_TRY(c, hipEventCreate(&e));
		c->bin_ev.push_back(e);
	}
	u32 chunk_no = 0;
	for (u64 begin = pos_begin; chunk_no < n_chunks; begin = cuts[chunk_no], chunk_no++) {
		const u64 end = cuts[chunk_no];
		const WcPlan plan = plan_wc(c, end - begin, f.hash_num, n_slices);
		const int q = two ? (int)(chunk_no & 1) : 0;
		u32* d_ctl = (u32*)c->bin_ctl[q].p;
		u32* d_ovf_count = d_ctl + ctl_words; // (+ words of NTE_WC_STATS counters)
		WcArgs w;
		w.b.seq = d_seq;
		w.b.n = n;
		w.b.chunk_begin = begin;
		w.b.chunk_end = end;
		w.b.f = f;
		w.b.p = c->dp;
		w.b.tabs = c->d_tab;
		w.b.n_slices = n_slices;
		w.b.slice_log2 = slog;
		w.b.records = (u64*)c->bin_records[q].p;
		w.fill = (u32*)c->bin_fill[q].p;
		w.ovf = (WcOvf*)c->bin_ovf[q].p;
		w.ovf_count = d_ovf_count;
		w.ovf_cap = ovf_cap;
		w.n_wg = plan.n_wg;
		w.cap = plan.cap;
		w.n_wtiles = plan.n_wtiles;
		w.wcodes = (u32)wc_codes_bytes(c->dp.k);
		hipEvent_t* tev = &c->bin_ev[4 * (size_t)chunk_no];
		if (c->h2d_piece_bytes) {
			// the chunk's bases (+ the k-1 behind its end) must have arrived
			u64 piece = (end + SCREEN_TILE) / c->h2d_piece_bytes;
			if (piece >= c->h2d_pieces) {
				piece = c->h2d_pieces - 1;
			}
			HIP_TRY(c, hipStreamWaitEvent(stream, c->h2d_ev[piece], 0));
		}
		if (two && chunk_no >= 2) {
			// this buffer set was last read by the probe of chunk_no - 2
			HIP_TRY(c, hipStreamWaitEvent(stream, c->bin_ev[4 * (size_t)(chunk_no - 2) + 3], 0));
		}
		HIP_TRY(c, hipMemsetAsync(d_ctl, 0, (ctl_words + 8) * 4, stream));
		HIP_TRY(c, hipEventRecord(tev[0], stream));
		if ((rc = run_wc_partition(c, stream, w))) {
			return rc;
		}
		HIP_TRY(c, hipEventRecord(tev[1], stream));
		if (two) {
			HIP_TRY(c, hipStreamWaitEvent(pstream, tev[1], 0));
		}
		HIP_TRY(c, hipEventRecord(tev[2], pstream));
		ProbeArgs pa;
		pa.filter = f.data;
		pa.records = (const u64*)c->bin_records[q].p;
		pa.fill = (const u32*)c->bin_fill[q].p;
		pa.n_slices = n_slices << parts_log2;
		pa.slog = slog;
		pa.parts_log2 = parts_log2;
		pa.n_wg = plan.n_wg;
		pa.cap = plan.cap;
		pa.ctl = d_ctl;
		pa.absent32 = (u32*)d_bitmap;
		pa.force_xcc = c->tune.force_xcc ? c->tune.force_xcc - 1 : PROBE_XCC_ANY;
		pa.counting = f.counting;
		pa.count_lo = c->dp.min_thr > 1 ? c->dp.min_thr : 1;
		hipLaunchKernelGGL(k_bin_probe, dim3(c->cu_count * (2048 / PROBE_TPB)), dim3(PROBE_TPB), 0, pstream, pa);
		hipLaunchKernelGGL(k_ovf_probe, dim3(64), dim3(256), 0, pstream, f.data, (const WcOvf*)c->bin_ovf[q].p, (const u32*)d_ovf_count, ovf_cap, slog,
		                   (u32*)d_bitmap, (u32)f.counting, pa.count_lo);
		// overflow entries that did not fit are lost probes: the caller must look at this before it trusts the bitmap
		hipLaunchKernelGGL(k_ovf_check, dim3(1), dim3(1), 0, pstream, (const u32*)d_ovf_count, ovf_cap, (u32*)c->bin_lost.p);
		HIP_TRY(c, hipGetLastError());
		HIP_TRY(c, hipEventRecord(tev[3], pstream));
		if (c->tune.bin_timing) {
			HIP_TRY(c, hipStreamSynchronize(pstream));
			float t_part = 0.f, t_probe = 0.f;
			(void)hipEventElapsedTime(&t_part, tev[0], tev[1]);
			(void)hipEventElapsedTime(&t_probe, tev[2], tev[3]);
			u32 ovf_n = 0;
			(void)hipMemcpy(&ovf_n, d_ovf_count, 4, hipMemcpyDeviceToHost);
#ifdef NTE_WC_STATS
			u32 stats[4] = { 0, 0, 0, 0 };
			(void)hipMemcpy(stats, d_ovf_count + 1, 16, hipMemcpyDeviceToHost);
			fprintf(stderr, "[ntedit_hip] scatter: %u wavefront rounds, %u extra passes, %u lanes with a record that found its ring half taken, %u deferred group flushes that had to wait again\n",
			        stats[1], stats[0], stats[2], stats[3]);
#endif
			fprintf(stderr, "[ntedit_hip] binned chunk %llu k-mers, %u slices of 2^%u bits, %u x %u-record runs per slice (%.2f GB), %u overflow records: partition %.3f ms, probe %.3f ms (stages timed alone)\n",
			        (unsigned long long)(end - begin), n_slices, slog, plan.n_wg, plan.cap, plan.record_bytes / 1e9, ovf_n, t_part, t_probe);
		}
	}
	if (two) {
		// everything the caller queues on `stream` behind this call sees the complete bitmap
		for (u32 j = chunk_no >= 2 ? chunk_no - 2 : 0; j < chunk_no; j++) {
			HIP_TRY(c, hipStreamWaitEvent(stream, c->bin_ev[4 * (size_t)j + 3], 0));
		}
	}
	c->bin_chunks_last = chunk_no;
	return 0;
}

// bytes [o, o + len) of a batch that arrives in the packed form (o a multiple of 128, or the whole batch): its codes and
// case bits cross PCIe, k_unpack turns them into bytes of c->seq -- all on `stream`
int
copy_packed_piece(ntedit_hip_ctx* c, const char* packed, u64 n, u64 o, u64 len, hipStream_t stream)
{
	const u64 codes_bytes = (n + 31) / 32 * 16;
	const u64 end = o + len;
	const u64 c0 = o / 2, c1 = end == n ? codes_bytes : end / 2;
	const u64 s0 = o / 8, s1 = end == n ? (n + 127) / 128 * 16 : end / 8;
	HIP_TRY(c, hipMemcpyAsync((char*)c->packed.p + c0, packed + c0, c1 - c0, hipMemcpyHostToDevice, stream));
	HIP_TRY(c, hipMemcpyAsync((char*)c->packed.p + codes_bytes + s0, packed + codes_bytes + s0, s1 - s0, hipMemcpyHostToDevice, stream));
	const u64 g0 = o / 16, g1 = (end + 15) / 16; // (c->seq has 64 bytes of slack behind the batch)
	const u64 blocks = (g1 - g0 + 255) / 256;
	hipLaunchKernelGGL(k_unpack, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, (const u8*)c->packed.p,
	                   (const u8*)c->packed.p + codes_bytes, (u8*)c->seq.p, g0, g1 - g0);
	HIP_TRY(c, hipGetLastError());
	return 0;
}

// copies (or adopts) the batch into HBM; returns the device pointer
int
stage_bases(ntedit_hip_ctx* c, const char* bases, u64 n, int on_device, const u8** out, bool copy = true)
{
	if (on_device == NTEDIT_HIP_BASES_DEVICE) {
		if ((uintptr_t)bases & 15) {
			return fail(c, NTEDIT_E_ARG, "device `bases` must be 16-byte aligned");
		}
		*out = (const u8*)bases;
		return 0;
	}
	int rc = ensure(c, c->seq, n + 64);
	if (rc) {
		return rc;
	}
	if (on_device == NTEDIT_HIP_BASES_PACKED) {
		// the packed form: codes + case bits to HBM, k_unpack writes the byte batch (in pieces: copy_packed_piece)
		if ((rc = ensure(c, c->packed, (size_t)((n + 31) / 32 * 16 + (n + 127) / 128 * 16)))) {
			return rc;
		}
		if (copy && (rc = copy_packed_piece(c, bases, n, 0, n, c->stream))) {
			return rc;
		}
	} else if (copy) {
		HIP_TRY(c, hipMemcpyAsync(c->seq.p, bases, n, hipMemcpyHostToDevice, c->stream));
	}
	*out = (const u8*)c->seq.p;
	return 0;
}

} // namespace

extern "C" {

void
ntedit_hip_params_default(ntedit_hip_params* p)
{
	nte_host::params_default(p);
}

void
ntedit_hip_params_clamp(ntedit_hip_params* p, char* warn, size_t cap)
{
	nte_host::params_clamp(p, warn, cap);
}

int
ntedit_hip_create(int device, ntedit_hip_ctx** out)
{
	if (!out) {
		return NTEDIT_E_ARG;
	}
	*out = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
		return NTEDIT_E_DEVICE;
	}
	if (device < 0 || device >= count) {
		return NTEDIT_E_ARG;
	}
	if (hipSetDevice(device) != hipSuccess) {
		return NTEDIT_E_DEVICE;
	}
	ntedit_hip_ctx* c = new ntedit_hip_ctx();
	c->device = device;
	nte_host::params_default(&c->hp);
	hipDeviceProp_t prop;
	if (hipGetDeviceProperties(&prop, device) == hipSuccess) {
		c->cu_count = (u32)prop.multiProcessorCount;
		int lds = 0;
		if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, device) == hipSuccess && lds > 0) {
			c->lds_per_block = (size_t)lds;
		}
	}
	if (hipStreamCreate(&c->stream) != hipSuccess || hipStreamCreate(&c->stream2) != hipSuccess) {
		delete c;
		return NTEDIT_E_DEVICE;
	}
	for (auto& e : c->ev) {
		if (hipEventCreate(&e) != hipSuccess) {
			delete c;
			return NTEDIT_E_DEVICE;
		}
	}
	for (auto& e : c->ev_assess) {
		if (hipEventCreate(&e) != hipSuccess) {
			delete c;
			return NTEDIT_E_DEVICE;
		}
	}
	{
		std::lock_guard<std::mutex> lk(g_live_mu);
		g_live_ctx.push_back(c);
	}
	*out = c;
	return 0;
}

void
ntedit_hip_destroy(ntedit_hip_ctx* c)
{
	if (!c) {
		return;
	}
	{
		std::lock_guard<std::mutex> lk(g_live_mu);
		for (size_t i = 0; i < g_live_ctx.size(); i++) {
			if (g_live_ctx[i] == c) {
				g_live_ctx.erase(g_live_ctx.begin() + (long)i);
				break;
			}
		}
	}
	(void)hipSetDevice(c->device);
	(void)hipStreamSynchronize(c->stream);
	for (auto& f : c->filt) {
		if (f.owned && f.data) {
			(void)hipFree(f.data);
		}
	}
	DevBuf* bufs[] = { &c->seq,      &c->bitmap,   &c->block_counts, &c->block_offsets, &c->events,
		               &c->first_chunk, &c->arena, &c->counters, &c->deferred,     &c->ws_nodes,      &c->ws_ov_pos,
		               &c->ws_ov_chr, &c->ws_prev, &c->ws_lps, &c->ws_win, &c->runmap, &c->packed, &c->bin_records[0], &c->bin_records[1], &c->bin_fill[0], &c->bin_fill[1], &c->bin_ctl[0], &c->bin_ctl[1], &c->bin_ovf[0], &c->bin_ovf[1], &c->bin_lost, &c->ev_cover, &c->ev_before, &c->ev_flags, &c->ev_list, &c->ev_bmax,       &c->offs,          &c->lens };
	for (DevBuf* b : bufs) {
		release(*b);
	}
	for (auto& pb : c->pin_pool) {
		(void)hipHostFree(pb.p);
	}
	c->pin_pool.clear();
	if (c->d_tab) {
		(void)hipFree(c->d_tab);
	}
	for (auto& e : c->ev) {
		if (e) {
			(void)hipEventDestroy(e);
		}
	}
	for (auto& e : c->ev_assess) {
		if (e) {
			(void)hipEventDestroy(e);
		}
	}
	for (auto& e : c->chunk_ev) {
		(void)hipEventDestroy(e);
	}
	for (auto& e : c->h2d_ev) {
		(void)hipEventDestroy(e);
	}
	for (auto& e : c->bin_ev) {
		(void)hipEventDestroy(e);
	}
	if (c->stream_copy) {
		(void)hipStreamSynchronize(c->stream_copy);
		(void)hipStreamDestroy(c->stream_copy);
	}
	if (c->stream3) {
		(void)hipStreamSynchronize(c->stream3);
		(void)hipStreamDestroy(c->stream3);
	}
	if (c->stream2) {
		(void)hipStreamSynchronize(c->stream2);
		(void)hipStreamDestroy(c->stream2);
	}
	if (c->stream) {
		(void)hipStreamDestroy(c->stream);
	}
	delete c;
}

const char*
ntedit_hip_last_error(const ntedit_hip_ctx* c)
{
	return c ? c->err.c_str() : "no context";
}

static int
drop_filter(ntedit_hip_ctx* c, int slot)
{
	DevFilter& f = c->filt[slot];
	if (f.owned && f.data) {
		HIP_TRY(c, hipFree(f.data));
	}
	f = DevFilter();
	c->dp_valid = false;
	return 0;
}

int
ntedit_hip_set_filter(
    ntedit_hip_ctx* c,
    int slot,
    const uint8_t* bits,
    uint64_t nbytes,
    uint32_t hash_num,
    uint32_t k,
    int counting)
{
	if (!c || slot < 0 || slot > 1 || !bits || nbytes == 0) {
		return fail(c, NTEDIT_E_ARG, "set_filter: bad argument");
	}
	HIP_TRY(c, hipSetDevice(c->device));
	int rc = drop_filter(c, slot);
	if (rc) {
		return rc;
	}
	DevFilter& f = c->filt[slot];
	// the slot arithmetic uses exactly nbytes (btllib takes the header's size as it is); only the
	// allocation is padded to whole 64-bit words (zero-filled: k_popcount reads words)
	const u64 padded = (nbytes + 7) / 8 * 8;
	HIP_TRY(c, hipMalloc((void**)&f.data, padded));
	f.owned = true;
	if (padded != nbytes) {
		HIP_TRY(c, hipMemset(f.data + (padded - 8), 0, 8));
	}
	HIP_TRY(c, hipMemcpy(f.data, bits, nbytes, hipMemcpyHostToDevice));
	f.nbytes = nbytes;
	f.hash_num = hash_num;
	f.k = k;
	f.counting = counting != 0;
	f.set = true;
	return 0;
}

int
ntedit_hip_set_filter_device(
    ntedit_hip_ctx* c,
    int slot,
    void* device_bits,
    uint64_t nbytes,
    uint32_t hash_num,
    uint32_t k,
    int counting)
{
	// (the caller's allocation must reach the next multiple of 8 bytes, zero-filled behind nbytes)
	if (!c || slot < 0 || slot > 1 || !device_bits || nbytes == 0 || ((uintptr_t)device_bits & 7)) {
		return fail(c, NTEDIT_E_ARG, "set_filter_device: bad argument");
	}
	HIP_TRY(c, hipSetDevice(c->device));
	int rc = drop_filter(c, slot);
	if (rc) {
		return rc;
	}
	DevFilter& f = c->filt[slot];
	f.data = (u8*)device_bits;
	f.owned = false;
	f.nbytes = nbytes;
	f.hash_num = hash_num;
	f.k = k;
	f.counting = counting != 0;
	f.set = true;
	return 0;
}

int
ntedit_hip_load_filter_file(ntedit_hip_ctx* c, int slot, const char* path)
{
	if (!c || slot < 0 || slot > 1 || !path) {
		return fail(c, NTEDIT_E_ARG, "load_filter_file: bad argument");
	}
	nte_host::BfHeader h;
	const char* why = nullptr;
	FILE* f = nte_host::bf_open(path, &h, &why);
	if (!f) {
		return fail(c, NTEDIT_E_IO, "`%s': %s", path, why ? why : "not a readable btllib Bloom filter file");
	}
	if (h.k < 12 || h.k > 200 || h.hash_num > MAX_HASHES) {
		fclose(f);
		return fail(c, NTEDIT_E_ARG, "`%s': k = %u, hash_num = %u: this build supports k in [12, 200] and at most %u hash functions",
		            path, h.k, h.hash_num, MAX_HASHES);
	}
	HIP_TRY(c, hipSetDevice(c->device));
	int rc = drop_filter(c, slot);
	if (rc) {
		fclose(f);
		return rc;
	}
	DevFilter& d = c->filt[slot];
	const u64 nbytes = (h.bytes + 7) / 8 * 8;
	if (hipMalloc((void**)&d.data, nbytes) != hipSuccess) {
		fclose(f);
		return fail(c, NTEDIT_E_DEVICE, "hipMalloc(%llu) failed", (unsigned long long)nbytes);
	}
	d.owned = true;
	(void)hipMemset(d.data, 0, nbytes);
	// stream the array through a pinned bounce buffer
	const size_t CH = 64u << 20;
	void* bounce = nullptr;
	if (hipHostMalloc(&bounce, CH, hipHostMallocDefault) != hipSuccess) {
		fclose(f);
		(void)drop_filter(c, slot);
		return fail(c, NTEDIT_E_DEVICE, "hipHostMalloc failed");
	}
	u64 done = 0;
	bool ok = true;
	while (done < h.bytes) {
		size_t want = (size_t)((h.bytes - done) < CH ? (h.bytes - done) : CH);
		size_t got = fread(bounce, 1, want, f);
		if (got != want) {
			ok = false;
			break;
		}
		if (hipMemcpy(d.data + done, bounce, got, hipMemcpyHostToDevice) != hipSuccess) {
			ok = false;
			break;
		}
		done += got;
	}
	(void)hipHostFree(bounce);
	fclose(f);
	if (!ok) {
		drop_filter(c, slot);
		return fail(c, NTEDIT_E_IO, "`%s': truncated Bloom filter file", path);
	}
	d.nbytes = h.bytes; // the size in the header IS the modulus (the allocation is padded to 64-bit words)
	d.hash_num = h.hash_num;
	d.k = h.k;
	d.counting = h.counting;
	d.set = true;
	return 0;
}

int
ntedit_hip_filter_info(
    const ntedit_hip_ctx* c,
    int slot,
    uint32_t* k,
    uint32_t* hash_num,
    uint64_t* nbytes,
    int* counting)
{
	if (!c || slot < 0 || slot > 1 || !c->filt[slot].set) {
		return NTEDIT_E_NOFILTER;
	}
	const DevFilter& f = c->filt[slot];
	if (k) {
		*k = f.k;
	}
	if (hash_num) {
		*hash_num = f.hash_num;
	}
	if (nbytes) {
		*nbytes = f.nbytes;
	}
	if (counting) {
		*counting = f.counting;
	}
	return 0;
}

void*
ntedit_hip_filter_device_ptr(const ntedit_hip_ctx* c, int slot)
{
	if (!c || slot < 0 || slot > 1 || !c->filt[slot].set) {
		return nullptr;
	}
	return c->filt[slot].data;
}

int
ntedit_hip_filter_alloc(ntedit_hip_ctx* c, int slot, uint64_t nbytes, uint32_t hash_num, uint32_t k)
{
	if (!c || slot < 0 || slot > 1 || nbytes == 0 || hash_num == 0 || hash_num > MAX_HASHES) {
		return fail(c, NTEDIT_E_ARG, "filter_alloc: bad argument");
	}
	HIP_TRY(c, hipSetDevice(c->device));
	int rc = drop_filter(c, slot);
	if (rc) {
		return rc;
	}
	nbytes = (nbytes + 7) / 8 * 8;
	DevFilter& f = c->filt[slot];
	HIP_TRY(c, hipMalloc((void**)&f.data, nbytes));
	f.owned = true;
	HIP_TRY(c, hipMemset(f.data, 0, nbytes));
	f.nbytes = nbytes;
	f.hash_num = hash_num;
	f.k = k;
	f.counting = false;
	f.set = true;
	return 0;
}

int
ntedit_hip_filter_insert(ntedit_hip_ctx* c, int slot, const char* bases, uint64_t n, int on_device)
{
	if (!c || slot < 0 || slot > 1 || !c->filt[slot].set || !bases) {
		return fail(c, NTEDIT_E_ARG, "filter_insert: bad argument");
	}
	HIP_TRY(c, hipSetDevice(c->device));
	const DevFilter& df = c->filt[slot];
	if (df.counting) {
		return fail(c, NTEDIT_E_UNSUPPORTED, "filter_insert: counting filters are built on the host (ntStat)");
	}
	// insertion only needs k, the multipliers and the seed tables
	ntedit_hip_params hp;
	nte_host::params_default(&hp);
	DevParams saved = c->dp;
	bool saved_valid = c->dp_valid;
	int rc = nte_host::make_dev_params(hp, df.k, df.hash_num, false, &c->dp);
	if (rc) {
		c->dp = saved;
		return fail(c, rc, "filter_insert: unsupported k/hash_num");
	}
	if (!c->d_tab) {
		HIP_TRY(c, hipMalloc((void**)&c->d_tab, TAB_WORDS * sizeof(u64)));
	}
	if (c->tab_k != df.k) {
		u64 tab[TAB_WORDS];
		build_seed_tables(df.k, tab);
		HIP_TRY(c, hipMemcpy(c->d_tab, tab, sizeof tab, hipMemcpyHostToDevice));
		c->tab_k = df.k;
	}
	if (on_device != NTEDIT_HIP_BASES_HOST && on_device != NTEDIT_HIP_BASES_DEVICE) {
		return fail(c, NTEDIT_E_ARG, "filter_insert: bases must be host or device bytes");
	}
	const u8* d_seq = nullptr;
	rc = stage_bases(c, bases, n, on_device, &d_seq);
	if (rc == 0) {
		HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
		rc = launch_screen<true>(c, d_seq, n, dev_filter(df), nullptr, 0);
		HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		HIP_TRY(c, hipEventElapsedTime(&c->last_ms, c->ev[0], c->ev[1]));
	}
	c->dp = saved;
	c->dp_valid = saved_valid;
	return rc;
}

int
ntedit_hip_filter_occupancy(ntedit_hip_ctx* c, int slot, uint64_t* occupied, uint64_t* slots)
{
	if (!c || slot < 0 || slot > 1 || !c->filt[slot].set || !occupied) {
		return fail(c, NTEDIT_E_ARG, "filter_occupancy: bad argument");
	}
	HIP_TRY(c, hipSetDevice(c->device));
	const DevFilter& f = c->filt[slot];
	int rc = ensure(c, c->counters, 256);
	if (rc) {
		return rc;
	}
	unsigned long long* d_total = (unsigned long long*)c->counters.p;
	HIP_TRY(c, hipMemsetAsync(d_total, 0, 8, c->stream));
	const u64 n_words = (f.nbytes + 7) / 8; // (allocations are whole 64-bit words, zero behind nbytes)
	hipLaunchKernelGGL(k_popcount, dim3((unsigned)(c->cu_count * 8)), dim3(256), 0, c->stream, (const u64*)f.data, n_words,
	                   f.counting ? 1 : 0, d_total);
	HIP_TRY(c, hipGetLastError());
	unsigned long long h = 0;
	HIP_TRY(c, hipMemcpyAsync(&h, d_total, 8, hipMemcpyDeviceToHost, c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	*occupied = h;
	if (slots) {
		*slots = f.counting ? f.nbytes : f.nbytes * 8;
	}
	return 0;
}

int
ntedit_hip_filter_download(const ntedit_hip_ctx* c, int slot, uint8_t* bits)
{
	if (!c || slot < 0 || slot > 1 || !c->filt[slot].set || !bits) {
		return NTEDIT_E_ARG;
	}
	if (hipSetDevice(c->device) != hipSuccess) {
		return NTEDIT_E_DEVICE;
	}
	const DevFilter& f = c->filt[slot];
	return hipMemcpy(bits, f.data, f.nbytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : NTEDIT_E_DEVICE;
}

int
ntedit_hip_filter_save_file(const ntedit_hip_ctx* c, int slot, const char* path)
{
	if (!c || slot < 0 || slot > 1 || !c->filt[slot].set || !path) {
		return NTEDIT_E_ARG;
	}
	const DevFilter& f = c->filt[slot];
	std::vector<u8> host(f.nbytes);
	int rc = ntedit_hip_filter_download(c, slot, host.data());
	if (rc) {
		return rc;
	}
	nte_host::BfHeader h;
	h.bytes = f.nbytes;
	h.hash_num = f.hash_num;
	h.k = f.k;
	h.counting = f.counting;
	return nte_host::bf_save(path, h, host.data()) ? NTEDIT_E_IO : 0;
}

int
ntedit_hip_set_params(ntedit_hip_ctx* c, const ntedit_hip_params* p)
{
	if (!c || !p) {
		return NTEDIT_E_ARG;
	}
	c->hp = *p;
	c->dp_valid = false;
	return 0;
}

int
ntedit_hip_screen(ntedit_hip_ctx* c, const char* bases, uint64_t n, int on_device, uint64_t* bitmap)
{
	if (!c || !bases || !bitmap) {
		return fail(c, NTEDIT_E_ARG, "screen: bad argument");
	}
	HIP_TRY(c, hipSetDevice(c->device));
	int rc = refresh_params(c);
	if (rc) {
		return rc;
	}
	if (on_device != NTEDIT_HIP_BASES_HOST && on_device != NTEDIT_HIP_BASES_DEVICE) {
		return fail(c, NTEDIT_E_ARG, "screen: bases must be host or device bytes");
	}
	const u64 n_words = (n + 63) / 64;
	const u8* d_seq = nullptr;
	rc = stage_bases(c, bases, n, on_device, &d_seq);
	if (rc) {
		return rc;
	}
	u64* d_bitmap = bitmap;
	if (!on_device) {
		rc = ensure(c, c->bitmap, (n_words + 1) * 8);
		if (rc) {
			return rc;
		}
		d_bitmap = (u64*)c->bitmap.p;
	}
	for (;;) {
		HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
		rc = launch_screen<false>(c, d_seq, n, dev_filter(c->filt[0]), d_bitmap, n_words);
		if (rc) {
			return rc;
		}
		HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
		HIP_TRY(c, hipStreamSynchronize(c->stream));
		bool lost = false;
		if ((rc = bin_records_lost(c, &lost))) {
			return rc;
		}
		if (!lost) {
			break;
		}
	}
	if (!on_device) {
		HIP_TRY(c, hipMemcpy(bitmap, d_bitmap, n_words * 8, hipMemcpyDeviceToHost));
	}
	HIP_TRY(c, hipEventElapsedTime(&c->last_ms, c->ev[0], c->ev[1]));
	return 0;
}

} // extern "C"

// ---- ntedit_hip_polish_batch, in stages --------------------------------------------------------------------
// plan()              chunk plan (whole contigs per pipeline chunk), staging of the batch, grow-only buffers
// launch_screening()  stream A: step 1 for every k-mer of the batch (direct or binned; H2D pieces underneath)
// run_chunk_events()  stream B, per chunk: absent bitmap -> ordered event list -> the event machine in rounds
// collect()           edit records to page-locked host memory; events parked by the budget are resolved in serial
//                     order and re-run (host/resolve.h)
// finish()            timings
namespace {

constexpr unsigned NTE_RESOLVE_ROUNDS = 6; // rounds of one-parked-event-per-contig re-runs before the re-runs are widened (collect())

struct PolishRun
{
	// the call
	ntedit_hip_ctx* c;
	const char* bases;
	u64 n;
	const uint64_t* offsets;
	const uint32_t* lens;
	u32 n_contigs;
	int on_device;
	ntedit_hip_result* r;

	// plan
	struct Chunk
	{
		u32 c0, c1; // contigs [c0, c1)
		u64 b0, b1; // byte range of those contigs (incl. their separators)
		u64 t0, t1; // screening tiles
	};
	std::vector<Chunk> chunks;
	size_t n_ch = 0;
	bool pipelined = false;
	u64 n_words = 0;
	const u8* d_seq = nullptr;
	u64* d_bitmap = nullptr;
	u64* d_runmap = nullptr;  // = d_bitmap unless k_assess runs
	bool use_assess = false;
	float ms_assess = 0.f;
	hipStream_t sA = nullptr; // screening
	hipStream_t sB = nullptr; // event extraction + event machine (and H2D pieces while the screening runs)
	u32 grid = 0;
	u64 grid_lo = 0;
	Filter f0;
	size_t screen_pad = 0;
	u64 h2d_piece = 0;
	bool h2d_overlap = false;
	u64 arena_chunks = 0;

	// one attempt (the batch is run again with more room when the arena or a rope window overflows)
	PinBuf early; // pass-1 edit records copied to the host while the sweeps run
	u64 early_chunks = 0;
	u32 h2d_launches = 0;
	u32 status = 0;
	u64 ev_total = 0, absent_total = 0, deferred_total = 0, skipped_total = 0;
	float ms_machine = 0.f;
	bool first_b = true;
	MachineArgs keep_a; // the last chunk's launch arguments (re-runs of parked events)
	// counters layout (bytes): [0] absent k-mers u64, [8] starts of the current chunk u64, [32] arena cursor u32,
	// [40] status u32, [44] deferred count u32, [52] parked events u32, [60] work counter u32, [64] round list length u32
	unsigned long long* d_counters = nullptr;
	u32 *d_arena_next = nullptr, *d_status = nullptr, *d_ndef = nullptr, *d_list_n = nullptr;

	PolishRun()
	{
		early.p = nullptr;
		early.cap = 0;
		memset(&keep_a, 0, sizeof keep_a);
	}

	// a failure from anywhere: nothing may leak (an early copy may still be in flight), *out stays null
	int bail(int code)
	{
		(void)hipDeviceSynchronize();
		pin_give(c, early);
		pin_give(c, r->arena_buf);
		pin_give(c, r->first_buf);
		delete r;
		r = nullptr;
		return code;
	}

	int plan();
	int begin_attempt();
	int launch_screening(int attempt);
	int launch_assess(u64 pos_begin, u64 pos_end);
	void launch_wave_pass(MachineArgs a, const u32* list, u32 count, hipStream_t stream = nullptr, u32* counter = nullptr, u32 blocks_per_cu = 8);
	bool wave_pass_in_lds(const MachineArgs& a) const;
	int extract_events(size_t j, u64* n_ev_out, u64** d_events_out, u32** d_first_out);
	int machine_setup(u64 n_ev, u64* d_events, u32* d_first, MachineArgs* out, u64* blocks_out, size_t* dyn_lds_out);
	int run_chunk_events(size_t j);
	int collect(bool* redo);
	int finish();
};

int
PolishRun::plan()
{
	int rc;
	n_words = (n + 63) / 64;
	// A batch that arrives in host memory crosses PCIe in pieces while the pieces that are already in HBM are
	// being screened (SURVEY 8d "kernel region": host buffer in, edit records out).  Page-locked buffers
	// (ntedit_hip_host_alloc, or any hipHostMalloc / registered memory) copy asynchronously at link speed;
	// pageable ones are staged by the runtime, the overlap is the same.
	h2d_piece = 128ull << 20;
	if (c->tune.h2d_piece != ~0ULL) { // tests / tuning: bytes per piece (0 = one copy up front)
		h2d_piece = c->tune.h2d_piece / SCREEN_TILE * SCREEN_TILE;
	}
	h2d_overlap = on_device != NTEDIT_HIP_BASES_DEVICE && h2d_piece > 0 && n > 2 * h2d_piece;
	if ((rc = stage_bases(c, bases, n, on_device, &d_seq, !h2d_overlap))) {
		return rc;
	}
	if ((rc = ensure(c, c->bitmap, (n_words + 1) * 8)) || (rc = ensure(c, c->counters, 256)) ||
	    (rc = ensure(c, c->offs, (size_t)n_contigs * 8)) || (rc = ensure(c, c->lens, (size_t)n_contigs * 4))) {
		return rc;
	}
	d_bitmap = (u64*)c->bitmap.p;
	d_runmap = d_bitmap;
	sA = c->stream;
	sB = c->stream2;
	HIP_TRY(c, hipMemcpyAsync(c->offs.p, offsets, (size_t)n_contigs * 8, hipMemcpyHostToDevice, sA));
	HIP_TRY(c, hipMemcpyAsync(c->lens.p, lens, (size_t)n_contigs * 4, hipMemcpyHostToDevice, sA));

	// ---- chunk plan: whole contigs, cut at SCREEN_TILE boundaries of the screening pass.
	// Chunk j's screening covers tiles [t0, t1) with t1 = ceil(end of its last contig / TILE),
	// so everything its events can touch has been screened when its screening launch ends.
	{
		const u64 total_tiles = (n + SCREEN_TILE - 1) / SCREEN_TILE;
		// Measured (3 Gbp, MI355X): overlapping the event machine of chunk j with the screening
		// of chunk j+1 does not pay -- both are bound by the L2-miss path, the machine kernels
		// just get slower (2 chunks 339 ms, 8 chunks 370 ms vs 331 ms for one) -- so the
		// default is a single chunk; the chunk pipeline stays available (and tested) for
		// bounded-memory operation.
		u64 target = n + 1;
		if (c->tune.chunk_bytes) { // tests: force many chunks
			target = c->tune.chunk_bytes;
		} else if (h2d_overlap && c->tune.h2d_chunks > 1 && n >= (1ULL << 30)) {
			// A batch that is still crossing PCIe looked like another matter -- the screening waits for the link most
			// of the time, the event machine of the chunks that have arrived could run underneath the copy of the
			// rest -- and is not: 3 Gbp from page-locked memory, 158 ms in one chunk, 198 / 199 / 206 / 223 ms in
			// 4 / 8 / 12 / 16 (the screening next to the machine and the copy takes 166-174 ms instead of 112).
			// Off unless asked for (`h2d_chunks`).
			target = n / c->tune.h2d_chunks + 1;
		}
		u32 c0 = 0;
		u64 t_prev = 0;
		while (c0 < n_contigs) {
			u32 c1 = c0;
			const u64 b0 = offsets[c0];
			u64 b1 = b0;
			while (c1 < n_contigs && (c1 == c0 || offsets[c1] + lens[c1] + 1 - b0 <= target)) {
				b1 = c1 + 1 < n_contigs ? offsets[c1 + 1] : n;
				c1++;
			}
			Chunk ch;
			ch.c0 = c0;
			ch.c1 = c1;
			ch.b0 = c0 == 0 ? 0 : b0;
			ch.b1 = b1;
			ch.t0 = t_prev;
			ch.t1 = c1 == n_contigs ? total_tiles : (b1 + SCREEN_TILE - 1) / SCREEN_TILE;
			if (ch.t1 < ch.t0) {
				ch.t1 = ch.t0;
			}
			t_prev = ch.t1;
			chunks.push_back(ch);
			c0 = c1;
		}
	}
	n_ch = chunks.size();
	pipelined = n_ch > 1;
	while (c->chunk_ev.size() < 2 * n_ch) {
		hipEvent_t e;
		HIP_TRY(c, hipEventCreate(&e));
		c->chunk_ev.push_back(e);
	}
	grid = c->dp.start_grid;
	grid_lo = 0;
	if (grid < 64) {
		for (u32 b = 0; b < 64; b += grid) {
			grid_lo |= 1ULL << b;
		}
	}
	f0 = dev_filter(c->filt[0]);
	// the run map (nte_assess.hip): where most absent positions cannot do anything -- every position is "absent" with
	// -s 1, a quarter of them with a counting filter and -p 2 -- they are taken out before the event machine sees them
	use_assess = n_ch == 1 && (c->tune.assess == 1 || (c->tune.assess == ~0u && (c->dp.snv || c->dp.counting)));
	if (use_assess) {
		if ((rc = ensure(c, c->runmap, (n_words + 8) * 8))) {
			return rc;
		}
		d_runmap = (u64*)c->runmap.p;
	}
	// with more than one chunk the screening kernel is held to 2 workgroups per CU (its speed
	// is set by the L2-miss path, not by occupancy) so the machine kernels of the previous
	// chunk get wave slots, registers and LDS on every CU
	screen_pad = pipelined ? 44 * 1024 : 0;
	arena_chunks = n / 160 + 65536;
	if (arena_chunks * CHUNK_ITEMS * sizeof(Item) < c->arena.cap) {
		arena_chunks = c->arena.cap / (CHUNK_ITEMS * sizeof(Item));
	}
	return 0;
}

int
PolishRun::begin_attempt()
{
	int rc;
	pin_give(c, early); // (streams are idle here)
	early_chunks = 0;
	if (arena_chunks > 0xFFFFFFF0ull) {
		return fail(c, NTEDIT_E_OVERFLOW, "edit-record arena exceeds 2^32 chunks");
	}
	if ((rc = ensure(c, c->arena, arena_chunks * CHUNK_ITEMS * sizeof(Item)))) {
		return rc;
	}
	HIP_TRY(c, hipMemsetAsync(c->counters.p, 0, 256, sA));
	HIP_TRY(c, hipStreamSynchronize(sA));
	d_counters = (unsigned long long*)c->counters.p;
	d_arena_next = (u32*)((char*)c->counters.p + 32);
	d_status = (u32*)((char*)c->counters.p + 40);
	d_ndef = (u32*)((char*)c->counters.p + 44);
	d_list_n = (u32*)((char*)c->counters.p + 64);
	h2d_launches = 0;
	status = 0;
	ev_total = absent_total = deferred_total = skipped_total = 0;
	ms_machine = 0.f;
	first_b = true;
	return 0;
}

// stream A: every chunk's screening, back to back
int
PolishRun::launch_screening(int attempt)
{
	int rc;
	HIP_TRY(c, hipEventRecord(c->ev[0], sA));
	u32 slog_unused = 0, nsl_unused = 0;
	const bool binned = binned_applicable(c, f0, n, &slog_unused, &nsl_unused);
	const u64 n_pieces = h2d_piece ? (n + h2d_piece - 1) / h2d_piece : 0;
	if (h2d_overlap && attempt == 0) {
		while (c->h2d_ev.size() < n_pieces) {
			hipEvent_t e;
			HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
			c->h2d_ev.push_back(e);
		}
	}
	hipStream_t s_copy = sB;
	auto copy_piece = [&](u64 j) -> hipError_t {
		const u64 o = j * h2d_piece, len = o + h2d_piece < n ? h2d_piece : n - o;
		if (on_device == NTEDIT_HIP_BASES_PACKED) {
			if (copy_packed_piece(c, bases, n, o, len, s_copy)) {
				return hipErrorUnknown;
			}
			return hipEventRecord(c->h2d_ev[j], s_copy);
		}
		hipError_t e = hipMemcpyAsync((char*)c->seq.p + o, bases + o, len, hipMemcpyHostToDevice, s_copy);
		return e != hipSuccess ? e : hipEventRecord(c->h2d_ev[j], s_copy);
	};
	if (!pipelined && h2d_overlap && attempt == 0 && !binned) {
		// direct kernel: piece j+1 is copied (stream B is idle until the screening is done) while piece j is
		// screened; the tiles of piece j read k-1 bases of piece j+1, so their launch waits for that copy
		HIP_TRY(c, hipEventRecord(c->chunk_ev[0], sA));
		HIP_TRY(c, copy_piece(0));
		const u64 tiles_per_piece = h2d_piece / SCREEN_TILE;
		const u64 total_tiles = (n + SCREEN_TILE - 1) / SCREEN_TILE;
		for (u64 j = 0; j < n_pieces; j++) {
			if (j + 1 < n_pieces) {
				HIP_TRY(c, copy_piece(j + 1));
			}
			HIP_TRY(c, hipStreamWaitEvent(sA, c->h2d_ev[j + 1 < n_pieces ? j + 1 : j], 0));
			const u64 t0 = j * tiles_per_piece;
			const u64 t1 = j + 1 < n_pieces ? t0 + tiles_per_piece : total_tiles;
			if ((rc = launch_screen_tiles<false>(c, sA, d_seq, n, f0, d_bitmap, n_words, t0, t1 - t0, 0))) {
				return rc;
			}
			h2d_launches++;
		}
		if ((rc = launch_assess(0, n))) {
			return rc;
		}
		HIP_TRY(c, hipEventRecord(c->chunk_ev[1], sA));
	} else if (!pipelined) {
		c->h2d_piece_bytes = 0;
		if (h2d_overlap && attempt == 0) {
			// the binned pipeline: all pieces are queued on stream B right away (page-locked memory: truly
			// asynchronous, at link speed) and every record chunk waits for its own bases only; a retry
			// finds the batch in HBM already
			for (u64 j = 0; j < n_pieces; j++) {
				HIP_TRY(c, copy_piece(j));
			}
			c->h2d_piece_bytes = h2d_piece;
			c->h2d_pieces = n_pieces;
		}
		HIP_TRY(c, hipEventRecord(c->chunk_ev[0], sA));
		rc = launch_screen<false>(c, d_seq, n, f0, d_bitmap, n_words);
		c->h2d_piece_bytes = 0;
		if (rc || (rc = launch_assess(0, n))) {
			return rc;
		}
		HIP_TRY(c, hipEventRecord(c->chunk_ev[1], sA));
	} else {
		c->h2d_piece_bytes = 0;
		if (h2d_overlap && attempt == 0) {
			// the pieces cross on a stream of their own (stream B runs the event machine of the chunks that are
			// through), every chunk's screening waits for its own bases
			if (!c->stream_copy) {
				HIP_TRY(c, hipStreamCreate(&c->stream_copy));
			}
			s_copy = c->stream_copy;
			HIP_TRY(c, hipStreamWaitEvent(s_copy, c->ev[0], 0)); // (behind whatever the earlier work on stream A still reads)
			for (u64 j = 0; j < n_pieces; j++) {
				HIP_TRY(c, copy_piece(j));
			}
			c->h2d_piece_bytes = h2d_piece;
			c->h2d_pieces = n_pieces;
		}
		u32 slog = 0, nsl = 0;
		const bool bin_chunks = binned_applicable(c, f0, n, &slog, &nsl);
		if (!bin_chunks) {
			c->bin_chunks_last = 0;
		}
		for (size_t j = 0; j < n_ch; j++) {
			HIP_TRY(c, hipEventRecord(c->chunk_ev[2 * j], sA));
			if (c->h2d_piece_bytes && !bin_chunks) {
				u64 piece = (chunks[j].t1 * SCREEN_TILE + SCREEN_TILE) / c->h2d_piece_bytes;
				piece = piece < n_pieces ? piece : n_pieces - 1;
				HIP_TRY(c, hipStreamWaitEvent(sA, c->h2d_ev[piece], 0));
			}
			if (bin_chunks) {
				// (the probe stage of chunk j+1 leaves room on the CUs for the event machine of chunk j)
				const u64 p0 = chunks[j].t0 * SCREEN_TILE, p1 = chunks[j].t1 * SCREEN_TILE;
				if (p1 > p0 && (rc = run_screen_binned(c, d_seq, n, f0, d_bitmap, n_words, slog, nsl, sA, p0, p1 < n ? p1 : n))) {
					return rc;
				}
			} else if ((rc = launch_screen_tiles<false>(
			                c, sA, d_seq, n, f0, d_bitmap, n_words, chunks[j].t0, chunks[j].t1 - chunks[j].t0, screen_pad))) {
				return rc;
			}
			HIP_TRY(c, hipEventRecord(c->chunk_ev[2 * j + 1], sA));
		}
		c->h2d_piece_bytes = 0;
	}
	HIP_TRY(c, hipEventRecord(c->ev[1], sA));
	return 0;
}

// stream A, behind the screening of [pos_begin, pos_end): the run map of those positions
int
PolishRun::launch_assess(u64 pos_begin, u64 pos_end)
{
	if (!use_assess || pos_end <= pos_begin) {
		return 0;
	}
	AssessArgs a;
	a.seq = d_seq;
	a.n_bytes = n;
	a.bitmap = d_bitmap;
	a.runmap = d_runmap;
	a.tabs = c->d_tab;
	a.p = c->dp;
	a.bloom = f0;
	a.rep = c->filt[1].set ? dev_filter(c->filt[1]) : f0;
	a.pos_begin = pos_begin;
	a.pos_end = pos_end;
	const u64 tile = (u64)assess_tile();
	a.n_tiles = (pos_end - pos_begin + tile - 1) / tile;
	const u64 cap = (u64)c->cu_count * 32;
	HIP_TRY(c, hipEventRecord(c->ev_assess[0], sA));
	launch_k_assess((unsigned)(a.n_tiles < cap ? a.n_tiles : cap), sA, a);
	HIP_TRY(c, hipGetLastError());
	HIP_TRY(c, hipEventRecord(c->ev_assess[1], sA));
	return 0;
}

// the wavefront-per-event kernel over a list of events of the current chunk
// (stream / counter: a launch that runs NEXT TO a thread-per-event launch has a stream and a work counter of its own)
void
PolishRun::launch_wave_pass(MachineArgs a, const u32* list, u32 count, hipStream_t stream, u32* counter, u32 blocks_per_cu)
{
	if (!stream) {
		stream = sB;
	}
	if (counter) {
		a.work_counter = counter;
	}
	a.defer = 0;
	a.ev_list = list;
	a.n_events = count;
	a.win_bytes += 64; // (one shared window for the positions of 64 lanes, run_lanes)
	const u64 per_block = (u64)MACHINE_TPB / (u64)machine_wave_group();
	const u64 want2 = ((u64)count + per_block - 1) / per_block;
	const u64 cap2 = (u64)c->cu_count * blocks_per_cu;
	const u64 b2 = want2 < cap2 ? want2 : cap2;
	// the wave kernel runs few events per block: window and workspace both fit in LDS
	size_t dyn2 = a.win_in_lds ? (size_t)a.win_bytes * per_block : 0;
	const u64 Wn = a.p.node_window;
	const u64 w16 = (Wn + 15) & ~15ull;
	const u64 slab = Wn * 16 + w16 * 4 + w16 * 2 + w16 + w16;
	const u64 win_area = ((u64)a.win_bytes * per_block + 15) & ~15ull;
	if (win_area + slab * per_block <= 40 * 1024 && !c->tune.no_lds_ws) {
		a.win_in_lds = 1;
		a.lds_ws_off = (u32)win_area;
		a.lds_slab = (u32)slab;
		dyn2 = (size_t)(win_area + slab * per_block);
	}
	(void)hipMemsetAsync(a.work_counter, 0, 4, stream);
	launch_k_machine_wave((unsigned)b2, dyn2, stream, a);
}

// the wave kernel keeps window AND workspace of its events in LDS (so it shares no per-worker slab of global
// memory with a thread-per-event launch that runs at the same time)
bool
PolishRun::wave_pass_in_lds(const MachineArgs& a) const
{
	const u64 per_block = (u64)MACHINE_TPB / (u64)machine_wave_group();
	const u64 Wn = a.p.node_window;
	const u64 w16 = (Wn + 15) & ~15ull;
	const u64 slab = Wn * 16 + w16 * 4 + w16 * 2 + w16 + w16;
	const u64 win_area = ((u64)(a.win_bytes + 64) * per_block + 15) & ~15ull;
	return win_area + slab * per_block <= 40 * 1024 && !c->tune.no_lds_ws;
}

// stream B, chunk j, as soon as its screening is done: absent bitmap -> ordered event list (count, single-workgroup
// scan, write).  *n_ev_out = its events, written behind those of the earlier chunks (d_events / d_first).
int
PolishRun::extract_events(size_t j, u64* n_ev_out, u64** d_events_out, u32** d_first_out)
{
	int rc;
	*n_ev_out = 0;
	const Chunk& ch = chunks[j];
	HIP_TRY(c, hipStreamWaitEvent(sB, c->chunk_ev[2 * j + 1], 0));
	if (first_b) {
		HIP_TRY(c, hipEventRecord(c->ev[2], sB));
		first_b = false;
	}
	const u64 w0 = ch.b0 / 64, w1 = (ch.b1 + 63) / 64;
	const u64 n_sblocks = (w1 - w0 + ST_TPB - 1) / ST_TPB;
	if (n_sblocks == 0) {
		return 0;
	}
	if ((rc = ensure(c, c->block_counts, n_sblocks * 4)) || (rc = ensure(c, c->block_offsets, n_sblocks * 8))) {
		return rc;
	}
	HIP_TRY(c, hipMemsetAsync((char*)c->counters.p + 8, 0, 8, sB));
	hipLaunchKernelGGL(
	    k_count_starts, dim3((unsigned)n_sblocks), dim3(ST_TPB), 0, sB, d_runmap, w0, w1, ch.b0, ch.b1, grid_lo, grid,
	    (u32*)c->block_counts.p, d_counters);
	hipLaunchKernelGGL(
	    k_scan_counts, dim3(1), dim3(1024), 0, sB, (const u32*)c->block_counts.p, n_sblocks,
	    (unsigned long long*)c->block_offsets.p, d_counters);
	unsigned long long h_counters[2] = { 0, 0 };
	HIP_TRY(c, hipMemcpyAsync(h_counters, d_counters, 16, hipMemcpyDeviceToHost, sB));
	HIP_TRY(c, hipStreamSynchronize(sB));
	const u64 n_ev = h_counters[1];
	absent_total = h_counters[0];
	if (n_ev == 0) {
		return 0;
	}
	if (ev_total + n_ev > 0xFFFFFFF0ull) {
		return fail(c, NTEDIT_E_OVERFLOW, "more than 2^32 events in one batch");
	}
	// grow-only buffers; (re)allocation happens on the first batches only
	if (ev_total + n_ev > c->events.cap / 8 || ev_total + n_ev > c->first_chunk.cap / 4) {
		// keep what earlier chunks wrote: allocate bigger buffers and copy
		const u64 want = (ev_total + n_ev) * 2 + 1024;
		DevBuf ne, nf;
		if ((rc = ensure(c, ne, want * 8)) || (rc = ensure(c, nf, want * 4))) {
			return rc;
		}
		if (ev_total) {
			HIP_TRY(c, hipMemcpyAsync(ne.p, c->events.p, ev_total * 8, hipMemcpyDeviceToDevice, sB));
			HIP_TRY(c, hipMemcpyAsync(nf.p, c->first_chunk.p, ev_total * 4, hipMemcpyDeviceToDevice, sB));
			HIP_TRY(c, hipStreamSynchronize(sB));
		}
		release(c->events);
		release(c->first_chunk);
		c->events = ne;
		c->first_chunk = nf;
	}
	if ((rc = ensure(c, c->deferred, n_ev * 4))) {
		return rc;
	}
	u64* d_events = (u64*)c->events.p + ev_total;
	u32* d_first = (u32*)c->first_chunk.p + ev_total;
	hipLaunchKernelGGL(
	    k_write_starts, dim3((unsigned)n_sblocks), dim3(ST_TPB), 0, sB, d_runmap, w0, w1, ch.b0, ch.b1, grid_lo, grid,
	    (const unsigned long long*)c->block_offsets.p, d_events);
	*n_ev_out = n_ev;
	*d_events_out = d_events;
	*d_first_out = d_first;
	return 0;
}

// the machine's launch arguments and per-thread workspace for a chunk of n_ev events
int
PolishRun::machine_setup(u64 n_ev, u64* d_events, u32* d_first, MachineArgs* out, u64* blocks_out, size_t* dyn_lds_out)
{
	int rc;
	const u64 max_threads = (u64)c->cu_count * 2048;
	u64 threads = n_ev < max_threads ? n_ev : max_threads;
	const u64 blocks = (threads + MACHINE_TPB - 1) / MACHINE_TPB;
	threads = blocks * MACHINE_TPB;
	const u64 W = c->dp.node_window;
	if ((rc = ensure(c, c->ws_nodes, threads * W * sizeof(Node))) || (rc = ensure(c, c->ws_ov_pos, threads * W * 4)) ||
	    (rc = ensure(c, c->ws_ov_chr, threads * W)) || (rc = ensure(c, c->ws_prev, threads * W)) ||
	    (rc = ensure(c, c->ws_lps, threads * W * 2))) {
		return rc;
	}
	MachineArgs a;
	a.seq = d_seq;
	a.n_bytes = n;
	a.offsets = (const u64*)c->offs.p;
	a.lens = (const u32*)c->lens.p;
	a.n_contigs = n_contigs;
	a.bitmap = d_bitmap;
	a.runmap = d_runmap;
	a.events = d_events;
	a.n_events = n_ev;
	a.tabs = c->d_tab;
	a.p = c->dp;
	a.bloom = f0;
	a.rep = c->filt[1].set ? dev_filter(c->filt[1]) : f0;
	a.ws_nodes = (Node*)c->ws_nodes.p;
	a.ws_ov_pos = (u32*)c->ws_ov_pos.p;
	a.ws_ov_chr = (u8*)c->ws_ov_chr.p;
	a.ws_prev = (u8*)c->ws_prev.p;
	a.ws_lps = (int16_t*)c->ws_lps.p;
	a.win_bytes = 2 * c->dp.k + c->dp.max_deletions + 8 + 32; // Machine::win_bytes() + slack
	a.win_in_lds = (size_t)a.win_bytes * MACHINE_TPB <= 40 * 1024 ? 1 : 0;
	a.ws_win = nullptr;
	if (!a.win_in_lds) {
		if ((rc = ensure(c, c->ws_win, threads * (a.win_bytes + 64)))) { // (+ 64: the wavefront-per-event launch, launch_wave_pass)
			return rc;
		}
		a.ws_win = (u8*)c->ws_win.p;
	}
	const size_t dyn_lds = a.win_in_lds ? (size_t)a.win_bytes * MACHINE_TPB : 0;
	a.arena = (Item*)c->arena.p;
	a.arena_next = d_arena_next;
	a.arena_chunks = (u32)arena_chunks;
	a.first_chunk = d_first;
	a.status = d_status;
	a.lds_ws_off = 0;
	a.lds_slab = 0;
	a.defer = 1;
	a.ev_list = nullptr;
	a.deferred = (u32*)c->deferred.p;
	a.n_deferred = d_ndef;
	a.n_unfinished = (u32*)((char*)c->counters.p + 52);
	a.work_counter = (u32*)((char*)c->counters.p + 60);
	if (n_ch != 1) {
		a.p.event_budget = 0; // (parked events are re-run per batch: single-chunk batches only)
	}
	a.cfg = c->tune.machine_cfg != ~0u ? c->tune.machine_cfg : machine_cfg_pick(a);
	*out = a;
	*blocks_out = blocks;
	*dyn_lds_out = dyn_lds;
	return 0;
}

// stream B, chunk j: its events through the event machine, in rounds
int
PolishRun::run_chunk_events(size_t j)
{
	int rc;
	const Chunk& ch = chunks[j];
	u64 n_ev = 0;
	u64* d_events = nullptr;
	u32* d_first = nullptr;
	if ((rc = extract_events(j, &n_ev, &d_events, &d_first)) || n_ev == 0) {
		return rc;
	}
	MachineArgs a;
	u64 blocks = 0;
	size_t dyn_lds = 0;
	if ((rc = machine_setup(n_ev, d_events, d_first, &a, &blocks, &dyn_lds))) {
		return rc;
	}

	// ---- rounds (see "event rounds" in nte_kernels.hip): primaries, then the secondaries their
	// primary's run does not overtake, then -- practically never -- whatever a verification rejects.
	// SNV mode runs everything in one round.
	// Rounds pay when the events outnumber the machine's threads several times over (3 Gbp: 4.4 M events, 42 ms against 55
	// in one round); a small batch is bound by the latency of its slowest events, which every round pays again
	// (375 Mbp, 0.44 M events: 9.0 ms in rounds, 7.0 ms in one; break-even at 1.8 M events).
	const bool rounds = !c->dp.snv && n_ev < 0xFFFFFF00ull && !c->tune.no_rounds &&
	                    (c->tune.force_rounds || n_ev >= (u64)c->cu_count * 8192);
	u32* d_list = nullptr;
	u64* d_before = nullptr;
	u64* d_bmax = nullptr;
	const u32 n32 = (u32)n_ev;
	const u32 sel_blocks = (n32 + EVR_TPB - 1) / EVR_TPB;
	const u32 pm_blocks = (n32 + 1023) / 1024;
	a.ev_cover = nullptr;
	a.ev_flags = nullptr;
	if (rounds) {
		if ((rc = ensure(c, c->ev_cover, n_ev * 8)) || (rc = ensure(c, c->ev_before, n_ev * 8)) ||
		    (rc = ensure(c, c->ev_flags, n_ev)) || (rc = ensure(c, c->ev_list, n_ev * 4)) ||
		    (rc = ensure(c, c->ev_bmax, (size_t)pm_blocks * 8 + 8))) {
			return rc;
		}
		a.ev_cover = (u64*)c->ev_cover.p;
		a.ev_flags = (u8*)c->ev_flags.p;
		d_list = (u32*)c->ev_list.p;
		d_before = (u64*)c->ev_before.p;
		d_bmax = (u64*)c->ev_bmax.p;
		HIP_TRY(c, hipMemsetAsync(a.ev_cover, 0, n_ev * 8, sB));
	}
	HIP_TRY(c, hipEventRecord(c->ev[3], sB));
	u32 n_def = 0;
	float p2_ms = 0.f;
	// A round in pieces: pass 1 (thread per event) of piece i + 1 runs NEXT TO the sweeps (wavefront per event) of
	// piece i, on two streams, each launch limited to half of a CU's wavefront slots.  Both kernels are bound by the
	// latency of dependent loads, not by the slots they hold (experiment 15), so the pair takes about as
	// long as the slower of the two.  The events of a round are independent of each other (that is what makes them
	// a round); the launches share the arena and the deferred list, both append-only, and nothing else: the sweep
	// launch has its own work counter and keeps its workspaces in LDS.
	auto run_round_pieces = [&](const u32* list, u32 count, u32 pieces, bool first_round) -> int {
		if (!c->stream3) {
			HIP_TRY(c, hipStreamCreate(&c->stream3));
		}
		hipStream_t sC = c->stream3;
		u32* wc2 = (u32*)((char*)c->counters.p + 72);
		HIP_TRY(c, hipMemsetAsync(d_ndef, 0, 4, sB));
		HIP_TRY(c, hipEventRecord(c->ev[5], sB));
		u32 nd_done = 0; // deferred events handed to a sweep launch so far
		u32 h_tail[4] = { 0, 0, 0, 0 };
		for (u32 i = 0; i < pieces && status == 0; i++) {
			const u32 b0 = (u32)((u64)count * i / pieces), b1 = (u32)((u64)count * (i + 1) / pieces);
			if (b1 > b0) {
				MachineArgs ra = a;
				ra.ev_list = list + b0;
				ra.n_events = b1 - b0;
				HIP_TRY(c, hipMemsetAsync(ra.work_counter, 0, 4, sB));
				const u64 want = ((u64)(b1 - b0) + MACHINE_TPB - 1) / MACHINE_TPB;
				const u64 cap = i == 0 ? blocks : (u64)c->cu_count * 2; // (piece 0 has the chip to itself)
				launch_k_machine_thread((unsigned)(want < cap ? want : cap), dyn_lds, sB, ra);
				HIP_TRY(c, hipGetLastError());
			}
			HIP_TRY(c, hipMemcpyAsync(h_tail, (char*)c->counters.p + 32, 16, hipMemcpyDeviceToHost, sB));
			HIP_TRY(c, hipStreamSynchronize(sB));
			status = h_tail[2];
			const u32 nd = h_tail[3];
			if (status != 0) {
				break;
			}
			const bool last = i + 1 == pieces;
			if (last) {
				// every sweep launch so far has to be over before the early copy below may read the arena
				HIP_TRY(c, hipStreamSynchronize(sC));
				HIP_TRY(c, hipMemcpyAsync(h_tail, (char*)c->counters.p + 32, 16, hipMemcpyDeviceToHost, sB));
				HIP_TRY(c, hipStreamSynchronize(sB));
				status = h_tail[2];
				if (status != 0) {
					break;
				}
				if (first_round && nd > 0 && n_ch == 1 && !c->tune.no_early_copy) {
					// everything written so far is final (the last sweep launch only appends chunks)
					early_chunks = h_tail[0] < arena_chunks ? h_tail[0] : arena_chunks;
					const u64 room = early_chunks + (u64)nd * 3 + 4096;
					int prc = pin_take(c, room * CHUNK_ITEMS * sizeof(Item) + 16, &early);
					if (prc) {
						return prc;
					}
					if (early_chunks) {
						HIP_TRY(c, hipMemcpyAsync(early.p, c->arena.p, early_chunks * CHUNK_ITEMS * sizeof(Item), hipMemcpyDeviceToHost, sA));
					}
				}
			}
			if (nd > nd_done) {
				// (the last one alone on the chip, on stream B; the others next to the following piece's pass 1)
				launch_wave_pass(a, (const u32*)c->deferred.p + nd_done, nd - nd_done, last ? sB : sC, wc2, last ? 8 : 2);
				HIP_TRY(c, hipGetLastError());
				nd_done = nd;
			}
		}
		HIP_TRY(c, hipStreamSynchronize(sC));
		HIP_TRY(c, hipEventRecord(c->ev[2], sB));
		HIP_TRY(c, hipMemcpyAsync(h_tail, (char*)c->counters.p + 32, 16, hipMemcpyDeviceToHost, sB));
		HIP_TRY(c, hipStreamSynchronize(sB));
		status = h_tail[2];
		n_def += nd_done;
		float t = 0.f;
		(void)hipEventElapsedTime(&t, c->ev[5], c->ev[2]);
		p2_ms += t; // (here: the whole round)
		return 0;
	};
	// one round = pass 1 over a list of events (indel sweeps postponed), pass 2 over the postponed ones
	auto run_round = [&](const u32* list, u32 count, bool first_round) -> int {
		if (count == 0) {
			return 0;
		}
		{
			u32 pieces = c->tune.machine_pieces;
			if (pieces == 0) {
				pieces = count >= c->cu_count * 4096u ? NTE_MACHINE_PIECES : 1;
			}
			if (pieces > 1 && list && count >= pieces && a.win_in_lds && wave_pass_in_lds(a)) {
				return run_round_pieces(list, count, pieces, first_round);
			}
		}
		MachineArgs ra = a;
		ra.ev_list = list;
		ra.n_events = count;
		if (a.p.snv && a.p.lanes && a.p.mode == 0 && !a.p.mask && !use_assess) {
			// -s 1: every position of every event is assessed, there is nothing a thread-per-event pass could settle
			// more cheaply -- all events go to the wavefront-per-event launch, 64 positions at a time (run_lanes)
			HIP_TRY(c, hipEventRecord(c->ev[5], sB));
			launch_wave_pass(ra, list, count);
			HIP_TRY(c, hipGetLastError());
			HIP_TRY(c, hipEventRecord(c->ev[2], sB));
			u32 h_t[4] = { 0, 0, 0, 0 };
			HIP_TRY(c, hipMemcpyAsync(h_t, (char*)c->counters.p + 32, 16, hipMemcpyDeviceToHost, sB));
			HIP_TRY(c, hipStreamSynchronize(sB));
			status = h_t[2];
			float t = 0.f;
			(void)hipEventElapsedTime(&t, c->ev[5], c->ev[2]);
			p2_ms += t;
			return 0;
		}
		HIP_TRY(c, hipMemsetAsync(d_ndef, 0, 4, sB));
		HIP_TRY(c, hipMemsetAsync(ra.work_counter, 0, 4, sB));
		const u64 want = ((u64)count + MACHINE_TPB - 1) / MACHINE_TPB;
		launch_k_machine_thread((unsigned)(want < blocks ? want : blocks), dyn_lds, sB, ra);
		HIP_TRY(c, hipGetLastError());
		u32 h_tail[4] = { 0, 0, 0, 0 };
		HIP_TRY(c, hipMemcpyAsync(h_tail, (char*)c->counters.p + 32, 16, hipMemcpyDeviceToHost, sB));
		HIP_TRY(c, hipEventRecord(c->ev[5], sB));
		HIP_TRY(c, hipStreamSynchronize(sB));
		const u32 nd = h_tail[3];
		n_def += nd;
		status = h_tail[2];
		if (first_round && nd > 0 && status == 0 && n_ch == 1 && !c->tune.no_early_copy) {
			// everything pass 1 wrote is final (later launches only append chunks): start moving it
			// to the host on the other stream while the sweeps run
			early_chunks = h_tail[0] < arena_chunks ? h_tail[0] : arena_chunks;
			const u64 room = early_chunks + (u64)nd * 3 + 4096;
			int prc = pin_take(c, room * CHUNK_ITEMS * sizeof(Item) + 16, &early);
			if (prc) {
				return prc;
			}
			if (early_chunks) {
				HIP_TRY(c, hipMemcpyAsync(early.p, c->arena.p, early_chunks * CHUNK_ITEMS * sizeof(Item), hipMemcpyDeviceToHost, sA));
			}
		}
		if (nd > 0 && status == 0) {
			MachineArgs a2 = ra;
#ifdef NTE_ABLATION
			if (const char* dbg = getenv("NTEDIT_HIP_PASS2_DEBUG")) {
				a2.p.debug_stop = (u32)atoi(dbg); // timing ablations; results are NOT valid (ablation build only)
			}
#endif
			launch_wave_pass(a2, (const u32*)c->deferred.p, nd);
			HIP_TRY(c, hipGetLastError());
			HIP_TRY(c, hipEventRecord(c->ev[2], sB));
			HIP_TRY(c, hipMemcpyAsync(h_tail, (char*)c->counters.p + 32, 16, hipMemcpyDeviceToHost, sB));
			HIP_TRY(c, hipStreamSynchronize(sB));
			status = h_tail[2];
			float t = 0.f;
			(void)hipEventElapsedTime(&t, c->ev[5], c->ev[2]);
			p2_ms += t;
		}
		return 0;
	};
	u32 n_A = n32, n_B = 0, n_C = 0;
	if (!rounds) {
		if ((rc = run_round(nullptr, n32, true))) {
			return rc;
		}
	} else {
		const u32 gap = c->dp.k + 16;
		auto list_count = [&](u32* out) -> int {
			HIP_TRY(c, hipMemcpyAsync(out, d_list_n, 4, hipMemcpyDeviceToHost, sB));
			HIP_TRY(c, hipStreamSynchronize(sB));
			return 0;
		};
		auto prefix_max = [&]() {
			hipLaunchKernelGGL(k_ev_prefix_max_1, dim3(pm_blocks), dim3(1024), 0, sB, (const u64*)a.ev_cover, n32, d_before, d_bmax);
			hipLaunchKernelGGL(k_ev_prefix_max_2, dim3(pm_blocks), dim3(1024), 0, sB, n32, d_before, (const u64*)d_bmax);
		};
		auto select = [&](int mode, u32* count) -> int {
			prefix_max();
			HIP_TRY(c, hipMemsetAsync(d_list_n, 0, 4, sB));
			if (mode == 0) {
				hipLaunchKernelGGL(k_ev_select<0>, dim3(sel_blocks), dim3(EVR_TPB), 0, sB, (const u64*)d_events, n32,
				                   (const u64*)a.ev_cover, (const u64*)d_before, a.ev_flags, d_first, d_list, d_list_n);
			} else {
				hipLaunchKernelGGL(k_ev_select<1>, dim3(sel_blocks), dim3(EVR_TPB), 0, sB, (const u64*)d_events, n32,
				                   (const u64*)a.ev_cover, (const u64*)d_before, a.ev_flags, d_first, d_list, d_list_n);
			}
			return list_count(count);
		};
		HIP_TRY(c, hipMemsetAsync(d_list_n, 0, 4, sB));
		hipLaunchKernelGGL(k_ev_primaries, dim3(sel_blocks), dim3(EVR_TPB), 0, sB, (const u64*)d_events, n32, gap, a.ev_flags, d_list, d_list_n);
		if ((rc = list_count(&n_A)) || (rc = run_round(d_list, n_A, true))) {
			return rc;
		}
		if (status == 0 && n_A < n32) {
			if ((rc = select(0, &n_B)) || (rc = run_round(d_list, n_B, false))) {
				return rc;
			}
			// (a run of round B that reaches a primary: that primary's secondaries cannot be taken for overtaken)
			u32 n_v = 0;
			for (int guard = 0; status == 0 && guard < 64; guard++) {
				if ((rc = select(1, &n_v))) {
					return rc;
				}
				if (n_v == 0) {
					break;
				}
				n_C += n_v;
				if ((rc = run_round(d_list, n_v, false))) {
					return rc;
				}
			}
		}
	}
	HIP_TRY(c, hipGetLastError());
	keep_a = a;
	HIP_TRY(c, hipEventRecord(c->ev[4], sB));
	u32 h_tail[4] = { 0, 0, 0, 0 };
	HIP_TRY(c, hipMemcpyAsync(h_tail, (char*)c->counters.p + 32, 16, hipMemcpyDeviceToHost, sB));
	HIP_TRY(c, hipStreamSynchronize(sB));
	status = h_tail[2];
	deferred_total += n_def;
	skipped_total += rounds ? (u64)n32 - n_A - n_B - n_C : 0;
	float p_all = 0.f;
	(void)hipEventElapsedTime(&p_all, c->ev[3], c->ev[4]);
	ms_machine += p_all;
	if (getenv("NTEDIT_HIP_DEBUG")) {
		fprintf(
		    stderr,
		    "[ntedit_hip] chunk %zu/%zu contigs %u-%u events %llu (round A %u, B %u, C %u; %llu skipped as overtaken) sweeps %u "
		    "machine %.3f ms (sweep launches %.3f ms) arena %u status %u window %u\n",
		    j + 1, n_ch, ch.c0, ch.c1, (unsigned long long)n_ev, n_A, n_B, n_C,
		    (unsigned long long)(rounds ? (u64)n32 - n_A - n_B - n_C : 0), n_def, p_all, p2_ms, h_tail[0], status, c->dp.node_window);
		unsigned long long pr[64];
		machine_wave_profile(pr);
		const unsigned long long tg = machine_thread_gathers();
		if (pr[15] || tg) {
			fprintf(stderr, "[ntedit_hip] machine filter gathers (profile build): thread-per-event launches %llu, wavefront-per-event launches %llu\n", tg, pr[15]);
		}
		if (pr[8]) {
			fprintf(stderr, "[ntedit_hip] wave-kernel events by duration (log2 cycles: count):");
			for (int b = 0; b < 32; b++) {
				if (pr[32 + b]) {
					fprintf(stderr, " %d:%llu", b, pr[32 + b]);
				}
			}
			fprintf(stderr, "; longest %llu cycles, most positions in one event %llu\n", pr[13], pr[14]);
			fprintf(stderr, "[ntedit_hip] wave-kernel phase cycles/event (n=%llu): seed %llu presence %llu first-miss %llu later-miss %llu advance %llu loop %llu housekeeping %llu flush %llu; positions/event %.1f failing %.1f\n",
			    pr[8], pr[0] / pr[8], pr[1] / pr[8], pr[2] / pr[8], pr[3] / pr[8], pr[4] / pr[8], pr[5] / pr[8], pr[6] / pr[8], pr[7] / pr[8],
			    (double)pr[9] / (double)pr[8], (double)pr[10] / (double)pr[8]);
			fprintf(stderr, "[ntedit_hip]   inside failing positions: window %llu step-2 %llu substitutions (lanes: the lanes' own phases) %llu indel sweeps %llu apply %llu; advance: stride %llu roll %llu; lane batches/event %.2f lanes/batch %.1f\n",
			    pr[16] / pr[8], pr[17] / pr[8], pr[18] / pr[8], pr[20] / pr[8], pr[19] / pr[8], pr[21] / pr[8], pr[22] / pr[8],
			    (double)pr[11] / (double)pr[8], pr[11] ? (double)pr[12] / (double)pr[11] : 0.0);
		}
	}
	ev_total += n_ev;
	return 0;
}

// edit records to the host; *redo = the batch has to be run again with more room
int
PolishRun::collect(bool* redo)
{
	int rc;
	*redo = false;
	u32 h_tail[4] = { 0, 0, 0, 0 };
	HIP_TRY(c, hipMemcpy(h_tail, (char*)c->counters.p + 32, 16, hipMemcpyDeviceToHost));
	const u64 used_chunks = h_tail[0] < arena_chunks ? h_tail[0] : arena_chunks;
	r->st.absent_kmers = absent_total;
	r->st.events = ev_total;
	r->st.events_deferred = deferred_total;
	r->arena_items = (size_t)used_chunks * CHUNK_ITEMS;
	u64 have = 0; // chunks already on the host
	if (early.p && early.cap >= r->arena_items * sizeof(Item) + 16) {
		r->arena_buf = early;
		early.p = nullptr;
		early.cap = 0;
		have = early_chunks;
	} else {
		pin_give(c, early);
		if ((rc = pin_take(c, r->arena_items * sizeof(Item) + 16, &r->arena_buf))) {
			return rc;
		}
	}
	if ((rc = pin_take(c, ev_total * 4 + 16, &r->first_buf))) {
		return rc;
	}
	if (used_chunks > have) {
		const size_t off = (size_t)have * CHUNK_ITEMS * sizeof(Item);
		HIP_TRY(c, hipMemcpyAsync((char*)r->arena_buf.p + off, (char*)c->arena.p + off, r->arena_items * sizeof(Item) - off, hipMemcpyDeviceToHost, sB));
	}
	if (ev_total) {
		HIP_TRY(c, hipMemcpyAsync(r->first_buf.p, c->first_chunk.p, ev_total * 4, hipMemcpyDeviceToHost, sB));
	}
	HIP_TRY(c, hipStreamSynchronize(sB));
	// Events parked by the budget: decide, in serial order, which of them are applied, re-run
	// exactly those to completion, carry on behind them (host/resolve.h).  Nothing to do in
	// the ordinary case.
	u32 n_unfinished = 0;
	HIP_TRY(c, hipMemcpy(&n_unfinished, (char*)c->counters.p + 52, 4, hipMemcpyDeviceToHost));
	if (n_unfinished && n_ch == 1) {
		nte_host::Resolver rs((const Item*)r->arena_buf.p, r->arena_items, (const u32*)r->first_buf.p, ev_total);
		std::vector<u32> rerun;
		if (!rs.start(rerun)) {
			return fail(c, NTEDIT_E_DEVICE, "malformed event records");
		}
		u64 have_chunks = used_chunks;
		unsigned rounds = 0;
		// The walk hands out one parked event per contig and round: the first one the serial order reaches.  That is
		// the cheapest plan when few events are parked (the ordinary case: a handful per 3 Gbp) or when the first
		// re-run covers its whole contig (a saturated filter), and a crawl when nearly everything is parked and
		// every run is short (a tiny budget: one launch per dependency level).  After NTE_RESOLVE_ROUNDS rounds the
		// re-runs are widened instead: ALL events still parked behind the waiting points at once, with a budget
		// that grows 16-fold per round (speculation again, bounded by that budget), until none is left.
		u32 wide_budget = keep_a.p.event_budget ? keep_a.p.event_budget : 1;
		while (!rerun.empty()) {
			const bool wide = rounds >= NTE_RESOLVE_ROUNDS;
			if (wide) {
				rerun.clear();
				rs.parked_behind(rerun);
				wide_budget = wide_budget < (1u << 26) ? wide_budget * 16u : 0u;
			}
			HIP_TRY(c, hipMemcpyAsync(c->deferred.p, rerun.data(), rerun.size() * 4, hipMemcpyHostToDevice, sB));
			MachineArgs ra = keep_a;
			ra.p.event_budget = wide ? wide_budget : 0;
			ra.ev_cover = nullptr; // (the rounds are over)
			ra.ev_flags = nullptr;
			launch_wave_pass(ra, (const u32*)c->deferred.p, (u32)rerun.size());
			HIP_TRY(c, hipGetLastError());
			u32 t2[4] = { 0, 0, 0, 0 };
			HIP_TRY(c, hipMemcpyAsync(t2, (char*)c->counters.p + 32, 16, hipMemcpyDeviceToHost, sB));
			HIP_TRY(c, hipStreamSynchronize(sB));
			if (t2[2]) {
				// a re-run ran out of arena / rope window: the whole batch again, with more room
				status = t2[2];
				*redo = true;
				pin_give(c, r->arena_buf);
				pin_give(c, r->first_buf);
				return 0;
			}
			const u64 now_chunks = t2[0] < arena_chunks ? t2[0] : arena_chunks;
			const size_t need = (size_t)now_chunks * CHUNK_ITEMS * sizeof(Item) + 16;
			if (r->arena_buf.cap < need) {
				PinBuf bigger;
				if ((rc = pin_take(c, need + need / 2, &bigger))) {
					return rc;
				}
				memcpy(bigger.p, r->arena_buf.p, (size_t)have_chunks * CHUNK_ITEMS * sizeof(Item));
				pin_give(c, r->arena_buf);
				r->arena_buf = bigger;
			}
			if (now_chunks > have_chunks) {
				const size_t off = (size_t)have_chunks * CHUNK_ITEMS * sizeof(Item);
				HIP_TRY(c, hipMemcpyAsync((char*)r->arena_buf.p + off, (char*)c->arena.p + off,
				                          (size_t)(now_chunks - have_chunks) * CHUNK_ITEMS * sizeof(Item), hipMemcpyDeviceToHost, sB));
			}
			HIP_TRY(c, hipMemcpyAsync(r->first_buf.p, c->first_chunk.p, ev_total * 4, hipMemcpyDeviceToHost, sB));
			HIP_TRY(c, hipStreamSynchronize(sB));
			have_chunks = now_chunks;
			r->arena_items = (size_t)now_chunks * CHUNK_ITEMS;
			rs.rebind((const Item*)r->arena_buf.p, r->arena_items, (const u32*)r->first_buf.p);
			rerun.clear();
			if (!rs.resume(rerun, wide) || ++rounds > 10000000u) {
				return fail(c, NTEDIT_E_DEVICE, "parked events could not be resolved");
			}
		}
		if (getenv("NTEDIT_HIP_DEBUG")) {
			fprintf(stderr, "[ntedit_hip] %u events parked by the budget, %u re-run round(s)\n", n_unfinished, rounds);
		}
	}
	return 0;
}

int
PolishRun::finish()
{
	HIP_TRY(c, hipEventRecord(c->ev[4], sB));
	HIP_TRY(c, hipStreamSynchronize(sB));
	// one entry per event, in position order; NONE32 = the event produced nothing (the renderer skips those)
	r->n_ev_first = ev_total;
	// timings: screening = sum of its launches (they may overlap machine kernels)
	float ms_screen = 0.f;
	const size_t n_scr = pipelined ? n_ch : 1;
	for (size_t j = 0; j < n_scr; j++) {
		float t = 0.f;
		(void)hipEventElapsedTime(&t, c->chunk_ev[2 * j], c->chunk_ev[2 * j + 1]);
		ms_screen += t;
	}
	r->st.ms_screen = ms_screen;
	r->st.screen_launches = h2d_launches ? h2d_launches : (uint32_t)n_scr;
	if (!pipelined && !h2d_launches && c->bin_chunks_last) {
		r->st.screen_binned = 1;
		r->st.screen_launches = c->bin_chunks_last;
		for (u32 q = 0; q < c->bin_chunks_last; q++) {
			float tp = 0.f, tq = 0.f;
			(void)hipEventElapsedTime(&tp, c->bin_ev[4 * q], c->bin_ev[4 * q + 1]);
			(void)hipEventElapsedTime(&tq, c->bin_ev[4 * q + 2], c->bin_ev[4 * q + 3]);
			r->st.ms_partition += tp;
			r->st.ms_probe += tq;
		}
	}
	r->st.events_skipped = (uint32_t)skipped_total;
	r->st.ms_machine = ms_machine;
	r->st.ms_extract = 0.f;
	if (use_assess) {
		(void)hipEventElapsedTime(&r->st.ms_extract, c->ev_assess[0], c->ev_assess[1]); // (the run map, k_assess)
	}
	HIP_TRY(c, hipEventElapsedTime(&r->st.ms_total, c->ev[0], c->ev[4]));
	c->last_ms = ms_screen;
	return 0;
}

} // namespace

extern "C" int
ntedit_hip_polish_batch(
    ntedit_hip_ctx* c,
    const char* bases,
    uint64_t n,
    const uint64_t* offsets,
    const uint32_t* lens,
    uint32_t n_contigs,
    int on_device,
    ntedit_hip_result** out)
{
	if (!c || !out || (n && !bases) || (n_contigs && (!offsets || !lens)) || on_device < 0 || on_device > NTEDIT_HIP_BASES_PACKED) {
		return fail(c, NTEDIT_E_ARG, "polish_batch: bad argument");
	}
	*out = nullptr;
	HIP_TRY(c, hipSetDevice(c->device));
	int rc = refresh_params(c);
	if (rc) {
		return rc;
	}
	for (u32 i = 0; i < n_contigs; i++) {
		if (offsets[i] + lens[i] > n || (i + 1 < n_contigs && offsets[i] + lens[i] >= offsets[i + 1])) {
			return fail(c, NTEDIT_E_ARG, "polish_batch: contig %u breaks the batch layout", i);
		}
	}
	ntedit_hip_result* r = new ntedit_hip_result();
	r->owner = c;
	memset(&r->st, 0, sizeof r->st);
	r->st.bases = n;
	r->snv = c->hp.snv ? 1 : 0;
	if (n == 0 || n_contigs == 0) {
		*out = r;
		return 0;
	}
	PolishRun run;
	run.c = c;
	run.bases = bases;
	run.n = n;
	run.offsets = offsets;
	run.lens = lens;
	run.n_contigs = n_contigs;
	run.on_device = on_device;
	run.r = r;
	if ((rc = run.plan())) {
		return run.bail(rc);
	}
	for (int attempt = 0;; attempt++) {
		if ((rc = run.begin_attempt()) || (rc = run.launch_screening(attempt))) {
			return run.bail(rc);
		}
		for (size_t j = 0; j < run.n_ch && run.status == 0; j++) {
			if ((rc = run.run_chunk_events(j))) {
				return run.bail(rc);
			}
		}
		if (hipStreamSynchronize(run.sA) != hipSuccess || hipStreamSynchronize(run.sB) != hipSuccess) {
			return run.bail(fail(c, NTEDIT_E_DEVICE, "stream synchronisation failed: %s", hipGetErrorString(hipGetLastError())));
		}
		{
			bool lost = false;
			if ((rc = bin_records_lost(c, &lost))) {
				return run.bail(rc);
			}
			if (lost) {
				// the bitmap of this attempt is void; the context screens with the direct kernel from now on.  (The loss
				// is deterministic and shows on the first attempt; a later one is out of retries rather than trusted.)
				if (attempt >= 4) {
					return run.bail(fail(c, NTEDIT_E_OVERFLOW, "the screening lost probe records and no retry is left"));
				}
				continue;
			}
		}
		if (run.status == 0) {
			bool redo = false;
			if ((rc = run.collect(&redo))) {
				return run.bail(rc);
			}
			if (!redo) {
				if ((rc = run.finish())) {
					return run.bail(rc);
				}
				break;
			}
		}
		if (attempt >= 4) {
			return run.bail(fail(c, NTEDIT_E_OVERFLOW, "event machine ran out of room (status %u)", run.status));
		}
		if (run.status & EV_ARENA_FULL) {
			run.arena_chunks *= 4;
		}
		if (run.status & EV_OVERFLOW) {
			c->dp.node_window *= 2;
		}
	}
	*out = r;
	return 0;
}

extern "C" {

void
ntedit_hip_result_free(ntedit_hip_result* r)
{
	if (!r) {
		return;
	}
	// (a result may outlive its context: its pinned buffers are then released directly)
	pin_give(r->owner, r->arena_buf);
	pin_give(r->owner, r->first_buf);
	delete r;
}

int
ntedit_hip_result_stats(const ntedit_hip_result* r, ntedit_hip_stats* s)
{
	if (!r || !s) {
		return NTEDIT_E_ARG;
	}
	*s = r->st;
	s->events_applied = r->rst.events_applied;
	s->substitutions = r->rst.substitutions;
	s->insertions = r->rst.insertions;
	s->deletions = r->rst.deletions;
	return 0;
}

struct ntedit_hip_annot
{
	nte_host::Annotations* a = nullptr;
};

int
ntedit_hip_annot_load(const char* vcf_path, ntedit_hip_annot** out)
{
	if (!vcf_path || !out) {
		return NTEDIT_E_ARG;
	}
	nte_host::Annotations* a = nte_host::annotations_load(vcf_path);
	if (!a) {
		return NTEDIT_E_IO;
	}
	*out = new ntedit_hip_annot();
	(*out)->a = a;
	return 0;
}

void
ntedit_hip_annot_free(ntedit_hip_annot* a)
{
	if (a) {
		nte_host::annotations_free(a->a);
		delete a;
	}
}

int
ntedit_hip_write_vcf_header(const char* vcf_path, const char* draft_filename)
{
	FILE* v = fopen(vcf_path, "wb");
	if (!v) {
		return NTEDIT_E_IO;
	}
	nte_host::write_vcf_header(v, draft_filename ? draft_filename : "");
	return fclose(v) == 0 ? 0 : NTEDIT_E_IO;
}

int
ntedit_hip_write_outputs_ex(
    const ntedit_hip_result* r,
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    const char* const* names,
    uint32_t n_contigs,
    const ntedit_hip_write_options* wo)
{
	if (!r || !wo || (n_contigs && (!bases || !offsets || !lens || !names))) {
		return NTEDIT_E_ARG;
	}
	const char* fa_path = wo->fa_path;
	const char* tsv_path = wo->tsv_path;
	const char* vcf_path = wo->vcf_path;
	const int append = wo->append;
	FILE* fa = fa_path ? fopen(fa_path, append ? "ab" : "wb") : nullptr;
	FILE* tsv = tsv_path ? fopen(tsv_path, append ? "ab" : "wb") : nullptr;
	FILE* vcf = vcf_path ? fopen(vcf_path, append ? "ab" : "wb") : nullptr;
	if ((fa_path && !fa) || (tsv_path && !tsv) || (vcf_path && !vcf)) {
		if (fa) {
			fclose(fa);
		}
		if (tsv) {
			fclose(tsv);
		}
		if (vcf) {
			fclose(vcf);
		}
		return NTEDIT_E_IO;
	}
	if (fa) {
		setvbuf(fa, nullptr, _IOFBF, 4 << 20);
	}
	if (tsv) {
		setvbuf(tsv, nullptr, _IOFBF, 1 << 20);
	}
	if (vcf) {
		setvbuf(vcf, nullptr, _IOFBF, 1 << 20);
	}
	ntedit_hip_result* rw = const_cast<ntedit_hip_result*>(r);
	rw->rst = nte_host::RenderStats();
	nte_host::RenderOptions opt;
	opt.threads = g_host_threads.load();
	opt.snv = r->snv != 0;
	opt.annot = wo->annot ? wo->annot->a : nullptr;
	opt.segments = wo->segments;
	opt.out_sizes = wo->out_sizes;
	if (wo->out_sizes) {
		memset(wo->out_sizes, 0, (size_t)n_contigs * 3 * sizeof(uint64_t));
	}
	int rc = nte_host::render_batch(
	    (const Item*)r->arena_buf.p,
	    r->arena_items,
	    (const u32*)r->first_buf.p,
	    r->n_ev_first,
	    bases,
	    offsets,
	    lens,
	    names,
	    n_contigs,
	    fa,
	    tsv,
	    &rw->rst,
	    vcf,
	    &opt);
	if (fa && fclose(fa) != 0) {
		rc = rc ? rc : -5;
	}
	if (tsv && fclose(tsv) != 0) {
		rc = rc ? rc : -5;
	}
	if (vcf && fclose(vcf) != 0) {
		rc = rc ? rc : -5;
	}
	if (rc == -7 || rc == -8) {
		return NTEDIT_E_SEGMENT;
	}
	return rc ? NTEDIT_E_IO : 0;
}

int
ntedit_hip_write_outputs_vcf(
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
    const ntedit_hip_annot* annot)
{
	(void)snv; // (SNV mode follows the parameters the batch was polished with)
	ntedit_hip_write_options wo;
	memset(&wo, 0, sizeof wo);
	wo.fa_path = fa_path;
	wo.tsv_path = tsv_path;
	wo.vcf_path = vcf_path;
	wo.append = append;
	wo.annot = annot;
	return ntedit_hip_write_outputs_ex(r, bases, offsets, lens, names, n_contigs, &wo);
}

int
ntedit_hip_result_cover_ends(const ntedit_hip_result* r, uint32_t n_contigs, uint32_t* cover_ends)
{
	if (!r || (n_contigs && !cover_ends)) {
		return NTEDIT_E_ARG;
	}
	return nte_host::cover_ends((const Item*)r->arena_buf.p, r->arena_items, (const u32*)r->first_buf.p, r->n_ev_first,
	                            n_contigs, cover_ends)
	           ? NTEDIT_E_ARG
	           : 0;
}

int
ntedit_hip_result_cuts_ok(const ntedit_hip_result* r, uint32_t n_contigs, const uint32_t* lens, const ntedit_hip_segment* segments, uint8_t* ok)
{
	if (!r || (n_contigs && (!lens || !segments || !ok))) {
		return NTEDIT_E_ARG;
	}
	std::vector<u32> halos(n_contigs);
	for (u32 i = 0; i < n_contigs; i++) {
		halos[i] = segments[i].halo;
	}
	return nte_host::cuts_ok((const Item*)r->arena_buf.p, r->arena_items, (const u32*)r->first_buf.p, r->n_ev_first, n_contigs, lens,
	                         halos.data(), ok)
	           ? NTEDIT_E_ARG
	           : 0;
}

int
ntedit_hip_result_edits(
    ntedit_hip_result* r,
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    uint32_t n_contigs,
    const ntedit_hip_segment* segments,
    const ntedit_hip_edit** edits,
    uint64_t* n_edits,
    const char** base_pool)
{
	if (!r || !edits || !n_edits || (n_contigs && (!bases || !offsets || !lens))) {
		return NTEDIT_E_ARG;
	}
	if (!r->edits_built) {
		r->edits.clear();
		r->edit_pool.clear();
		std::vector<const char*> names(n_contigs, "");
		nte_host::RenderOptions opt;
		opt.threads = g_host_threads.load();
		opt.snv = r->snv != 0;
		opt.segments = segments;
		opt.edits = &r->edits;
		opt.edit_pool = &r->edit_pool;
		nte_host::RenderStats st;
		const int rc = nte_host::render_batch(
		    (const Item*)r->arena_buf.p, r->arena_items, (const u32*)r->first_buf.p, r->n_ev_first, bases, offsets, lens,
		    names.data(), n_contigs, nullptr, nullptr, &st, nullptr, &opt);
		if (rc) {
			r->edits.clear();
			r->edit_pool.clear();
			return rc == -7 || rc == -8 ? NTEDIT_E_SEGMENT : NTEDIT_E_ARG;
		}
		r->edits_built = true;
	}
	*edits = r->edits.data();
	*n_edits = r->edits.size();
	if (base_pool) {
		*base_pool = r->edit_pool.c_str();
	}
	return 0;
}

int
ntedit_hip_write_outputs(
    const ntedit_hip_result* r,
    const char* bases,
    const uint64_t* offsets,
    const uint32_t* lens,
    const char* const* names,
    uint32_t n_contigs,
    const char* fa_path,
    const char* tsv_path,
    int append)
{
	return ntedit_hip_write_outputs_vcf(r, bases, offsets, lens, names, n_contigs, fa_path, tsv_path, nullptr, append, 0, nullptr);
}

int
ntedit_hip_write_tsv_header(const char* tsv_path, uint32_t k, uint32_t jump, int counting)
{
	FILE* tsv = fopen(tsv_path, "wb");
	if (!tsv) {
		return NTEDIT_E_IO;
	}
	nte_host::write_tsv_header(tsv, k, jump, counting != 0);
	return fclose(tsv) == 0 ? 0 : NTEDIT_E_IO;
}

void*
ntedit_hip_host_alloc(size_t bytes)
{
	void* p = nullptr;
	if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
		return nullptr;
	}
	return p;
}

void
ntedit_hip_host_free(void* p)
{
	if (p) {
		(void)hipHostFree(p);
	}
}

int
ntedit_hip_bind_near_device(int device)
{
	if (getenv("NTEDIT_HIP_NO_BIND")) {
		return -1;
	}
	char id[64] = { 0 };
	if (hipDeviceGetPCIBusId(id, (int)sizeof id - 1, device) != hipSuccess) {
		return -1;
	}
	for (char* q = id; *q; q++) {
		*q = (char)tolower((unsigned char)*q);
	}
	char path[256];
	snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", id);
	FILE* f = fopen(path, "r");
	if (!f) {
		return -1;
	}
	int node = -1;
	if (fscanf(f, "%d", &node) != 1) {
		node = -1;
	}
	fclose(f);
	if (node < 0) {
		return -1;
	}
	snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
	f = fopen(path, "r");
	if (!f) {
		return -1;
	}
	// "0-63,128-191"
	cpu_set_t want;
	CPU_ZERO(&want);
	int lo = 0, hi = 0, n_set = 0;
	for (;;) {
		if (fscanf(f, "%d", &lo) != 1) {
			break;
		}
		hi = lo;
		int ch = fgetc(f);
		if (ch == '-') {
			if (fscanf(f, "%d", &hi) != 1) {
				break;
			}
			ch = fgetc(f);
		}
		for (int cpu = lo; cpu <= hi && cpu < CPU_SETSIZE; cpu++) {
			CPU_SET(cpu, &want);
			n_set++;
		}
		if (ch != ',') {
			break;
		}
	}
	fclose(f);
	// only CPUs the process may use anyway; nothing to do when that leaves none (or all)
	cpu_set_t have;
	if (n_set == 0 || sched_getaffinity(0, sizeof have, &have) != 0) {
		return -1;
	}
	cpu_set_t both;
	CPU_AND(&both, &want, &have);
	if (CPU_COUNT(&both) == 0 || CPU_COUNT(&both) == CPU_COUNT(&have)) {
		return -1;
	}
	if (sched_setaffinity(0, sizeof both, &both) != 0) {
		return -1;
	}
	return node;
}

void
ntedit_hip_set_host_threads(unsigned n)
{
	g_host_threads.store(n);
}

float
ntedit_hip_last_kernel_ms(const ntedit_hip_ctx* c)
{
	return c ? c->last_ms : 0.f;
}

#ifndef NTE_BUILD_ID
#define NTE_BUILD_ID "unknown"
#endif
const char*
ntedit_hip_build_id(void)
{
	return NTE_BUILD_ID;
}

int
ntedit_hip_set_tuning(ntedit_hip_ctx* c, const char* key, uint64_t value)
{
	if (!c || !key) {
		return fail(c, NTEDIT_E_ARG, "set_tuning: bad argument");
	}
	const std::string k(key);
	auto& t = c->tune;
	if (k == "screen_mode") {
		t.screen_mode = (u32)value;
	} else if (k == "bin_chunk") {
		t.bin_chunk = value;
	} else if (k == "bin_cap_percent") {
		t.bin_cap_percent = (u32)value;
	} else if (k == "force_xcc") {
		t.force_xcc = (u32)value;
	} else if (k == "bin_timing") {
		t.bin_timing = (u32)value;
	} else if (k == "chunk_bytes") {
		t.chunk_bytes = value;
	} else if (k == "h2d_piece") {
		t.h2d_piece = value;
	} else if (k == "inline_tries") {
		t.inline_tries = (u32)value;
		c->dp_valid = false;
	} else if (k == "assess") {
		t.assess = (u32)value;
	} else if (k == "machine_cfg") {
		t.machine_cfg = value ? ~0u : 0u; // (only "general" can be forced: a specialised instantiation is wrong for other configurations)
	} else if (k == "lanes") {
		t.lanes = (u32)value;
		c->dp_valid = false;
	} else if (k == "defer_run") {
		t.defer_run = (u32)value;
		c->dp_valid = false;
	} else if (k == "screen_lds_pad") {
		t.screen_lds_pad = (u32)value;
	} else if (k == "force_rounds") {
		t.force_rounds = (u32)value;
	} else if (k == "machine_pieces") {
		t.machine_pieces = (u32)value;
	} else if (k == "no_rounds") {
		t.no_rounds = (u32)value;
	} else if (k == "no_early_copy") {
		t.no_early_copy = (u32)value;
	} else if (k == "no_lds_ws") {
		t.no_lds_ws = (u32)value;
	} else if (k == "probe_parts_log2") {
		t.probe_parts_log2 = (u32)value < 4 ? (u32)value : 3;
	} else if (k == "bin_scatter") {
		t.bin_scatter = (u32)value;
	} else if (k == "bin_overlap") {
		t.bin_overlap = (u32)value;
	} else if (k == "h2d_chunks") {
		t.h2d_chunks = (u32)value;
	} else if (k == "records_uncached") {
		t.records_uncached = (u32)value;
		for (int q = 0; q < 2; q++) {
			if (c->bin_records[q].uncached != (value != 0)) {
				release(c->bin_records[q]);
				c->bin_records[q].uncached = value != 0;
			}
		}
	} else if (k == "bin_fallback") {
		c->bin_fallback = value != 0;
	} else {
		return fail(c, NTEDIT_E_ARG, "set_tuning: unknown key '%s'", key);
	}
	return 0;
}

int
ntedit_hip_gather_bench(ntedit_hip_ctx* c, uint64_t nbytes, uint64_t n_probes, double* probes_per_s, float* ms)
{
	if (!c || nbytes < 4096 || (nbytes & (nbytes - 1))) {
		return fail(c, NTEDIT_E_ARG, "gather_bench: nbytes must be a power of two");
	}
	HIP_TRY(c, hipSetDevice(c->device));
	u8* buf = nullptr;
	u32* sink = nullptr;
	HIP_TRY(c, hipMalloc((void**)&buf, nbytes));
	HIP_TRY(c, hipMalloc((void**)&sink, 64));
	HIP_TRY(c, hipMemset(buf, 0x5A, nbytes));
	const u64 threads = (u64)c->cu_count * 2048 * 2;
	u64 per_thread = (n_probes + threads - 1) / threads;
	per_thread = (per_thread + 11) / 12 * 12;
	const u64 mask = nbytes * 8 - 1;
	// warm-up + timed run
	hipLaunchKernelGGL(k_gather, dim3((unsigned)(threads / 256)), dim3(256), 0, c->stream, buf, mask, (u64)12, sink);
	HIP_TRY(c, hipEventRecord(c->ev[0], c->stream));
	hipLaunchKernelGGL(k_gather, dim3((unsigned)(threads / 256)), dim3(256), 0, c->stream, buf, mask, per_thread, sink);
	HIP_TRY(c, hipEventRecord(c->ev[1], c->stream));
	HIP_TRY(c, hipStreamSynchronize(c->stream));
	float t = 0.f;
	HIP_TRY(c, hipEventElapsedTime(&t, c->ev[0], c->ev[1]));
	(void)hipFree(buf);
	(void)hipFree(sink);
	if (ms) {
		*ms = t;
	}
	if (probes_per_s) {
		*probes_per_s = (double)(per_thread * threads) / ((double)t * 1e-3);
	}
	c->last_ms = t;
	return 0;
}

} // extern "C"
