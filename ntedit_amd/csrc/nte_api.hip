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
#include <chrono>
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
	DevBuf candmap; // -s 1 on a plain filter: 4 bits per position, the first-probe bits of its substitution candidates (k_wc_scatter_b<3, ., 1>)
	DevBuf seq, bitmap, block_counts, block_offsets, events, first_chunk, arena, counters, deferred;
	DevBuf ws_nodes, ws_ov_pos, ws_ov_chr, ws_prev, ws_lps, ws_win;
	DevBuf offs, lens;
	DevBuf bin_records, bin_fill, bin_ctl, bin_ovf, bin_state; // the binned screening: records, run fills, probe control words, overflow list, chunk flags
	struct BinRange
	{
		u64 begin, end;  // k-mer starts of one record chunk
		bool recovered;  // its overflow list ran out and the direct kernel has screened it again
		bool gate;       // a chunk of the candidate map (-s 1): "screened again" = its part of the map set to unknown
	};
	std::vector<BinRange> bin_ranges; // the record chunks of the current call, in launch order (bin_state's flags)
	std::vector<u32> bin_flags_host;
	u32 bin_chunks_direct = 0;        // chunks of the current call the direct kernel had to screen again
	u64 bin_ovf_entries = 0;          // overflow entries handed out in the current call (as of the last bin_recover)
	hipStream_t stream_copy = nullptr; // H2D pieces of a host batch that is polished in pipeline chunks (stream2 runs the event machine then)
	struct Tuning                     // ntedit_hip_set_tuning(): test / tuning knobs, none of which can change a result
	{
		u32 screen_mode = 0;     // overrides params.screen_mode when not 0
		u64 bin_chunk = 0;       // k-mer starts per record chunk of the binned screening (tests: several chunks)
		u32 bin_cap_percent = 0; // run capacity in percent of the expected records (tests: force the overflow list)
		u64 bin_ovf_cap = 0;     // entries of the overflow list (tests: a list that runs out; 0: an eighth of the chunk's records)
		u32 bin_fallback = 0;    // 1: the direct kernel whatever the sizes (as "screen_mode" 1; kept for callers of round 5)
		u32 force_xcc = 0;       // x + 1: every probe wavefront pretends to run on XCD x (tests)
		u32 bin_timing = 0;      // per-stage times of the binned screening on stderr
		u64 chunk_bytes = 0;     // pipeline chunk size (tests: many chunks)
		u64 h2d_piece = ~0ULL;   // bytes per host-to-device piece (~0: default)
		u32 inline_tries = ~0u;  // candidates of an indel sweep the deferring launch tries itself (~0: default)
		u32 assess = ~0u;        // k_assess before the event machine: 0 never, 1 always, ~0: with -s 1 and with counting filters
		u32 machine_cfg = ~0u;   // 0: always the general instantiation of the machine kernels (~0: the most specific one)
		u32 lanes = ~0u;         // DevParams::lanes (~0: default)
		u32 defer_run = ~0u;     // DevParams::defer_run (~0: default)
		u32 defer_fail = ~0u;    // DevParams::defer_fail (~0: default)
		u32 defer_fail_snv = ~0u; // DevParams::defer_fail_snv (~0: default)
		u32 h2d_fixed_schedule = 0; // a batch arriving in pieces: round 3's chunk schedule (1, 3, 8 pieces, the rest) instead of chunks by arrival
		u32 candmap = 0;         // the candidate map of -s 1 (nte_bin_wc.inc MODE 1): 1 = whenever the configuration allows it (measured slower: off)
		u32 snv_wave = 0;        // -s 1 with the run map: the events go to the wavefront-per-event launch (experiment)
		u32 no_rounds = 0, no_early_copy = 0;
		u32 force_rounds = 0;     // event rounds whatever the number of events (tests: small inputs)
		u32 bin_scatter = 0;      // partition kernel: 0 barrier-phased (k_wc_scatter_b), 1 barrier-free (k_wc_scatter)
		u32 probe_parts_log2 = ~0u; // probe stage: slices probed in 2^x parts (~0: by slice size)
	} tune;
	DevBuf ev_cover, ev_before, ev_flags, ev_list, ev_bmax; // event rounds
	u32 cu_count = 256;
	size_t lds_per_block = 160 * 1024;
	double alloc_ms = 0.0;            // host time spent in hipFree + hipMalloc of the grow-only buffers (ensure())
	u32 alloc_calls = 0;
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
	u32 part_margin = 0; // k + max deletions + slack of those parameters (RenderOptions::part_margin)
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
	const auto t0 = std::chrono::steady_clock::now();
	if (b.p) {
		HIP_TRY(c, hipFree(b.p));
		b.p = nullptr;
		b.cap = 0;
	}
	size_t want = bytes + bytes / 8 + 256;
	HIP_TRY(c, hipMalloc(&b.p, want));
	b.cap = want;
	const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
	c->alloc_ms += ms;
	c->alloc_calls++;
	if (ms > 1.0 && getenv("NTEDIT_HIP_DEBUG")) {
		fprintf(stderr, "[ntedit_hip] device buffer of %.1f MB: %.2f ms of hipFree + hipMalloc\n", want / 1e6, ms);
	}
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
	if (c->tune.defer_fail != ~0u) {
		c->dp.defer_fail = c->tune.defer_fail;
	}
	if (c->tune.defer_fail_snv != ~0u) {
		c->dp.defer_fail_snv = c->tune.defer_fail_snv;
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

#include "nte_api_screen.inc"

} // namespace

#include "nte_api_filters.inc"
#include "nte_api_polish.inc"
#include "nte_api_results.inc"
