// nte_machine.h -- the per-event edit-search state machine.
//
// One event = one "absent" k-mer start found by the screening kernel.  The
// machine restates, for a thread that starts in a CLEAN state (no earlier edit
// inside its k-mer window), exactly what the reference's strictly serial
// per-contig loop does from that position on (ntedit.cpp:1798-2139): confirm
// the k-mer is missing on a k/j subset (step 2), sweep substitution candidates
// (step 3), sweep insertion / deletion candidates (steps 4-5, tryIndels /
// tryDeletion ntedit.cpp:1451-1744), apply the best edit to the edited-sequence
// rope (makeEdit 1250-1448 with makeInsertion 625-714 / makeDeletion 719-809),
// and keep rolling while the window still overlaps an edit.  It stops as soon
// as it is clean again and the next position belongs to another event (or is
// present in the filter), reporting how far it got ("cover_end"); the host
// drops every speculative event that starts below an earlier event's
// cover_end, which reproduces the reference's serial order exactly.
//
// Design notes (MI355X): one thread per event; the rope is a small sliding
// window of nodes in a per-thread global workspace (finalised nodes are
// streamed out), modified draft characters live in a tiny overlay, all output
// goes through 16-byte items appended to 128-byte chunks of a global arena.
// The same source compiles for the host so the control logic can be tested
// without a GPU (tests/hostsim) -- that build is test-only and is not part of
// the shipped library.
#pragma once
#include "nte_common.h"
#include "nte_lanes.h"

namespace nte {

// optional work counters for the host-side profile of the control logic (tests/hostsim)
#if defined(NTE_COUNTERS)
struct WorkCounters
{
	unsigned long long probes, fast_rolls, slow_rolls, ins_cands, del_cands, sweeps, windows_fast, windows_slow, windows_fail;
	unsigned long long lane_batches, lane_positions, lane_walked, lane_edits;
};
extern WorkCounters g_wc;
#define NTE_COUNT(f, n) (g_wc.f += (n))
#else
#define NTE_COUNT(f, n) ((void)0)
#endif

// Phase timers of the wavefront-per-event kernel (build nte_machine_wave.hip with
// -DNTE_PROFILE): shader cycles per phase, summed over events into g_prof[].
#if defined(NTE_PROFILE)
static __device__ unsigned long long g_prof[64]; // (one per translation unit: the kernels are compiled per configuration)
// events that took more than 2^22 ticks, for the question "which events is a launch waiting for": [0] = entries written,
// then {start position in the batch, begin tick, end tick, positions the run covered} per event (k_machine's wrapper)
constexpr unsigned NTE_EVLOG_CAP = 8192;
static __device__ unsigned long long g_evlog[4 + 4 * NTE_EVLOG_CAP];
// filter bytes gathered by the events that start in each 4 Mbase stretch of the batch (k_machine's wrapper): where the work is
constexpr unsigned NTE_REGION_SHIFT = 22, NTE_REGIONS = 4096;
static __device__ unsigned long long g_region[NTE_REGIONS];
#endif
#if defined(NTE_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
#define NTE_PROF_DECL unsigned long long prof_t = __builtin_amdgcn_s_memtime(), prof_t0 = prof_t, prof_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }
#define NTE_PROF_COUNT(slot) (prof_cnt[slot]++)
#define NTE_PROF_ADD(slot, n) (prof_cnt[slot] += (n))
#define NTE_GATHER(n) (prof_gathers += (n)) // filter bytes gathered by this lane (the edit search's probes)
#define NTE_PROF(slot)                                                    \
	do {                                                                  \
		const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
		prof_acc[slot] += now_ - prof_t;                                  \
		prof_t = now_;                                                    \
	} while (0)
#define NTE_PROF_SUB(slot)                                                \
	do {                                                                  \
		const unsigned long long now_ = __builtin_amdgcn_s_memtime();     \
		prof_sub[slot] += now_ - prof_sub_t;                              \
		prof_sub_t = now_;                                                \
	} while (0)
#define NTE_PROF_FLUSH                                                    \
	do {                                                                  \
		if (prof_gathers) {                                               \
			atomicAdd(&g_prof[15], prof_gathers); /* (every lane its own) */ \
		}                                                                 \
		if ((threadIdx.x & (e.wave_size - 1u)) == 0) {                    \
			for (int i_ = 0; i_ < 8; i_++) {                              \
				atomicAdd(&g_prof[i_], prof_acc[i_]);                     \
				atomicAdd(&g_prof[16 + i_], prof_sub[i_]);                \
			}                                                             \
			atomicAdd(&g_prof[8], 1ull);                                  \
			{                                                             \
				const unsigned long long dur_ = __builtin_amdgcn_s_memtime() - prof_t0; \
				atomicAdd(&g_prof[32 + (63 - __builtin_clzll(dur_ | 1ull)) % 32], 1ull); \
				atomicMax(&g_prof[13], dur_);                             \
				atomicMax(&g_prof[14], (unsigned long long)prof_cnt[0]);  \
			}                                                             \
			for (int i_ = 0; i_ < 4; i_++) {                              \
				atomicAdd(&g_prof[9 + i_], prof_cnt[i_]);                 \
			}                                                             \
		}                                                                 \
	} while (0)
#else
#define NTE_PROF_DECL ((void)0)
#define NTE_PROF_COUNT(slot) ((void)0)
#define NTE_PROF_ADD(slot, n) ((void)0)
#define NTE_GATHER(n) ((void)0)
#define NTE_PROF(slot) ((void)0)
#define NTE_PROF_SUB(slot) ((void)0)
#define NTE_PROF_FLUSH ((void)0)
#endif



struct EventEnv
{
	const u8* seq;  // contig bases
	const u8* batch_end; // one past the last byte of the batch buffer
	u32 len;
	u32 contig;
	u64 gbase;      // global index of seq[0] in the batch (bitmap coordinates)
	const u64* bitmap;  // 1 bit per k-mer start: all bases accepted and the k-mer absent (the screening pass)
	const u64* runmap;  // the positions the serial walk stops at / goes on from: the same bitmap, or a subset of it that
	                    // leaves out the positions whose assessment cannot do anything (k_assess, nte_assess.hip)
	const u64* tab; // seed tables (LDS on the device)
	const DevParams* p;
	Filter bloom, rep;
	// per-thread workspace
	Node* nodes;
	u32* ov_pos;
	u8* ov_chr;
	u8* win;       // character-code window of the failing position: k + (k + max_del + 2) bytes
	u32 win_stride; // distance between consecutive window bytes (1 = private slice, else interleaved)
	u8* prev;      // scratch for the previous-insertion string (node_window bytes)
	int16_t* lps;  // scratch for its KMP failure table (node_window entries)
	// output arena
	Item* arena;
	u32* arena_next;
	u32 arena_chunks;
	// pass 1 of the two-pass launch: stop (and emit nothing) at the first indel sweep
	bool defer_sweeps;
	// 1 = one thread per event; 64 = one wavefront per event: every lane runs the same
	// serial machine on the same state (uniform control flow, same-value stores) and the
	// lanes split the indel candidate sweep between them
	u32 wave_size;
};

NTE_HD bool
bit_absent(const u64* bitmap, u64 g)
{
	return (bitmap[g >> 6] >> (g & 63)) & 1;
}

// absent bits among positions g, g + jump, g + 2 jump, ... below g + k (step 2 in the clean state reads the screening's
// answers, ntedit.cpp:1826-1858): the words that hold them are loaded first, all at once, and the bits are counted from
// registers -- k / jump loads of single bits, one after the other in a loop of unknown length, were nine round trips to
// the L2 at k = 25 (round 6).  k <= 200: at most five words.
NTE_HD u32
absent_count_stride(const u64* bitmap, u64 g, u32 k, u32 jump)
{
	const u64 w0 = g >> 6;
	const u32 gb = (u32)(g & 63);
	const u32 nw = (gb + k + 63) >> 6;
	u64 w[5];
	NTE_UNROLL
	for (int i = 0; i < 5; i++) {
		w[i] = (u32)i < nw ? bitmap[w0 + i] : 0;
	}
	u32 n = 0, q = 0;
	NTE_UNROLL
	for (int i = 0; i < 5; i++) {
		const u32 lim = 64u * (u32)(i + 1) - gb; // positions q < lim lie in word i
		while (q < k && q < lim) {
			n += (u32)((w[i] >> ((gb + q) & 63)) & 1);
			q += jump;
		}
	}
	return n;
}

// event-start predicate shared by the extraction kernel and the machine
NTE_HD bool
is_event_start(const u64* bitmap, u64 g, u32 grid)
{
	if (!bit_absent(bitmap, g)) {
		return false;
	}
	if (g == 0 || !bit_absent(bitmap, g - 1)) {
		return true;
	}
	return (g % grid) == 0;
}

struct Best
{
	u32 edit_type; // 0 none, 1 substitution, 2 insertion, 3 deletion
	u8 indel[INDEL_BYTES];
	u32 n_indel;
	u8 sub_base;
	u32 num_support;
	u8 altbase1, altbase2, altbase3;
	u32 altsupp1, altsupp2, altsupp3;
};

// CFG: what the launch knows about its configuration at compile time.  The machine kernels are hundreds of KB of code
// against a 64 KB instruction cache, and a wavefront walks through most of it once per event: every path a configuration
// cannot take (counting filters, the secondary filter, -s 1, modes 1 / 2, -a, filters that are not a power of two) is
// code it still drags through the cache.  CFG = 0 is the general machine.
enum MachineCfg : u32
{
	CFG_MODE0 = 1, // -m 0 and no -a
	CFG_PLAIN = 2, // plain Bloom filter(s), no -s 1
	CFG_NOSEC = 4, // no secondary filter
	CFG_POW2 = 8   // every filter's size is a power of two
};

template<u32 CFG>
struct MachineT
{
	const EventEnv& e;
	const DevParams& p;

	NTE_HD u32 mode() const { return (CFG & CFG_MODE0) ? 0u : p.mode; }
	NTE_HD bool mask() const { return (CFG & CFG_MODE0) ? false : p.mask != 0; }
	NTE_HD bool snv() const { return (CFG & CFG_PLAIN) ? false : p.snv != 0; }
	NTE_HD bool secbf() const { return (CFG & CFG_NOSEC) ? false : p.secbf != 0; }
	NTE_HD bool counting() const { return (CFG & CFG_PLAIN) ? false : e.bloom.counting != 0; }
	NTE_HD bool fcounting(const Filter& f) const { return (CFG & CFG_PLAIN) ? false : f.counting != 0; }
	NTE_HD u64 slot(const Filter& f, u64 hv) const { return (CFG & CFG_POW2) ? (hv & f.mask) : filter_slot(f, hv); }

	// btllib contains() / the counting filter's minimum (nte_common.h) with the configuration folded in
	NTE_HD u32
	min_count(const Filter& f, u64 base) const
	{
		u32 mn = 255;
		for (unsigned i = 0; i < f.hash_num; i++) {
			const u32 c = f.data[slot(f, hash_extend(base, p, i))];
			NTE_GATHER(1);
			mn = c < mn ? c : mn;
			if (mn == 0) {
				break;
			}
		}
		return mn;
	}

	NTE_HD bool
	contains(const Filter& f, const HashState& s) const
	{
		const u64 base = s.fh + s.rh;
		if (fcounting(f)) {
			return min_count(f, base) > 0;
		}
		for (unsigned i = 0; i < f.hash_num; i++) {
			const u64 n = slot(f, hash_extend(base, p, i));
			NTE_GATHER(1);
			if (!((f.data[n >> 3] >> (n & 7)) & 1)) {
				return false;
			}
		}
		return true;
	}

	// cursors (ntedit.cpp:1773-1795)
	u32 h_seq_i, t_seq_i, h_node, t_node;
	HashState hs;
	// rope window
	u32 nbase, nsize;
	bool rope_touched;
	// overlay of modified draft characters
	u32 n_ov;
	bool tmp_on;
	u32 tmp_pos;
	u8 tmp_chr;
	int64_t last_sub_pos; // right-most substituted draft position so far
	// output
	u32 first_chunk, cur_chunk, fill;
	u32 flags;
	// window of the failing position (see fill_window)
	bool win_ok;
	bool wc_valid;      // the window holds a clean stretch of draft codes starting at wc_pos0
	u32 wc_pos0, wc_len;
	u32 win_off;        // offset of the current failing position inside that stretch
	u32 cand_l1 = ~0u;  // k_assess with the candidate map of -s 1: first-probe bits of this position's candidates (~0: not known)
	u32 there_known = ~0u; // k_assess, second phase: the candidates whose own k-mer is there, found by the first phase (~0: ask)
	// presence of the next k-mers while the window still overlaps an edit (see build_lookahead)
	u32 la_mask, la_n, la_i;
	bool la_off;
	// what the support count of an applied substitution already knows about the k-mers behind it (see process_missing):
	// valid for a look-ahead built with the head cursor at la_known_pos
	u32 la_known, la_known_vals, la_known_pos;
	bool changed_seq; // the last failing position applied an edit (else the sequence, and the look-ahead, still stand)
	bool la_win;      // the character window is still the one the look-ahead was hashed from (the stride needs it)
#if defined(NTE_PROFILE) && defined(__HIP_DEVICE_COMPILE__)
	mutable unsigned long long prof_sub_t = 0, prof_sub[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, prof_cnt[4] = { 0, 0, 0, 0 }, prof_gathers = 0;
#endif

	NTE_HD
	MachineT(const EventEnv& env)
	  : e(env)
	  , p(*env.p)
	{
	}

#include "nte_machine_rope.inc"
#include "nte_machine_filters.inc"
#include "nte_machine_sweeps.inc"
#include "nte_machine_position.inc"
#include "nte_machine_lanes.inc"
#include "nte_machine_run.inc"
};

typedef MachineT<0> Machine; // the general machine

// the most specific machine a configuration allows (host side: which kernel / host instantiation to use)
inline u32
machine_cfg_of(const DevParams& p, const Filter& bloom, const Filter& rep)
{
	u32 cfg = 0;
	if (p.mode == 0 && !p.mask) {
		cfg |= CFG_MODE0;
	}
	if (!bloom.counting && !(p.secbf && rep.counting) && !p.snv) {
		cfg |= CFG_PLAIN;
	}
	if (!p.secbf) {
		cfg |= CFG_NOSEC;
	}
	if (bloom.mask && (!p.secbf || rep.mask)) {
		cfg |= CFG_POW2;
	}
	return cfg;
}

} // namespace nte
