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


#if defined(__HIP_DEVICE_COMPILE__)
#define NTE_ATOMIC_INC(p) atomicAdd((p), 1u)
#else
#define NTE_ATOMIC_INC(p) ((*(p))++)
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
	u8 indel[12];
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

	// ------------------------------------------------------- wave helpers
	NTE_HD u32
	wave_lane() const
	{
#if defined(__HIP_DEVICE_COMPILE__)
		return threadIdx.x & (e.wave_size - 1u);
#else
		return 0;
#endif
	}

	NTE_HD u64
	wave_ballot(bool pred) const
	{
#if defined(__HIP_DEVICE_COMPILE__)
		if (e.wave_size > 1) {
			// the lanes of this event's group (a wavefront may carry 64 / wave_size events)
			const u64 all = __ballot(pred);
			const u32 shift = threadIdx.x & 63u & ~(e.wave_size - 1u);
			return e.wave_size >= 64 ? all : ((all >> shift) & ((1ull << e.wave_size) - 1ull));
		}
#endif
		return pred ? 1ull : 0ull;
	}

	NTE_HD u32
	wave_shfl(u32 v, u32 src) const
	{
#if defined(__HIP_DEVICE_COMPILE__)
		if (e.wave_size > 1) {
			return (u32)__shfl((int)v, (int)src, (int)e.wave_size);
		}
#endif
		(void)src;
		return v;
	}

	// one arena chunk for the whole wave (lane 0 allocates, everybody learns the index)
	NTE_HD u32
	alloc_chunk()
	{
#if defined(__HIP_DEVICE_COMPILE__)
		if (e.wave_size > 1) {
			u32 c = 0;
			if ((threadIdx.x & (e.wave_size - 1u)) == 0) {
				c = atomicAdd(e.arena_next, 1u);
			}
			return (u32)__shfl((int)c, 0, (int)e.wave_size);
		}
#endif
		return NTE_ATOMIC_INC(e.arena_next);
	}

	// ------------------------------------------------------------ output
	NTE_HD void
	emit(const Item& it)
	{
		if (flags & (EV_OVERFLOW | EV_ARENA_FULL)) {
			return;
		}
		if (cur_chunk == NONE32 || fill == CHUNK_ITEMS) {
			u32 c = alloc_chunk();
			if (c >= e.arena_chunks) {
				flags |= EV_ARENA_FULL;
				return;
			}
			if (cur_chunk == NONE32) {
				first_chunk = c;
				fill = 2; // link + event header
			} else {
				Item link;
				link.w[0] = c;
				link.w[1] = fill;
				link.w[2] = link.w[3] = 0;
				e.arena[(u64)cur_chunk * CHUNK_ITEMS] = link;
				fill = 1;
			}
			cur_chunk = c;
		}
		e.arena[(u64)cur_chunk * CHUNK_ITEMS + fill] = it;
		fill++;
	}

	NTE_HD void
	emit_node(const Node& n)
	{
		Item it;
		it.w[0] = TAG_NODE | ((u32)(u8)n.type << 8) | ((u32)n.c << 16);
		it.w[1] = n.s_pos;
		it.w[2] = n.e_pos;
		it.w[3] = n.support;
		emit(it);
	}

	NTE_HD void
	emit_mod(u32 pos, u8 c)
	{
		Item it;
		it.w[0] = TAG_MOD | ((u32)c << 8);
		it.w[1] = pos;
		it.w[2] = it.w[3] = 0;
		emit(it);
	}

	// ------------------------------------------------------- rope window
	NTE_HD Node
	unset_node() const
	{
		Node n;
		n.s_pos = n.e_pos = 0;
		n.support = 0;
		n.type = -1;
		n.c = 0;
		return n;
	}

	NTE_HD Node
	nget(u32 idx) const
	{
		if (idx >= nsize || idx < nbase) {
			return unset_node(); // (U1) reads past the end give an unset node
		}
		return e.nodes[idx - nbase];
	}

	NTE_HD void
	nset(u32 idx, const Node& n)
	{
		if (idx < nbase) {
			flags |= EV_OVERFLOW; // would touch an already streamed-out node
			return;
		}
		e.nodes[idx - nbase] = n;
	}

	NTE_HD void
	nput(u32 idx, const Node& n)
	{
		// "assign if idx < size else push_back" idiom of the reference
		if (idx < nsize) {
			nset(idx, n);
		} else {
			if (nsize - nbase >= p.node_window) {
				flags |= EV_OVERFLOW;
				return;
			}
			e.nodes[nsize - nbase] = n;
			nsize++;
		}
	}

	NTE_HD void
	set_type(u32 idx, int8_t t)
	{
		if (idx >= nbase && idx < nsize) {
			e.nodes[idx - nbase].type = t;
		}
	}

	// stream out rope nodes that can no longer be touched
	NTE_HD void
	housekeeping()
	{
		if (nsize - nbase + 40 > p.node_window) {
			u32 keep = h_node < t_node ? h_node : t_node;
			while (keep > nbase && nget(keep - 1).type == 1) {
				keep--;
			}
			if (keep > nbase) {
				keep--; // the position node in front of the run (see drop_prev_insertion)
			}
			if (keep > nbase) {
				for (u32 i = nbase; i < keep; i++) {
					emit_node(e.nodes[i - nbase]);
				}
				u32 n = nsize - keep;
				for (u32 i = 0; i < n; i++) {
					e.nodes[i] = e.nodes[i + (keep - nbase)];
				}
				nbase = keep;
			}
			if (nsize - nbase + 24 > p.node_window) {
				flags |= EV_OVERFLOW;
			}
		}
		// (every read of a draft character scans the overlay: entries behind the head cursor go as soon as there are a few)
		if (n_ov >= 48 || n_ov + 8 > p.node_window) {
			u32 w = 0;
			for (u32 i = 0; i < n_ov; i++) {
				if (e.ov_pos[i] >= h_seq_i) {
					e.ov_pos[w] = e.ov_pos[i];
					e.ov_chr[w] = e.ov_chr[i];
					w++;
				}
			}
			n_ov = w;
			if (n_ov + 8 > p.node_window) {
				flags |= EV_OVERFLOW;
			}
		}
	}

	// -------------------------------------------------- draft characters
	NTE_HD u8
	seq_at(u32 pos) const
	{
		if (pos >= e.len) {
			return 0; // contigSeq.at() would throw; unreachable in practice
		}
		if (tmp_on && pos == tmp_pos) {
			return tmp_chr;
		}
		for (u32 i = n_ov; i > 0; i--) {
			if (e.ov_pos[i - 1] == pos) {
				return e.ov_chr[i - 1];
			}
		}
		return e.seq[pos];
	}

	NTE_HD void
	set_seq(u32 pos, u8 c)
	{
		for (u32 i = 0; i < n_ov; i++) {
			if (e.ov_pos[i] == pos) {
				if (e.ov_chr[i] != c) {
					e.ov_chr[i] = c;
					emit_mod(pos, c);
				}
				return;
			}
		}
		if (e.seq[pos] == c) {
			return;
		}
		if (n_ov < p.node_window) {
			e.ov_pos[n_ov] = pos;
			e.ov_chr[n_ov] = c;
			n_ov++;
		} else {
			flags |= EV_OVERFLOW;
		}
		emit_mod(pos, c);
	}

	// ntedit.cpp:812-823
	NTE_HD u8
	get_character(u32 pos, const Node& n) const
	{
		if (n.type == 0) {
			return seq_at(pos);
		}
		if (n.type == 1) {
			return n.c;
		}
		return 0;
	}

	// ntedit.cpp:826-844
	NTE_HD void
	increment(u32& pos, u32& node_index) const
	{
		Node n = nget(node_index);
		if (n.type == 0) {
			pos++;
			if (pos > n.e_pos) {
				node_index++;
				Node nx = nget(node_index);
				if (nx.type == 0) {
					pos = nx.s_pos;
				}
			}
		} else if (n.type == 1) {
			node_index++;
			Node nx = nget(node_index);
			if (nx.type == 0) {
				pos = nx.s_pos;
			}
		}
	}

	// increment() with the node under the cursor kept by the caller (n == nget(node_index) on entry and on return)
	NTE_HD void
	increment_cached(u32& pos, u32& node_index, Node& n) const
	{
		if (n.type == 0) {
			pos++;
			if (pos > n.e_pos) {
				node_index++;
				n = nget(node_index);
				if (n.type == 0) {
					pos = n.s_pos;
				}
			}
		} else if (n.type == 1) {
			node_index++;
			n = nget(node_index);
			if (n.type == 0) {
				pos = n.s_pos;
			}
		}
	}

	// ntedit.cpp:1216-1247
	NTE_HD bool
	roll(u32& hs_i, u32& ts_i, u32& hn, u32& tn, u8& char_out, u8& char_in) const
	{
		if (hs_i >= e.len || hn >= nsize) {
			return false;
		}
		char_out = get_character(hs_i, nget(hn));
		increment(hs_i, hn);
		if (ts_i >= e.len || tn >= nsize) {
			return false;
		}
		increment(ts_i, tn);
		if (ts_i >= e.len || tn >= nsize) {
			return false;
		}
		char_in = get_character(ts_i, nget(tn));
		return true;
	}

	// --------------------------------------------------------- filters
	NTE_HD bool
	in_bloom(const HashState& s) const
	{
		NTE_COUNT(probes, 1);
		return contains(e.bloom, s);
	}

	// BFWrapper::get_count (ntedit.cpp:373-376): min counter, or 1 for a plain filter
	NTE_HD u32
	count_of(const HashState& s) const
	{
		return counting() ? min_count(e.bloom, s.fh + s.rh) : 1u;
	}

	// the main loop's test (ntedit.cpp:1806): not contained, or (counting) seen fewer than -p times
	NTE_HD bool
	screen_absent(const HashState& s) const
	{
		NTE_COUNT(probes, 1);
		if (counting()) {
			const u32 c = min_count(e.bloom, s.fh + s.rh);
			return c == 0 || c < p.min_thr;
		}
		return !contains(e.bloom, s);
	}

	// is_kmer_solid (ntedit.cpp:465-473)
	NTE_HD bool
	solid(const HashState& s) const
	{
		if (secbf() && contains(e.rep, s)) {
			return false;
		}
		if (counting()) {
			const u32 c = min_count(e.bloom, s.fh + s.rh);
			return c <= p.max_thr && c >= p.min_thr;
		}
		return true;
	}

	NTE_HD bool
	present_solid(const HashState& s) const
	{
		return in_bloom(s) && solid(s);
	}

	// Membership of up to G k-mers at once.  The h probes of ONE k-mer depend on each other
	// only through the early exit, which has no side effect; probing level by level keeps
	// G independent gathers in flight instead of one (the event machine is bound by memory
	// latency, not bandwidth).  All array indices are compile-time constants so the group
	// lives in registers.  b[i] = canonical hash fh+rh; returns a bit mask of members.
	template<int G>
	NTE_HD u32
	probe_group(const Filter& f, const u64 (&b)[G], u32 n) const
	{
		return probe_group_range<G>(f, b, n, 1, 255);
	}

	// counting filters: members are the k-mers whose min counter lies in [lo, hi] (lo >= 1);
	// plain filters ignore the range
	template<int G>
	NTE_HD u32
	probe_group_range(const Filter& f, const u64 (&b)[G], u32 n, u32 lo, u32 hi, u32 only = 0xFFFFFFFFu) const
	{
		u32 alive = ((1u << n) - 1) & only;
		NTE_COUNT(probes, n);
		if (fcounting(f)) {
			u8 mn[G];
			NTE_UNROLL
			for (int i = 0; i < G; i++) {
				mn[i] = 255;
			}
			for (u32 h = 0; h < f.hash_num && alive; h++) {
				u8 byte[G];
				NTE_UNROLL
				for (int i = 0; i < G; i++) {
					byte[i] = 255;
					if ((alive >> i) & 1) {
						byte[i] = f.data[slot(f, hash_extend(b[i], p, h))];
						NTE_GATHER(1);
					}
				}
				NTE_UNROLL
				for (int i = 0; i < G; i++) {
					mn[i] = byte[i] < mn[i] ? byte[i] : mn[i];
					if (mn[i] == 0) {
						alive &= ~(1u << i);
					}
				}
			}
			u32 m = 0;
			NTE_UNROLL
			for (int i = 0; i < G; i++) {
				if (((alive >> i) & 1) && mn[i] >= lo && mn[i] <= hi) {
					m |= 1u << i;
				}
			}
			return m;
		}
		for (u32 h = 0; h < f.hash_num && alive; h++) {
			u8 byte[G];
			u8 sh[G];
			NTE_UNROLL
			for (int i = 0; i < G; i++) {
				byte[i] = 0xFF;
				sh[i] = 0;
				if ((alive >> i) & 1) {
					const u64 sl = slot(f, hash_extend(b[i], p, h));
					NTE_GATHER(1);
					byte[i] = f.data[sl >> 3];
					sh[i] = (u8)(sl & 7);
				}
			}
			NTE_UNROLL
			for (int i = 0; i < G; i++) {
				if (!((byte[i] >> sh[i]) & 1)) {
					alive &= ~(1u << i);
				}
			}
		}
		return alive;
	}

	// present in the primary filter and (solid_check) not in the secondary one
	template<int G>
	NTE_HD u32
	present_group(const u64 (&b)[G], u32 n, bool solid_check) const
	{
		// contains() alone, or contains() && is_kmer_solid() (counter within [-p, -q] and not
		// in the secondary filter)
		u32 lo = 1, hi = 255;
		if (solid_check && counting()) {
			lo = p.min_thr > 1 ? p.min_thr : 1;
			hi = p.max_thr;
		}
		u32 m = probe_group_range<G>(e.bloom, b, n, lo, hi);
		if (solid_check && secbf() && m) {
			// the secondary filter only matters for k-mers that are present; members of it are not solid
			const u32 in_rep = probe_group_range<G>(e.rep, b, n, 1, 255, m);
			m &= ~in_rep;
		}
		return m;
	}

	NTE_HD static u32
	popc32(u32 x)
	{
		u32 c = 0;
		while (x) {
			x &= x - 1;
			c++;
		}
		return c;
	}

	// Walks rolls kk0..last (rf performs roll number kk on the hash state and returns false to
	// abort), gathers the subset k-mers (kk % jump == 0) in groups of G and probes each group
	// at once.  extra: a k-mer counted ahead of the walk (the changed k-mer of a deletion).
	// need_present / need_absent (0 = off): give up as soon as the count can no longer reach it.
	struct SubsetResult
	{
		u32 present, total;
		bool aborted, gave_up;
		u64 pmask; // bit n: the n-th k-mer of the subset (in walk order; the first 64) is there
	};

	// SPLIT: the first group holds only as many k-mers as it takes to know that need_present is out of reach
	// (subset size - need_present + 1, all of them absent).  The candidates of an indel sweep are wrong but for one, and
	// the sweep is bound by the number of gathers: 6 instead of 8 per wrong insertion at k=25, jump=3, -y 9.
	template<int G, bool SPLIT = false, typename RollFn>
	NTE_HD SubsetResult
	subset_scan(HashState ts, u32 kk0, u32 last, bool solid_check, bool have_extra, u64 extra, u32 need_present, u32 need_absent, RollFn rf) const
	{
		SubsetResult r;
		r.present = 0;
		r.total = 0;
		r.pmask = 0;
		r.aborted = false;
		r.gave_up = false;
		u32 kk = kk0;
		u32 cap = G;
		if (SPLIT && need_present && !need_absent) {
			u32 n_sub = have_extra ? 1u : 0u;
			if (kk0 <= last) {
				n_sub += last / p.jump - (kk0 ? (kk0 - 1) / p.jump : 0) + (kk0 == 0 ? 1u : 0u);
			}
			if (n_sub >= need_present && n_sub - need_present + 1 < (u32)G) {
				cap = n_sub - need_present + 1;
			}
		}
		while (true) {
			u64 b[G];
			u32 nb = 0;
			NTE_UNROLL
			for (int u = 0; u < G; u++) {
				b[u] = 0;
			}
			if (have_extra) {
				b[0] = extra;
				nb = 1;
				have_extra = false;
			}
			NTE_UNROLL
			for (int u = 0; u < G; u++) {
				if (nb == (u32)u && (u32)u < cap) {
					while (kk <= last) {
						if (!rf(kk, ts)) {
							r.aborted = true;
							return r;
						}
						const bool is_sub = (kk % p.jump) == 0;
						kk++;
						if (is_sub) {
							b[u] = ts.fh + ts.rh;
							nb = (u32)u + 1;
							break;
						}
					}
				}
			}
			if (nb == 0) {
				break;
			}
			{
				const u32 pm = present_group<G>(b, nb, solid_check);
				r.present += popc32(pm);
				if (r.total < 64) {
					r.pmask |= (u64)pm << r.total;
				}
			}
			r.total += nb;
			cap = G;
			if (kk > last) {
				break;
			}
			const u32 left = probes_left(kk - 1, last);
			if ((need_present && r.present + left < need_present) ||
			    (need_absent && (r.total - r.present) + left < need_absent)) {
				r.gave_up = true;
				break;
			}
		}
		return r;
	}

	NTE_HD void
	roll_hash(HashState& s, u8 char_out, u8 char_in) const
	{
		NTE_COUNT(slow_rolls, 1);
		const u8 co = char_code(char_out), ci = char_code(char_in);
		if ((co == CODE_BAD && is_exotic(char_out)) || (ci == CODE_BAD && is_exotic(char_in))) {
			hash_roll_raw(s, p.k, char_out, char_in); // rare: U, '-', '*', ... inside a hashed k-mer
		} else {
			hash_roll(s, e.tab, co, ci);
		}
	}

	NTE_HD void
	changelast(HashState& s, u8 char_out, u8 char_in) const
	{
		const u8 co = char_code(char_out), ci = char_code(char_in);
		if ((co == CODE_BAD && is_exotic(char_out)) || (ci == CODE_BAD && is_exotic(char_in))) {
			hash_changelast_raw(s, p.k, char_out, char_in);
		} else {
			hash_changelast(s, e.tab, co, ci);
		}
	}

	// ------------------------------------------------ rope edit primitives
	// ntedit.cpp:625-714
	NTE_HD void
	make_insertion(u32& tn, u32 insert_pos, const u8* ins, u32 n_ins, u32 support)
	{
		rope_touched = true;
		Node orig = nget(tn);
		Node cn;
		cn.s_pos = cn.e_pos = 0;
		cn.type = 1;
		cn.support = (u16)support;
		if ((orig.type == 0 && insert_pos <= orig.s_pos) || orig.type == 1) {
			// shift the run of valid nodes starting at tn right by n_ins
			u32 end = tn;
			while (end < nsize && nget(end).type != -1) {
				end++;
			}
			u32 n_re = end - tn;
			while (nsize < tn + n_ins + n_re && !(flags & EV_OVERFLOW)) {
				nput(nsize, unset_node()); // grow first, then shift in place
			}
			for (u32 q = n_re; q > 0; q--) {
				nset(tn + n_ins + q - 1, nget(tn + q - 1));
			}
			for (u32 q = 0; q < n_ins; q++) {
				cn.c = ins[q];
				nput(tn + q, cn);
			}
		} else if (orig.type == 0) {
			Node after;
			after.type = 0;
			after.s_pos = insert_pos;
			after.e_pos = orig.e_pos;
			after.c = 0;
			after.support = 0;
			orig.e_pos = insert_pos - 1;
			nset(tn, orig);
			for (u32 q = 0; q < n_ins; q++) {
				cn.c = ins[q];
				nput(tn + q + 1, cn);
			}
			nput(tn + n_ins + 1, after);
			tn++;
		}
	}

	// ntedit.cpp:719-809 (the reference recurses on the leftover; here a loop)
	NTE_HD void
	make_deletion(u32& tn, u32& pos, u32 num_del, u32 support)
	{
		rope_touched = true;
		while (true) {
			Node orig = nget(tn);
			u32 leftover = 0;
			if (orig.type == 0) {
				if (pos <= orig.s_pos) {
					if ((u64)pos + num_del <= orig.e_pos) {
						// deleting off the beginning of a position node
						orig.s_pos = pos + num_del;
						orig.support = (u16)support;
						nset(tn, orig);
						pos = orig.s_pos;
						return;
					}
					// the whole position node goes; later nodes move down one slot
					leftover = (u32)((u64)pos + num_del - orig.e_pos);
					pos = orig.e_pos + 1;
					u32 i = tn + 1;
					while (i < nsize && nget(i).type != -1) {
						nset(i - 1, nget(i));
						set_type(i, -1);
						i++;
					}
				} else {
					if ((u64)pos + num_del <= orig.e_pos) {
						// deleting in the middle of a position node: split it
						Node split;
						split.type = 0;
						split.s_pos = pos + num_del;
						split.e_pos = orig.e_pos;
						split.c = 0;
						split.support = (u16)support;
						Node front = orig;
						front.e_pos = pos - 1;
						nset(tn, front);
						pos = split.s_pos;
						tn++;
						nput(tn, split);
						return;
					}
					// from the middle of a position node past its end
					leftover = (u32)((u64)pos + num_del - orig.e_pos);
					Node front = orig;
					front.e_pos = pos - 1;
					nset(tn, front);
					pos = orig.e_pos + 1;
					tn++;
				}
			} else if (orig.type == 1) {
				u32 i = tn;
				leftover = num_del;
				while (i < nsize && nget(i).type == 1 && leftover > 0) {
					set_type(i, -1);
					leftover--;
					i++;
				}
				u32 j = tn;
				while (i < nsize && nget(i).type != -1) {
					nset(j, nget(i));
					set_type(i, -1);
					i++;
					j++;
				}
			} else {
				return;
			}
			if (leftover == 0) {
				return;
			}
			Node nx = nget(tn);
			if (!(tn < nsize && nx.type != -1)) {
				return;
			}
			if (nx.type == 0) {
				pos = nx.s_pos;
			}
			num_del = leftover; // pass the rest of the deletion to the next node
		}
	}

	// ntedit.cpp:848-903.  Instead of returning the k-mer string the hash of it
	// is accumulated directly (seed of ntedit.cpp:412-413).  Returns false (and
	// h = t = len, hash of an all-zero-seed string) when no k-mer is left.
	NTE_HD bool
	find_accepted_kmer()
	{
		u32 temp_t_node = t_node;
		Node curr = nget(t_node);
		u32 i = t_seq_i;
		while (i < e.len && temp_t_node < nsize && nget(temp_t_node).type != -1) {
			u8 c = get_character(i, curr);
			if (char_code(c) != CODE_BAD) {
				u32 n = 1;
				u8 code = char_code(c);
				u64 fh = tab_f(e.tab, code);
				u64 rh = tab_r(e.tab, code);
				u32 temp_h_node = temp_t_node;
				u32 j = i;
				increment(j, temp_t_node);
				while (j < e.len && temp_t_node < nsize && nget(temp_t_node).type != -1) {
					curr = nget(temp_t_node);
					c = get_character(j, curr);
					code = char_code(c);
					if (code == CODE_BAD) {
						i = j;
						break;
					}
					fh = srol1(fh) ^ tab_f(e.tab, code);
					rh ^= sroln(tab_r(e.tab, code), n);
					n++;
					if (n == p.k) {
						break;
					}
					increment(j, temp_t_node);
				}
				if (n == p.k) {
					h_seq_i = i;
					t_seq_i = j;
					h_node = temp_h_node;
					t_node = temp_t_node;
					hs.fh = fh;
					hs.rh = rh;
					return true;
				}
			}
			increment(i, temp_t_node);
		}
		h_seq_i = e.len;
		t_seq_i = e.len;
		hs.fh = 0;
		hs.rh = 0;
		return false;
	}

	// ntedit.cpp:501-520
	NTE_HD static u8
	rc_char(u8 c)
	{
		switch (c) {
		case 'A':
		case 'a':
			return 'T';
		case 'T':
		case 't':
			return 'A';
		case 'G':
		case 'g':
			return 'C';
		case 'C':
		case 'c':
			return 'G';
		default:
			return 'N';
		}
	}

	// ntedit.cpp:907-922
	NTE_HD u32
	get_prev_insertion(u8* out) const
	{
		u32 n = 0;
		u32 idx = t_node;
		Node tn = nget(idx);
		if ((idx < nsize && tn.type == 0 && t_seq_i == tn.s_pos) || tn.type == 1) {
			idx--;
		}
		while (idx < nsize && idx >= nbase && nget(idx).type == 1) {
			if (n + 16 < p.node_window) {
				out[n] = rc_char(nget(idx).c);
			}
			n++;
			idx--;
		}
		return n;
	}

	// ntedit.cpp:561-596: is s a whole-number repetition of a shorter word?
	NTE_HD static bool
	is_repeat(const u8* s, int n, int16_t* lps)
	{
		if (n <= 0) {
			return false;
		}
		int len = 0, i = 1;
		lps[0] = 0;
		while (i < n) {
			if (s[i] == s[len]) {
				len++;
				lps[i] = (int16_t)len;
				i++;
			} else if (len != 0) {
				len = lps[len - 1];
			} else {
				lps[i] = 0;
				i++;
			}
		}
		len = lps[n - 1];
		return len > 0 && n % (n - len) == 0;
	}

	// ntedit.cpp:1321-1334 / 1352-1366: pull the nodes behind the tail over
	// the previous run of inserted characters
	NTE_HD void
	drop_prev_insertion(u32 count)
	{
		rope_touched = true;
		u32 j = 1;
		Node tn = nget(t_node);
		if (tn.type == 0 && t_seq_i == tn.s_pos) {
			j = 0;
		}
		for (u32 i = count; i > 0; i--) {
			u32 dst = t_node - i; // wraps like the reference when i > t_node
			Node src = nget(t_node + j);
			if (t_node + j < nsize && src.type != -1) {
				if (dst < nsize) {
					nset(dst, src);
				}
				set_type(t_node + j, -1);
				j++;
			} else if (dst < nsize) {
				if (dst < nbase) {
					flags |= EV_OVERFLOW;
				}
				set_type(dst, -1);
			}
		}
	}

	// i-th insertion candidate behind an index base: the index base followed by
	// every word over A<C<G<T of length 0..4 in length-then-lexicographic order
	// (the enumeration of ntedit.cpp:203-348, generated instead of stored)
	NTE_HD static u32
	insertion_candidate(u8 index_char, u32 i, u8* out)
	{
		u32 extra = 0, first = 0, count = 1;
		while (i >= first + count) {
			first += count;
			count *= 4;
			extra++;
		}
		u32 r = i - first;
		out[0] = index_char;
		for (u32 q = 0; q < extra; q++) {
			u32 d = r & 3;
			out[extra - q] = d == 0 ? 'A' : d == 1 ? 'C' : d == 2 ? 'G' : 'T';
			r >>= 2;
		}
		return extra + 1;
	}

	// ------------------------------------------------ window of a failing position
	// Every candidate evaluated at one failing position (step 2, the <=4
	// substitutions, the <=341 insertions per index base, the <=10 deletions)
	// rolls over the SAME characters: O[i], the i-th character leaving at the
	// head, and I[i], the i-th character entering behind the tail.  They are
	// collected once (as 4-bit codes) so that a candidate costs ALU + its Bloom
	// probes instead of a walk over the rope in global memory.  The fast
	// evaluators below are used only when the next k+max_del+1 rolls all succeed
	// (never true within k+d of a contig end); otherwise the general rope-walking
	// code paths, which restate the reference loop by loop, are used.
	NTE_HD u32
	win_len_in() const
	{
		return p.k + p.max_deletions + 1;
	}

	// window storage: win_bytes() bytes.  After a "clean" fill it simply holds the codes of
	// the draft from wc_pos0 on (O and I are the same stretch of sequence, I = O shifted by
	// k), so the following failing positions of the same absent run reuse it with an offset
	// instead of being re-read; the extra WIN_AHEAD codes make that possible.
	static constexpr u32 WIN_AHEAD = 32;

	NTE_HD u32
	win_bytes() const
	{
		return 2 * p.k + p.max_deletions + 1 + WIN_AHEAD;
	}

	NTE_HD u8
	win_o(u32 i) const
	{
		return e.win[(u64)(win_off + i) * e.win_stride];
	}

	NTE_HD u8
	win_i(u32 i) const
	{
		return e.win[(u64)(win_off + p.k + i) * e.win_stride];
	}

	NTE_HD bool
	fill_window()
	{
		if (!e.win) {
			return false; // no window storage: always take the general code paths
		}
		const u32 K = win_len_in();
		Node hn = nget(h_node);
		// clean fast fill: both cursors in one position node that extends far enough
		if (h_node == t_node && hn.type == 0 && !tmp_on && n_ov == 0 && t_seq_i == h_seq_i + p.k - 1 &&
		    (u64)t_seq_i + K <= hn.e_pos && h_seq_i >= hn.s_pos) {
			if (wc_valid && h_seq_i >= wc_pos0 && h_seq_i + p.k + K <= wc_pos0 + wc_len) {
				win_off = h_seq_i - wc_pos0; // still covered by the last clean fill
				return true;
			}
			// read the draft in aligned 8-byte words (a byte loop costs one load per base)
			u32 want = p.k + K + WIN_AHEAD;
			const u64 room = (u64)hn.e_pos + 1 - h_seq_i;
			if (want > room) {
				want = (u32)room;
			}
			const u64 g0 = e.gbase + h_seq_i; // byte index in the batch buffer
			const u8* base = e.seq - e.gbase; // batch buffer start (16-byte aligned)
			const u64 a0 = g0 & ~7ULL;
			u32 filled = 0;
			u32 skip = (u32)(g0 - a0);
			bool exotic = false;
			for (u64 a = a0; filled < want; a += 8) {
				u64 w = 0;
				if (base + a + 8 <= e.batch_end) {
					w = *reinterpret_cast<const u64*>(base + a);
				} else {
					for (u32 b = 0; b < 8 && base + a + b < e.batch_end; b++) {
						w |= (u64)base[a + b] << (8 * b);
					}
				}
				w >>= 8 * skip;
				for (u32 b = skip; b < 8 && filled < want; b++) {
					const u8 ch = (u8)(w & 0xFF);
					const u8 code = char_code(ch);
					exotic |= code == CODE_BAD && is_exotic(ch);
					e.win[(u64)filled * e.win_stride] = code;
					w >>= 8;
					filled++;
				}
				skip = 0;
			}
			if (exotic) {
				// a byte whose hash seeds the 4-bit codes cannot express (U, '-', ...): let the
				// reference-shaped code paths hash it from the raw bytes
				wc_valid = false;
				return false;
			}
			wc_valid = true;
			wc_pos0 = h_seq_i;
			wc_len = want;
			win_off = 0;
			return true;
		}
		// general case (window overlaps edits): walk the rope, copying the stretches that
		// lie in position nodes straight from the draft (overlay applied afterwards)
		wc_valid = false;
		win_off = 0;
		if (tmp_on) {
			return false;
		}
		if (!collect_codes(h_seq_i, h_node, false, p.k, 0)) {
			return false;
		}
		return collect_codes(t_seq_i, t_node, true, K, p.k);
	}

	// Writes `count` character codes of the edited sequence into the window at dst..:
	// starting AT cursor (pos, node) or, when after_cursor, with the character behind it --
	// the characters roll() would deliver.  false if the rope ends first (contig end, unset
	// slot), in which case the caller falls back to the reference-shaped code path.
	NTE_HD bool
	collect_codes(u32 pos, u32 node, bool after_cursor, u32 count, u32 dst)
	{
		u32 done = 0;
		bool skip_one = after_cursor;
		while (done < count) {
			const Node n = nget(node);
			if (node >= nsize || pos >= e.len) {
				return false;
			}
			if (n.type == 1) {
				if (!skip_one) {
					e.win[(u64)(dst + done) * e.win_stride] = char_code(n.c);
					done++;
				}
				skip_one = false;
				node++;
				const Node nx = nget(node);
				if (nx.type == 0) {
					pos = nx.s_pos;
				}
				continue;
			}
			if (n.type != 0 || pos < n.s_pos || pos > n.e_pos) {
				return false;
			}
			u32 from = pos;
			if (skip_one) {
				from++;
				skip_one = false;
			}
			u32 avail = from <= n.e_pos ? n.e_pos - from + 1 : 0;
			u32 take = count - done < avail ? count - done : avail;
			if (take) {
				const u64 g0 = e.gbase + from;
				const u8* base = e.seq - e.gbase;
				const u64 a0 = g0 & ~7ULL;
				u32 filled = 0;
				u32 skip = (u32)(g0 - a0);
				for (u64 a = a0; filled < take; a += 8) {
					u64 w = 0;
					if (base + a + 8 <= e.batch_end) {
						w = *reinterpret_cast<const u64*>(base + a);
					} else {
						for (u32 b = 0; b < 8 && base + a + b < e.batch_end; b++) {
							w |= (u64)base[a + b] << (8 * b);
						}
					}
					w >>= 8 * skip;
					for (u32 b = skip; b < 8 && filled < take; b++) {
						const u8 ch = (u8)(w & 0xFF);
						const u8 code = char_code(ch);
						if (code == CODE_BAD && is_exotic(ch)) {
							return false; // see fill_window
						}
						e.win[(u64)(dst + done + filled) * e.win_stride] = code;
						w >>= 8;
						filled++;
					}
					skip = 0;
				}
				// modified draft characters inside the copied stretch
				for (u32 i = 0; i < n_ov; i++) {
					const u32 op = e.ov_pos[i];
					if (op >= from && op < from + take) {
						e.win[(u64)(dst + done + (op - from)) * e.win_stride] = char_code(e.ov_chr[i]);
					}
				}
				done += take;
			}
			if (done < count) {
				// leave this node the way increment() does
				pos = n.e_pos + 1;
				node++;
				const Node nx = nget(node);
				if (nx.type == 0) {
					pos = nx.s_pos;
				}
			}
		}
		return true;
	}

	// number of subset positions (kk % jump == 0) in (kk, last]
	NTE_HD u32
	probes_left(u32 kk, u32 last) const
	{
		if (kk >= last) {
			return 0;
		}
		return last / p.jump - kk / p.jump;
	}

	// fast form of try_deletion's support count (ntedit.cpp:1479-1519)
	NTE_HD u32
	fast_deletion_support(u8 draft_code, u32 num_del) const
	{
		NTE_COUNT(del_cands, 1);
		HashState ts = hs;
		hash_changelast(ts, e.tab, draft_code, win_i(num_del - 1));
		const SubsetResult r = subset_scan<8, true>(
		    ts, 1, p.k - 2, true, true, ts.fh + ts.rh, p.thr_edit_del, 0, [&](u32 kk, HashState& t) {
			    hash_roll(t, e.tab, win_o(kk - 1), win_i(num_del + kk - 1));
			    return true;
		    });
		return (!r.gave_up && r.present >= p.thr_edit_del) ? r.present : 0;
	}

	// fast form of the insertion support count (ntedit.cpp:1600-1645);
	// ins = inserted bases (m of them, ins[0] = index base)
	NTE_HD u32
	fast_insertion_support(u8 draft_code, const u8* ins, u32 m) const
	{
		NTE_COUNT(ins_cands, 1);
		HashState ts = hs;
		hash_changelast(ts, e.tab, draft_code, char_code(ins[0]));
		const SubsetResult r =
		    subset_scan<8, true>(ts, 0, p.k - 2, true, false, 0, p.thr_edit, 0, [&](u32 kk, HashState& t) {
			    u8 in;
			    if (kk + 1 < m) {
				    in = char_code(ins[kk + 1]);
			    } else if (kk + 1 == m) {
				    in = draft_code;
			    } else {
				    in = win_i(kk - m);
			    }
			    hash_roll(t, e.tab, win_o(kk), in);
			    return true;
		    });
		return r.gave_up ? 0 : r.present;
	}

	// ntedit.cpp:1451-1545; returns the support (0 = rejected)
	NTE_HD u32
	try_deletion(u8 draft_char, u32 num_deletions, u8* deleted, u32& n_deleted)
	{
		if (win_ok) {
			// only the LENGTH of the deleted run is consumed downstream
			n_deleted = num_deletions;
			for (u32 i = 0; i < num_deletions; i++) {
				deleted[i] = 0;
			}
			return fast_deletion_support(char_code(draft_char), num_deletions);
		}
		HashState ts = hs;
		u32 th = h_seq_i, tt = t_seq_i, thn = h_node, ttn = t_node;
		u8 char_out = 0, char_in = 0;
		n_deleted = 0;
		for (u32 i = 0; i < num_deletions; i++) {
			deleted[n_deleted++] = get_character(tt, nget(ttn));
			increment(tt, ttn);
		}
		changelast(ts, draft_char, get_character(tt, nget(ttn)));
		u32 check_present = 0;
		if (present_solid(ts)) {
			check_present++;
		}
		for (u32 k = 1; k <= (p.k - 2) && th < e.len; k++) {
			if (roll(th, tt, thn, ttn, char_out, char_in)) {
				roll_hash(ts, char_out, char_in);
				if (k % p.jump == 0 && present_solid(ts)) {
					check_present++;
				}
			}
		}
		return check_present >= p.thr_edit_del ? check_present : 0;
	}

	NTE_HD static void
	copy_bytes(u8* dst, const u8* src, u32 n)
	{
		for (u32 i = 0; i < n; i++) {
			dst[i] = src[i];
		}
	}

	// Mode-0 sweep (first accepted candidate wins, ntedit.cpp:1587-1730), candidates
	// evaluated wave_size at a time.  The reference tries, in this order,
	//   ins[0], del(nd0), ins[1], del(nd0+1), ... , ins[D-1], del(nd0+D-1), ins[D], ins[D+1], ...
	// (one deletion of growing length after each of the first D insertions, D = deletion
	// lengths still untried at this failing position).  try number t of that list goes to
	// lane t % wave_size; the first accepted try in list order is the result, so evaluating
	// later tries speculatively cannot change it.  Needs the character window (win_ok).
	// limit: stop after the first `limit` tries of the list (0 = all of them); returns 1 accepted, 0 nothing accepted
	// among all tries, -1 undecided (none of the first `limit`, more to try: nothing has been changed)
	NTE_HD int
	try_indels_first_accepted(u8 draft_char, u8 index_char, u32& num_deletions, Best& b, u32 limit = 0)
	{
		const u32 W = e.wave_size;
		const u32 lane = wave_lane();
		const u8 draft_code = char_code(draft_char);
		const u32 nd0 = num_deletions;
		u32 D = nd0 <= p.max_deletions ? p.max_deletions - nd0 + 1 : 0;
		if (D > p.ins_tries) {
			D = p.ins_tries;
		}
		const u32 total = p.ins_tries + D;
		const u32 stop = limit && limit < total ? limit : total;
		for (u32 base = 0; base < stop; base += W) {
			const u32 t = base + lane;
			u32 support = 0;
			bool is_del = false;
			u32 idx = 0; // insertion index or deletion length
			if (t < stop) {
				if (t < 2 * D) {
					is_del = (t & 1) != 0;
					idx = is_del ? nd0 + (t >> 1) : (t >> 1);
				} else {
					idx = t - D;
				}
				if (is_del) {
					support = fast_deletion_support(draft_code, idx);
				} else {
					u8 ins[12];
					const u32 m = insertion_candidate(index_char, idx, ins);
					const u32 cp = fast_insertion_support(draft_code, ins, m);
					support = cp >= p.thr_edit ? cp : 0;
				}
			}
			const u64 acc = wave_ballot(support > 0);
			if (acc) {
				u32 win = 0;
				while (!((acc >> win) & 1)) {
					win++;
				}
				const u32 tw = base + win;
				const u32 sup = wave_shfl(support, win);
				if (tw < 2 * D && (tw & 1)) {
					b.edit_type = 3;
					b.n_indel = nd0 + (tw >> 1);
					for (u32 i = 0; i < b.n_indel && i < 12; i++) {
						b.indel[i] = 0; // only the length of a deletion is consumed
					}
				} else {
					const u32 ii = tw < 2 * D ? (tw >> 1) : tw - D;
					b.edit_type = 2;
					b.n_indel = insertion_candidate(index_char, ii, b.indel);
				}
				b.num_support = sup;
				return 1;
			}
		}
		if (stop < total) {
			return -1;
		}
		num_deletions = nd0 + D;
		return 0;
	}

	// scratch byte t of the sweep below: the KMP table and the previous-insertion string are idle during a sweep
	// (3 x node_window >= 504 bytes; a sweep has at most 341 + 10 tries)
	NTE_HD u8&
	sweep_byte(u32 t) const
	{
		const u32 w2 = 2 * p.node_window;
		return t < w2 ? reinterpret_cast<u8*>(e.lps)[t] : e.prev[t - w2];
	}

	// Modes 1 and 2 (the best candidate wins, ntedit.cpp:1587-1744), with the character window: the support of EVERY
	// try of the list (see try_indels_first_accepted for its order) is computed first -- try t by lane t % wave_size --
	// and then the reference's running best ("a try that reaches the bar and is at least as good as the best so far takes
	// over, the support it displaces becomes the alternate") is replayed over the supports in list order.
	NTE_HD bool
	try_indels_all(u8 draft_char, u8 index_char, u32& num_deletions, Best& b)
	{
		const u32 W = e.wave_size;
		const u32 lane = wave_lane();
		const u8 draft_code = char_code(draft_char);
		const u32 nd0 = num_deletions;
		u32 D = nd0 <= p.max_deletions ? p.max_deletions - nd0 + 1 : 0;
		if (D > p.ins_tries) {
			D = p.ins_tries;
		}
		const u32 total = p.ins_tries + D;
		for (u32 t = lane; t < total; t += W) {
			u32 support = 0;
			if (t < 2 * D && (t & 1)) {
				support = fast_deletion_support(draft_code, nd0 + (t >> 1));
			} else {
				u8 ins[12];
				const u32 m = insertion_candidate(index_char, t < 2 * D ? (t >> 1) : t - D, ins);
				const u32 cp = fast_insertion_support(draft_code, ins, m);
				support = cp >= p.thr_edit ? cp : 0;
			}
			sweep_byte(t) = (u8)(support < 255 ? support : 255); // (a support is at most k <= 200)
		}
		lanes_sync();
		u32 best_support = 0, alt_support = 0, best_t = 0;
		for (u32 t = 0; t < total; t++) {
			const u32 s = sweep_byte(t);
			if (s && s >= best_support) {
				if (best_support) {
					alt_support = best_support;
				}
				best_support = s;
				best_t = t;
			}
		}
		lanes_sync(); // (the scratch goes back to its owners)
		num_deletions = nd0 + D;
		if (best_support == 0) {
			return false;
		}
		if ((mode() == 2 && best_support > b.num_support) || mode() == 1) {
			if (best_t < 2 * D && (best_t & 1)) {
				b.edit_type = 3;
				b.n_indel = nd0 + (best_t >> 1);
				for (u32 i = 0; i < b.n_indel && i < 12; i++) {
					b.indel[i] = 0; // only the length of a deletion is consumed
				}
			} else {
				b.edit_type = 2;
				b.n_indel = insertion_candidate(index_char, best_t < 2 * D ? (best_t >> 1) : best_t - D, b.indel);
			}
			b.num_support = best_support;
			b.altsupp1 = alt_support;
		}
		return true;
	}

	// ntedit.cpp:1548-1744
	NTE_HD bool
	try_indels(u8 draft_char, u8 index_char, u32& num_deletions, Best& b)
	{
		NTE_COUNT(sweeps, 1);
		if (mode() == 0 && win_ok) {
			return try_indels_first_accepted(draft_char, index_char, num_deletions, b) > 0;
		}
		if (win_ok && p.ins_tries + p.max_deletions + 1 <= 3 * p.node_window) {
			return try_indels_all(draft_char, index_char, num_deletions, b);
		}
		u32 temp_best_support = 0, temp_alt_support = 0;
		u8 temp_best_indel[12];
		u32 temp_best_n = 0;
		u32 temp_best_type = 0;
		u8 char_in = 0, char_out = 0;

		for (u32 i = 0; i < p.ins_tries; i++) {
			u8 ins[12];
			u32 n_ins = insertion_candidate(index_char, i, ins);
			ins[n_ins++] = draft_char;

			u32 check_present = 0;
			if (win_ok) {
				check_present = fast_insertion_support(char_code(draft_char), ins, n_ins - 1);
			} else {
			HashState ts = hs;
			u32 th = h_seq_i, tt = t_seq_i, thn = h_node, ttn = t_node;
			changelast(ts, draft_char, index_char);
			u32 k = 0;
			// k-mers that end inside the inserted bases
			for (; k < n_ins - 1 && th < e.len; k++) {
				roll_hash(ts, get_character(th, nget(thn)), ins[k + 1]);
				increment(th, thn);
				if (k % p.jump == 0 && present_solid(ts)) {
					check_present++;
				}
			}
			// k-mers that end behind the insertion
			for (; k < p.k - 1 && th < e.len; k++) {
				if (roll(th, tt, thn, ttn, char_out, char_in)) {
					roll_hash(ts, char_out, char_in);
					if (k % p.jump == 0 && present_solid(ts)) {
						check_present++;
					}
				}
			}
			}
			n_ins--; // drop the draft base again
			if (check_present >= p.thr_edit) {
				if (mode() == 0) {
					b.edit_type = 2;
					copy_bytes(b.indel, ins, n_ins);
					b.n_indel = n_ins;
					b.num_support = check_present;
					return true;
				}
				if (check_present >= temp_best_support) {
					if (temp_best_support) {
						temp_alt_support = temp_best_support;
					}
					temp_best_type = 2;
					copy_bytes(temp_best_indel, ins, n_ins);
					temp_best_n = n_ins;
					temp_best_support = check_present;
				}
			}

			if (num_deletions <= p.max_deletions) {
				u8 deleted[12];
				u32 n_deleted = 0;
				u32 del_support = try_deletion(draft_char, num_deletions, deleted, n_deleted);
				if (del_support > 0) {
					if (mode() == 0) {
						b.edit_type = 3;
						copy_bytes(b.indel, deleted, n_deleted);
						b.n_indel = n_deleted;
						b.num_support = del_support;
						return true;
					}
					if (del_support >= temp_best_support) {
						if (temp_best_support) {
							temp_alt_support = temp_best_support;
						}
						temp_best_type = 3;
						copy_bytes(temp_best_indel, deleted, n_deleted);
						temp_best_n = n_deleted;
						temp_best_support = del_support;
					}
				}
				num_deletions++;
			}
		}

		if (temp_best_support > 0) {
			if ((mode() == 2 && temp_best_support > b.num_support) || mode() == 1) {
				b.edit_type = temp_best_type;
				copy_bytes(b.indel, temp_best_indel, temp_best_n);
				b.n_indel = temp_best_n;
				b.num_support = temp_best_support;
				b.altsupp1 = temp_alt_support;
			}
			return true;
		}
		return false;
	}

	NTE_HD void
	note_sub(u32 pos)
	{
		if ((int64_t)pos > last_sub_pos) {
			last_sub_pos = pos;
		}
	}

	NTE_HD void
	reseed_after_skip()
	{
		// findAcceptedKmer + NTMC64 seed (ntedit.cpp:1335-1343); the hash of a
		// missing k-mer is defined as 0 (the reference hashes "" there)
		find_accepted_kmer();
	}

	// ntedit.cpp:1250-1448
	NTE_HD void
	make_edit(u8 draft_char, Best& b)
	{
		Node t_nd = nget(t_node);
		switch (b.edit_type) {
		case 1: {
			if (t_nd.type == 0) {
				set_seq(t_seq_i, b.sub_base);
				note_sub(t_seq_i);
				u8 a1 = 0, a2 = 0, a3 = 0;
				u32 s1 = 0, s2 = 0, s3 = 0;
				if (b.altsupp1 && b.altbase1 != b.sub_base) {
					a1 = b.altbase1;
					s1 = b.altsupp1;
				}
				if (b.altsupp2 && b.altbase2 != b.altbase1) {
					a2 = b.altbase2;
					s2 = b.altsupp2;
				}
				if (b.altsupp3 && b.altbase3 != b.altbase2) {
					a3 = b.altbase3;
					s3 = b.altsupp3;
				}
				Item it;
				it.w[0] = TAG_SUB | ((u32)draft_char << 8) | ((u32)b.sub_base << 16) |
				          ((b.num_support & 0xFF) << 24);
				it.w[1] = t_seq_i;
				it.w[2] = (u32)a1 | ((s1 & 0xFF) << 8) | ((u32)a2 << 16) | ((s2 & 0xFF) << 24);
				it.w[3] = (u32)a3 | ((s3 & 0xFF) << 8);
				emit(it);
			} else if (t_nd.type == 1) {
				t_nd.c = b.sub_base;
				nset(t_node, t_nd);
				rope_touched = true;
				note_sub(t_seq_i);
			}
			changelast(hs, draft_char, b.sub_base);
			break;
		}
		case 2: {
			bool skipped_repeat = false;
			u8* prev = e.prev;
			int16_t* lps = e.lps;
			u32 n_prev = get_prev_insertion(prev);
			if (n_prev + 16 >= p.node_window) {
				flags |= EV_OVERFLOW;
				break;
			}
			if (n_prev + b.n_indel >= p.k) {
				if (is_repeat(prev, (int)n_prev, lps) || n_prev + b.n_indel >= p.insertion_cap) {
					drop_prev_insertion(n_prev);
					reseed_after_skip();
					skipped_repeat = true;
				} else {
					for (u32 w = 0; w < b.n_indel; w++) {
						for (u32 q = n_prev; q > 0; q--) {
							prev[q] = prev[q - 1];
						}
						prev[0] = rc_char(b.indel[w]);
						n_prev++;
						if (is_repeat(prev, (int)n_prev, lps)) {
							drop_prev_insertion(n_prev - w);
							reseed_after_skip();
							skipped_repeat = true;
						}
					}
				}
			}
			if (!skipped_repeat) {
				make_insertion(t_node, t_seq_i, b.indel, b.n_indel, b.num_support);
				changelast(hs, draft_char, b.indel[0]);
			}
			break;
		}
		case 3:
			make_deletion(t_node, t_seq_i, b.n_indel, b.num_support);
			changelast(hs, draft_char, get_character(t_seq_i, nget(t_node)));
			break;
		case 0:
			if (mask()) {
				u8 lc = (draft_char >= 'A' && draft_char <= 'Z') ? (u8)(draft_char + 32) : draft_char;
				if (t_nd.type == 0) {
					set_seq(t_seq_i, lc);
				} else if (t_nd.type == 1) {
					t_nd.c = lc;
					nset(t_node, t_nd);
					rope_touched = true;
				}
				changelast(hs, draft_char, lc);
			}
			if (snv() && b.altsupp1) {
				// -s 1: a position that keeps its base but has supported alternatives is still
				// reported (VCF only): sub_base == draft_char marks "no edit" (ntedit.cpp:1428-1443)
				Item it;
				it.w[0] = TAG_SUB | ((u32)draft_char << 8) | ((u32)draft_char << 16) | ((b.num_support & 0xFF) << 24);
				it.w[1] = t_seq_i;
				it.w[2] = (u32)b.altbase1 | ((b.altsupp1 & 0xFF) << 8) | ((u32)b.altbase2 << 16) |
				          ((b.altsupp2 & 0xFF) << 24);
				it.w[3] = (u32)b.altbase3 | ((b.altsupp3 & 0xFF) << 8);
				emit(it);
			}
			break;
		default:
			break;
		}
	}

	// substitution candidates for a draft base: polish_bases_array / snv_bases_array
	// (ntedit.cpp:180-199)
	NTE_HD static u32
	candidate_bases(u8 draft_char, bool snv, u8* out)
	{
		const char* s;
		switch (draft_char) {
		case 'A':
			s = "TCG";
			break;
		case 'T':
			s = "ACG";
			break;
		case 'C':
			s = "ATG";
			break;
		case 'G':
			s = "ATC";
			break;
		case 'R':
			s = snv ? "ATCG" : "TC";
			break;
		case 'Y':
			s = snv ? "ATCG" : "AG";
			break;
		case 'S':
			s = snv ? "ATCG" : "AT";
			break;
		case 'W':
			s = snv ? "ATCG" : "CG";
			break;
		case 'K':
			s = snv ? "ATCG" : "AC";
			break;
		case 'M':
			s = snv ? "ATCG" : "TG";
			break;
		case 'B':
			s = snv ? "ATCG" : "A";
			break;
		case 'D':
			s = snv ? "ATCG" : "C";
			break;
		case 'H':
			s = snv ? "ATCG" : "G";
			break;
		case 'V':
			s = snv ? "ATCG" : "T";
			break;
		case 'N':
			s = "ATCG";
			break;
		default:
			s = "";
		}
		u32 n = 0;
		while (s[n]) {
			out[n] = (u8)s[n];
			n++;
		}
		return n;
	}

	// The state is "clean" when everything the machine will do from here on is
	// a function of the un-edited draft: both cursors sit in the open-ended last
	// position node, the window is k contiguous draft bases, no substituted
	// base is still inside it.
	// is_clean() for cursors held by the caller (tail = nget(tn))
	NTE_HD bool
	clean_at(u32 hs_i, u32 ts_i, u32 hn, u32 tn, const Node& tail) const
	{
		if (hn != tn) {
			return false;
		}
		if (tail.type != 0 || tail.e_pos != e.len - 1) {
			return false;
		}
		if (tn + 1 < nsize && nget(tn + 1).type != -1) {
			return false;
		}
		if (ts_i != hs_i + p.k - 1 || hs_i < tail.s_pos) {
			return false;
		}
		return (int64_t)hs_i > last_sub_pos;
	}

	NTE_HD bool
	is_clean() const
	{
		if (h_node != t_node) {
			return false;
		}
		Node n = nget(t_node);
		if (n.type != 0 || n.e_pos != e.len - 1) {
			return false;
		}
		if (t_node + 1 < nsize && nget(t_node + 1).type != -1) {
			return false;
		}
		if (t_seq_i != h_seq_i + p.k - 1 || h_seq_i < n.s_pos) {
			return false;
		}
		// indels keep the state dirty until both cursors have moved into the
		// open node behind them, which the node tests above detect
		return (int64_t)h_seq_i > last_sub_pos;
	}

	// After an edit the next k-1 k-mers contain the edited base(s) and have to be probed one
	// by one as the cursors roll on (ntedit.cpp:1806 at every position).  Their hashes only
	// depend on characters that are already known, so they are computed ahead from the
	// window and probed together: in groups of 8 by a single thread, one k-mer per lane by a
	// wavefront.  Bit i of la_mask = "the k-mer i rolls ahead of the cursor is present".
	NTE_HD void
	build_lookahead()
	{
		la_n = la_i = 0;
		la_mask = 0;
		const bool use_known = la_known_pos == h_seq_i; // (see process_missing; good for this one look-ahead only)
		la_known_pos = NONE32;
		if (!fill_window()) {
			la_off = true; // near the contig end: probe position by position
			return;
		}
		u32 L = p.k < 32 ? p.k : 32;
		// a non-accepted character ends the stretch the main loop walks position by position
		for (u32 i = 0; i + 1 < L; i++) {
			if (win_i(i) == CODE_BAD) {
				L = i + 1;
				break;
			}
		}
		if (e.wave_size > 1) {
			const u32 lane = wave_lane();
			for (u32 base = 0; base < L; base += e.wave_size) {
				bool present = false;
				if (base + lane < L) {
					HashState ts = hs;
					for (u32 i = 0; i < base + lane; i++) {
						hash_roll(ts, e.tab, win_o(i), win_i(i));
					}
					present = !screen_absent(ts);
				}
				la_mask |= (u32)((wave_ballot(present) & 0xFFFFFFFFull) << base);
			}
			la_n = L;
			la_win = true;
			return;
		}
		HashState ts = hs;
		u32 n = 0;
		// (k-mers the last substitution's support count has probed already: see process_missing)
		const u32 known = use_known ? la_known : 0u;
		const u32 known_vals = use_known ? la_known_vals : 0u;
		while (n < L) {
			u64 b[8];
			u32 nb = 0;
			NTE_UNROLL
			for (int u = 0; u < 8; u++) {
				b[u] = 0;
				if (n + (u32)u < L) {
					if (n + (u32)u > 0) {
						hash_roll(ts, e.tab, win_o(n + u - 1), win_i(n + u - 1));
					}
					b[u] = ts.fh + ts.rh;
					nb = (u32)u + 1;
				}
			}
			const u32 skip = (known >> n) & 0xFFu;
			la_mask |= (probe_group_range<8>(e.bloom, b, nb, p.min_thr > 1 ? p.min_thr : 1, 255, ~skip) | ((known_vals >> n) & skip & ((1u << nb) - 1))) << n;
			n += nb;
		}
		la_n = L;
		la_win = true;
	}

	// ranks a substitution candidate that reached the support bar among the ones seen so far
	// (best + up to three alternates, ntedit.cpp:1999-2050)
	NTE_HD static void
	note_candidate(Best& b, u8 sub_base, u32 check_present)
	{
		if (check_present >= b.num_support) {
			if (b.altsupp2) {
				b.altbase3 = b.altbase2;
				b.altsupp3 = b.altsupp2;
			}
			if (b.altsupp1) {
				b.altbase2 = b.altbase1;
				b.altsupp2 = b.altsupp1;
			}
			if (b.num_support) {
				b.altsupp1 = b.num_support;
				b.altbase1 = b.sub_base;
			}
			b.edit_type = 1;
			b.sub_base = sub_base;
			b.num_support = check_present;
		} else {
			if (!b.altsupp1) {
				b.altbase1 = sub_base;
				b.altsupp1 = check_present;
			} else if (!b.altsupp2) {
				if (check_present < b.altsupp1) {
					b.altbase2 = sub_base;
					b.altsupp2 = check_present;
				} else {
					b.altbase2 = b.altbase1;
					b.altsupp2 = b.altsupp1;
					b.altbase1 = sub_base;
					b.altsupp1 = check_present;
				}
			} else if (!b.altsupp3) {
				if (check_present < b.altsupp2) {
					b.altbase3 = sub_base;
					b.altsupp3 = check_present;
				} else if (check_present < b.altsupp1) {
					b.altbase3 = b.altbase2;
					b.altsupp3 = b.altsupp2;
					b.altbase2 = sub_base;
					b.altsupp2 = check_present;
				} else {
					b.altbase3 = b.altbase2;
					b.altsupp3 = b.altsupp2;
					b.altbase2 = b.altbase1;
					b.altsupp2 = b.altsupp1;
					b.altbase1 = sub_base;
					b.altsupp1 = check_present;
				}
			}
		}
	}

	// steps 2-5 + makeEdit for the k-mer currently under the cursors
	NTE_HD void
	process_missing(u8 char_in_at_t)
	{
		HashState ts = hs;
		u32 th = h_seq_i, tt = t_seq_i, thn = h_node, ttn = t_node;
		u8 draft_char = char_in_at_t;
		if (draft_char >= 'a' && draft_char <= 'z') {
			draft_char -= 32;
		}
		u8 char_out = 0, char_in = 0;

		// step 2: confirm on the k/j subset (ntedit.cpp:1826-1858)
		u32 check_missing = 0;
		bool do_not_fix = false;
		NTE_PROF_SUB(7); // (time outside process_missing)
		win_ok = fill_window();
		NTE_PROF_SUB(0);
		u32 check_there = 0, there_median = 0;
		if (counting() || snv()) {
			// counting filter / SNV mode (ntedit.cpp:1842-1861,1873,1890-1914): besides the
			// missing count, the k-mers that ARE there matter -- their median coverage decides
			// whether a fix is attempted, and in SNV mode their number is the draft base's own
			// support that a substitution has to match
			for (u32 k = 0; k < p.k && th < e.len; k++) {
				u8 in;
				if (win_ok) {
					in = win_i(k);
					hash_roll(ts, e.tab, win_o(k), in);
				} else {
					if (!roll(th, tt, thn, ttn, char_out, char_in)) {
						do_not_fix = true;
						break;
					}
					in = char_code(char_in);
					roll_hash(ts, char_out, char_in);
				}
				if (in == CODE_BAD) {
					do_not_fix = true;
					break;
				}
				if (k % p.jump == 0) {
					const u32 c = counting() ? count_of(ts) : (in_bloom(ts) ? 1u : 0u);
					if (c == 0) {
						check_missing++;
					} else if ((draft_char == 'A' || draft_char == 'C' || draft_char == 'G' || draft_char == 'T') &&
					           c >= p.min_thr) {
						if (check_there < p.node_window) {
							e.prev[check_there] = (u8)c;
						}
						check_there++;
					}
				}
			}
			// median of the coverages (ntedit.cpp:455-463): sort, take element n/2; 0 when empty
			u32 median = 0;
			const u32 nm = check_there < p.node_window ? check_there : p.node_window;
			if (nm) {
				for (u32 i = 1; i < nm; i++) {
					const u8 v = e.prev[i];
					u32 j = i;
					while (j > 0 && e.prev[j - 1] > v) {
						e.prev[j] = e.prev[j - 1];
						j--;
					}
					e.prev[j] = v;
				}
				median = e.prev[nm / 2];
			}
			there_median = median;
			if (!snv() &&
			    (do_not_fix || !(check_missing >= p.thr_missing || (counting() && median < p.min_thr)))) {
				return;
			}
		} else if (win_ok && wc_valid && is_clean()) {
			// Clean state: the k-mers of the confirmation subset are un-edited draft k-mers,
			// i.e. exactly what the screening pass already answered -- read their bits
			// (bit = all bases accepted AND absent) instead of probing the filter again.
			for (u32 k = 0; k < p.k; k++) {
				if (win_i(k) == CODE_BAD) {
					do_not_fix = true;
					break;
				}
			}
			if (!do_not_fix) {
				const u64 g = e.gbase + h_seq_i + 1;
				for (u32 k = 0; k < p.k; k += p.jump) {
					check_missing += bit_absent(e.bitmap, g + k) ? 1u : 0u;
				}
			}
		} else if (win_ok) {
			const SubsetResult r =
			    subset_scan<8>(ts, 0, p.k - 1, false, false, 0, 0, p.thr_missing, [&](u32 k, HashState& t) {
				    const u8 in = win_i(k);
				    hash_roll(t, e.tab, win_o(k), in);
				    return in != CODE_BAD;
			    });
			if (r.aborted) {
				do_not_fix = true;
			} else if (r.gave_up) {
				return; // the confirmation can no longer succeed
			}
			check_missing = r.total - r.present;
		} else
		for (u32 k = 0; k < p.k && th < e.len; k++) {
			if (roll(th, tt, thn, ttn, char_out, char_in)) {
				roll_hash(ts, char_out, char_in);
				if (char_code(char_in) == CODE_BAD) {
					do_not_fix = true;
					break;
				}
				if (k % p.jump == 0 && !in_bloom(ts)) {
					check_missing++;
				}
			} else {
				do_not_fix = true;
				break;
			}
		}
		if ((!snv() && (do_not_fix || (!counting() && check_missing < p.thr_missing))) || p.debug_stop == 2) {
			return;
		}

		NTE_PROF_SUB(1); // step 2
		u32 num_deletions = 1;
		Best b;
		b.edit_type = 0;
		b.n_indel = 0;
		b.sub_base = 0; // (U2) the reference leaves these uninitialised
		b.num_support = 0;
		b.altbase1 = b.altbase2 = b.altbase3 = 0;
		b.altsupp1 = b.altsupp2 = b.altsupp3 = 0;
		if (snv() && check_there >= p.thr_edit) {
			// the draft base's own support is the bar (ntedit.cpp:1890-1903)
			b.sub_base = draft_char;
			b.num_support = counting() ? there_median : check_there;
		}

		u8 cand[4];
		u32 n_cand = candidate_bases(draft_char, snv(), cand);
		u64 cand_mask[4] = { 0, 0, 0, 0 }; // presence of the subset k-mers behind every candidate (SubsetResult::pmask)
		u32 cand_total[4] = { 0, 0, 0, 0 };
		Node t_nd = nget(t_node);
		for (u32 ci = 0; ci < n_cand; ci++) {
			u8 sub_base = cand[ci];
			ts = hs;
			changelast(ts, draft_char, sub_base);
			if (!(present_solid(ts) || mode() == 2)) {
				continue;
			}
			th = h_seq_i;
			tt = t_seq_i;
			thn = h_node;
			ttn = t_node;
			// temporarily substitute (ntedit.cpp:1936-1940)
			if (t_nd.type == 0) {
				tmp_on = true;
				tmp_pos = t_seq_i;
				tmp_chr = sub_base;
			} else if (t_nd.type == 1) {
				Node m = t_nd;
				m.c = sub_base;
				nset(t_node, m);
			}
			u32 check_present = 0;
			if (win_ok) {
				// the substituted base is the last one to leave the window
				tmp_on = false;
				const u8 sub_code = char_code(sub_base);
				const u32 last = p.k - 1;
				const SubsetResult r =
				    subset_scan<8>(ts, 0, last, true, false, 0, p.thr_edit, 0, [&](u32 k, HashState& t) {
					    hash_roll(t, e.tab, k == last ? sub_code : win_o(k), win_i(k));
					    return true;
				    });
				check_present = r.gave_up ? 0 : r.present;
				cand_mask[ci & 3] = r.pmask;
				cand_total[ci & 3] = r.gave_up ? 0 : r.total;
			} else
			for (u32 k = 0; k < p.k && th < e.len && tt < e.len; k++) {
				if (roll(th, tt, thn, ttn, char_out, char_in)) {
					roll_hash(ts, char_out, char_in);
					if (k % p.jump == 0 && present_solid(ts)) {
						check_present++;
					}
				} else {
					break;
				}
			}
			// revert -- with the UPPER-cased draft base (ntedit.cpp:1975-1981)
			if (t_nd.type == 0) {
				tmp_on = false;
				set_seq(t_seq_i, draft_char);
			} else if (t_nd.type == 1) {
				Node m = t_nd;
				m.c = draft_char;
				nset(t_node, m);
				t_nd = m;
				rope_touched = true;
			}

			if (check_present >= p.thr_edit) {
				note_candidate(b, sub_base, check_present);
				if (mode() == 0 || mode() == 1) {
					continue;
				}
			}
			if (mode() == 2 || b.edit_type != 1) {
				if (p.debug_stop == 3) {
					return; // timing ablation: everything up to the first indel sweep
				}
				bool accepted;
				if (e.defer_sweeps && p.ins_tries > 0) {
					// the candidate sweep is ~100x the cost of everything else an event
					// does; running it next to 63 cheap lanes would idle the wave, so the
					// first pass hands such events to a second, sweep-only launch.  But two thirds of
					// those events only ever meet sweeps whose first few tries succeed (a one-base
					// indel: the index base alone, or a deletion of one): the first inline_tries
					// candidates are tried here, in list order -- the first accepted one is the sweep's
					// result whatever comes behind it.
					int r = -1;
					if (mode() == 0 && win_ok && p.inline_tries) {
						r = try_indels_first_accepted(draft_char, sub_base, num_deletions, b, p.inline_tries);
					}
					if (r < 0) {
						flags |= EV_DEFERRED;
						return;
					}
					accepted = r > 0;
				} else {
					NTE_PROF_SUB(2);
					accepted = try_indels(draft_char, sub_base, num_deletions, b);
					NTE_PROF_SUB(4); // indel sweep
				}
				if (accepted) {
					if (mode() == 0 || mode() == 1) {
						break;
					}
				}
			}
		}
		NTE_PROF_SUB(2); // candidates (substitutions + indel sweeps)
		changed_seq = b.edit_type != 0;
		la_known = la_known_vals = 0;
		la_known_pos = NONE32;
		if (b.edit_type == 1 && win_ok && mode() != 2 && !secbf() && !counting() && !snv() && t_nd.type == 0 && linear()) {
			// The k-mers behind a substitution are probed again as the cursors roll over them (the look-ahead).  The
			// support count of the winning candidate has probed every jump-th of them already -- with a plain filter and
			// no secondary one "there and solid" is the main loop's "present": the next look-ahead takes those over.
			for (u32 ci = 0; ci < n_cand; ci++) {
				if (cand[ci] == b.sub_base && cand_total[ci] && cand_total[ci] <= 64) {
					// look-ahead index i (built one roll from here) = the k-mer i + 1 rolls behind this one = walk index i,
					// in the subset when i % jump == 0
					for (u32 i = 0, n = 0; i < 32 && n < cand_total[ci]; i += p.jump, n++) {
						la_known |= 1u << i;
						la_known_vals |= (u32)((cand_mask[ci] >> n) & 1) << i;
					}
					la_known_pos = h_seq_i + 1;
				}
			}
		}
		make_edit(draft_char, b);
		NTE_PROF_SUB(3);
	}


	// ------------------------------------------------------------ runs of failing positions, one position per lane
	// A failing position that ends without an edit leaves the machine exactly where it was: the assessment of the next
	// position is a function of the (edited) sequence and the filter alone.  While both cursors sit in the open last
	// position node ("linear": the edited sequence from the head on is the draft with the overlay applied) the next
	// 64 positions are therefore assessed TOGETHER, lane i taking position q + i: every lane runs the same phases on its
	// own offset of one shared character window -- presence of its k-mer where the window still holds a substituted base,
	// step 2, the substitution candidates (ntedit.cpp:1826-2062) -- which tells it whether its position ends with no edit,
	// with a substitution, or needs an indel sweep.  Then the positions are taken in serial order: the first one that
	// is not a plain "no edit" gets its sweeps (all lanes on the candidates of that one position, as before) and, if an
	// edit comes out, the machine is placed there and applies it; everything the lanes behind it computed is dropped
	// (the serial program would have assessed them in another state).  First edit wins + restart is the reference's
	// order, position by position (ntedit.cpp:1798-2139).  On the host (tests/hostsim) the lanes are a loop.
	static constexpr u32 N_LANES = 64;

#if defined(__HIP_DEVICE_COMPILE__)
	template<typename T>
	struct PerLane
	{
		T v;
		NTE_HD T& at(u32) { return v; }
	};
#define NTE_FOR_LANES(l, n) for (u32 l = wave_lane(), nte_once_ = 1; nte_once_ && l < (n); nte_once_ = 0)
#else
	template<typename T>
	struct PerLane
	{
		T v[N_LANES];
		NTE_HD T& at(u32 l) { return v[l]; }
	};
#define NTE_FOR_LANES(l, n) for (u32 l = 0; l < (n); l++)
#endif

	// value of lane `src`
	NTE_HD u32
	lanes_get(PerLane<u32>& x, u32 src) const
	{
#if defined(__HIP_DEVICE_COMPILE__)
		return (u32)__shfl((int)x.v, (int)src, 64);
#else
		return x.v[src];
#endif
	}

	NTE_HD u64
	lanes_get(PerLane<u64>& x, u32 src) const
	{
#if defined(__HIP_DEVICE_COMPILE__)
		const u32 lo = (u32)__shfl((int)(u32)x.v, (int)src, 64);
		const u32 hi = (u32)__shfl((int)(u32)(x.v >> 32), (int)src, 64);
		return ((u64)hi << 32) | lo;
#else
		return x.v[src];
#endif
	}

	// lowest lane in [from, n) whose value is not zero; n if there is none
	NTE_HD u32
	lanes_first(PerLane<u32>& x, u32 from, u32 n) const
	{
#if defined(__HIP_DEVICE_COMPILE__)
		const u32 l = wave_lane();
		const u64 m = __ballot(l >= from && l < n && x.v != 0);
		return m ? (u32)__builtin_ctzll(m) : n;
#else
		for (u32 l = from; l < n; l++) {
			if (x.v[l]) {
				return l;
			}
		}
		return n;
#endif
	}

	// stores of one lane become visible to the others of its wavefront
	NTE_HD void
	lanes_sync() const
	{
#if defined(__HIP_DEVICE_COMPILE__)
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
	}

	// both cursors in the open last position node, k draft positions apart: is_clean() without the test for
	// substituted bases inside the window
	NTE_HD bool
	linear() const
	{
		if (h_node != t_node) {
			return false;
		}
		Node n = nget(t_node);
		if (n.type != 0 || n.e_pos != e.len - 1) {
			return false;
		}
		if (t_node + 1 < nsize && nget(t_node + 1).type != -1) {
			return false;
		}
		return t_seq_i == h_seq_i + p.k - 1 && h_seq_i >= n.s_pos;
	}

	NTE_HD static u8
	code_letter(u8 code)
	{
		// the accepted bases by code (char_code): upper case
		return code < 14 ? (u8)"ACGTRYSWKMBDHV"[code] : (u8)'N';
	}

	// i-th of up to LANE_COUNTS counts packed into four words
	static constexpr u32 LANE_COUNTS = 32;
	NTE_HD static u32
	packed_byte(const u64 (&w)[4], u32 i)
	{
		const u64 v = i < 16 ? (i < 8 ? w[0] : w[1]) : (i < 24 ? w[2] : w[3]);
		return (u32)((v >> (8 * (i & 7))) & 0xFF);
	}

	// The per-lane assessment keeps the counts of the subset k-mers that are there (counting filters, -s 1: their median
	// decides) in registers: configurations whose subset is larger take the position-by-position paths.
	NTE_HD bool
	lane_counts_fit() const
	{
		return !(counting() || snv()) || (p.k - 1) / p.jump + 1 <= LANE_COUNTS;
	}

	// what one lane knows about its position after the phases every lane runs by itself
	enum LaneState : u32
	{
		LANE_NONE = 0,   // the position ends without an edit
		LANE_EDIT = 1,   // a substitution (no indel sweep on the way to it)
		LANE_SWEEP = 2   // the outcome depends on an indel sweep
	};

	// The candidate loop of a failing position (ntedit.cpp:1916-2062) replayed from the candidates' substitution
	// support: S[ci] = bit 31 "the k-mer with candidate ci in place of the draft base is there", low bits = its
	// support on the k/j subset.  sweeps = false: stop where an indel sweep would start (LANE_SWEEP);
	// sweeps = true: run them (all lanes of the wavefront on that one position; hs / win_off are that position's).
	NTE_HD u32
	decide_position(const u32 (&S)[4], u8 draft_char, u32 check_there, u32 there_median, bool sweeps, Best& b)
	{
		b.edit_type = 0;
		b.n_indel = 0;
		b.sub_base = 0;
		b.num_support = 0;
		b.altbase1 = b.altbase2 = b.altbase3 = 0;
		b.altsupp1 = b.altsupp2 = b.altsupp3 = 0;
		if (snv() && check_there >= p.thr_edit) {
			b.sub_base = draft_char;
			b.num_support = counting() ? there_median : check_there;
		}
		u8 cand[4];
		const u32 n_cand = candidate_bases(draft_char, snv(), cand);
		u32 num_deletions = 1;
		for (u32 ci = 0; ci < n_cand; ci++) {
			if (!(S[ci] >> 31)) {
				continue;
			}
			const u32 sup = S[ci] & 0x7FFFFFFFu;
			if (sup >= p.thr_edit) {
				note_candidate(b, cand[ci], sup);
				continue;
			}
			if (b.edit_type != 1 && p.ins_tries > 0) {
				if (!sweeps) {
					return LANE_SWEEP;
				}
				if (try_indels_first_accepted(draft_char, cand[ci], num_deletions, b) > 0) {
					break;
				}
			}
		}
		return b.edit_type ? LANE_EDIT : LANE_NONE;
	}

	// how many of the positions q, q + 1, ... the serial program walks through if none of them makes an edit (<= 64):
	// a position whose window still holds a substituted base is walked whatever the filter says; elsewhere the walk goes
	// on while the screening bitmap has the k-mer absent and no other event starts there.  Every one of them needs the
	// full character window inside the contig.
	NTE_HD u32
	lanes_run_length(u32 q) const
	{
		const u32 K = win_len_in();
#if defined(__HIP_DEVICE_COMPILE__)
		const u32 l = wave_lane();
		const u64 P = (u64)q + l;
		bool ok = P + p.k - 1 + K <= (u64)e.len - 1;
		if (ok && l > 0 && (int64_t)P > last_sub_pos) {
			const u64 g = e.gbase + P;
			ok = bit_absent(e.runmap, g) && !is_event_start(e.runmap, g, p.start_grid);
		}
		const u64 m = ~__ballot(ok);
		return m ? (u32)__builtin_ctzll(m) : N_LANES;
#else
		u32 n = 0;
		for (u32 l = 0; l < N_LANES; l++) {
			const u64 P = (u64)q + l;
			bool ok = P + p.k - 1 + K <= (u64)e.len - 1;
			if (ok && l > 0 && (int64_t)P > last_sub_pos) {
				const u64 g = e.gbase + P;
				ok = bit_absent(e.runmap, g) && !is_event_start(e.runmap, g, p.start_grid);
			}
			if (!ok) {
				break;
			}
			n++;
		}
		return n;
#endif
	}

	// character codes of the edited sequence from position q on for n lanes, into the shared window.  false: a byte
	// the 4-bit codes cannot express (see fill_window) -- the caller takes the reference-shaped paths.  *n_io shrinks
	// to the lanes whose k-mer holds accepted bases only.
	NTE_HD bool
	lanes_fill_window(u32 q, u32* n_io)
	{
		const u32 K = win_len_in();
		u32 n = *n_io;
		u32 want = (n - 1) + p.k + K + WIN_AHEAD;
		const u64 room = (u64)e.len - q;
		if (want > room) {
			want = (u32)room;
		}
		const u8* src = e.seq + q;
		const u32 W = e.wave_size;
		bool exotic = false;
		u32 first_bad = 0xFFFFFFFFu; // (per lane: the lowest index it saw)
		for (u32 i = wave_lane(); i < want; i += W) {
			const u8 ch = src[i];
			const u8 code = char_code(ch);
			exotic |= code == CODE_BAD && is_exotic(ch);
			if (code == CODE_BAD && first_bad == 0xFFFFFFFFu) {
				first_bad = i;
			}
			e.win[(u64)i * e.win_stride] = code;
		}
		if (wave_ballot(exotic)) {
			wc_valid = false;
			return false;
		}
		lanes_sync();
		// modified draft characters (substituted bases, changes of case)
		for (u32 i = 0; i < n_ov; i++) {
			const u32 op = e.ov_pos[i];
			if (op >= q && op - q < want) {
				e.win[(u64)(op - q) * e.win_stride] = char_code(e.ov_chr[i]);
			}
		}
		lanes_sync();
		// the first non-accepted character ends the stretch (the main loop skips over it, ntedit.cpp:2119-2138)
#if defined(__HIP_DEVICE_COMPILE__)
		if (W > 1) {
			for (u32 off = 32; off > 0; off >>= 1) {
				const u32 o = (u32)__shfl_xor((int)first_bad, (int)off, 64);
				first_bad = o < first_bad ? o : first_bad;
			}
		}
#endif
		if (first_bad != 0xFFFFFFFFu) {
			// lane l reads window bytes l .. l + k - 1 for its own k-mer
			const u32 lim = first_bad >= p.k ? first_bad - p.k + 1 : 0;
			if (lim < n) {
				n = lim;
			}
		}
		wc_valid = n_ov == 0;
		wc_pos0 = q;
		wc_len = want;
		*n_io = n;
		return n > 0;
	}

	// presence (members of the filter; solid_check as present_group) of up to 16 subset k-mers of a walk at once is not
	// needed here: the walks below go through subset_scan, which probes 8 at a time.

	// step 2 for one lane when the counts matter (counting filter / SNV mode, ntedit.cpp:1842-1861,1873,1890-1914):
	// missing subset k-mers, the subset k-mers that are there (check_there) and the median of their counts.
	// false = do_not_fix (a non-accepted character enters)
	NTE_HD bool
	lane_step2_counts(u8 draft_char, u32& check_missing, u32& check_there, u32& there_median) const
	{
		HashState ts = hs;
		u64 cw[4] = { 0, 0, 0, 0 }; // (lane_counts_fit(): at most LANE_COUNTS of them)
		check_missing = 0;
		check_there = 0;
		there_median = 0;
		const bool acgt = draft_char == 'A' || draft_char == 'C' || draft_char == 'G' || draft_char == 'T';
		const u32 lo = 1, hi = 255;
		(void)lo;
		(void)hi;
		u32 k = 0;
		bool ok = true;
		while (k < p.k && ok) {
			u64 b[8];
			u32 nb = 0;
			NTE_UNROLL
			for (int u = 0; u < 8; u++) {
				b[u] = 0;
				if (nb == (u32)u) {
					while (k < p.k) {
						const u8 in = win_i(k);
						hash_roll(ts, e.tab, win_o(k), in);
						if (in == CODE_BAD) {
							ok = false;
							break;
						}
						const bool is_sub = (k % p.jump) == 0;
						k++;
						if (is_sub) {
							b[u] = ts.fh + ts.rh;
							nb = (u32)u + 1;
							break;
						}
					}
				}
			}
			if (nb == 0) {
				break;
			}
			// (the k-mers gathered before a non-accepted character are counted as the serial loop counts them)
			u8 mn[8];
			count_group<8>(e.bloom, b, nb, mn);
			NTE_UNROLL
			for (int u = 0; u < 8; u++) {
				if ((u32)u < nb) {
					const u32 c = mn[u];
					if (c == 0) {
						check_missing++;
					} else if (acgt && c >= p.min_thr) {
						if (check_there < LANE_COUNTS) {
							const u64 bits = (u64)c << (8 * (check_there & 7));
							cw[0] |= check_there < 8 ? bits : 0;
							cw[1] |= check_there >= 8 && check_there < 16 ? bits : 0;
							cw[2] |= check_there >= 16 && check_there < 24 ? bits : 0;
							cw[3] |= check_there >= 24 ? bits : 0;
						}
						check_there++;
					}
				}
			}
		}
		// median (ntedit.cpp:455-463): element n/2 of the sorted counts
		const u32 nm = check_there;
		if (nm) {
			for (u32 i = 0; i < nm; i++) {
				const u32 vi = packed_byte(cw, i);
				u32 rank = 0;
				for (u32 j = 0; j < nm; j++) {
					const u32 vj = packed_byte(cw, j);
					rank += (vj < vi || (vj == vi && j < i)) ? 1u : 0u;
				}
				if (rank == nm / 2) {
					there_median = vi;
					break;
				}
			}
		}
		return ok;
	}

	// smallest counter of up to G k-mers (plain filter: 1 = contained, 0 = not), level by level
	template<int G>
	NTE_HD void
	count_group(const Filter& f, const u64 (&b)[G], u32 n, u8 (&mn)[G]) const
	{
		NTE_COUNT(probes, n);
		if (!fcounting(f)) {
			const u32 m = probe_group_range<G>(f, b, n, 1, 255);
			NTE_UNROLL
			for (int i = 0; i < G; i++) {
				mn[i] = (u8)((m >> i) & 1);
			}
			return;
		}
		u32 alive = (1u << n) - 1;
		NTE_UNROLL
		for (int i = 0; i < G; i++) {
			mn[i] = 255;
		}
		for (u32 h = 0; h < f.hash_num && alive; h++) {
			u8 byte[G];
			NTE_UNROLL
			for (int i = 0; i < G; i++) {
				byte[i] = 255;
				if ((alive >> i) & 1) {
					byte[i] = f.data[slot(f, hash_extend(b[i], p, h))];
					NTE_GATHER(1);
				}
			}
			NTE_UNROLL
			for (int i = 0; i < G; i++) {
				mn[i] = byte[i] < mn[i] ? byte[i] : mn[i];
				if (mn[i] == 0) {
					alive &= ~(1u << i);
				}
			}
		}
	}

	// The phases one lane runs by itself for the position whose k-mer starts at window offset win_off with hash hs.
	// dirty: the window holds a substituted base (the screening bitmap does not speak for these k-mers).
	// GATE (k_assess): the same answer, cheapest test first -- without a candidate whose own k-mer is there the position
	// cannot do anything (no substitution, no indel sweep -- its index base is such a candidate --, no upper-cased
	// revert, no -s 1 report), whatever step 2 says; one position in a hundred gets past that test
	template<bool GATE = false>
	NTE_HD u32
	assess_lane(u64 g_pos, bool dirty, u32 (&S)[4], u8& draft_char_out, u32& check_there, u32& there_median, bool& reverted)
	{
		S[0] = S[1] = S[2] = S[3] = 0;
		check_there = 0;
		there_median = 0;
		reverted = false;
		const u8 draft_code = win_o(p.k - 1);
		const u8 draft_char = code_letter(draft_code);
		draft_char_out = draft_char;
		// the main loop's test of the k-mer under the cursors (ntedit.cpp:1806)
		if (dirty && !snv() && !screen_absent(hs)) {
			return LANE_NONE;
		}
		// the substitution candidates (ntedit.cpp:1916-1934); their changed k-mers are probed together
		u8 cand[4];
		const u32 n_cand = candidate_bases(draft_char, snv(), cand);
		u64 cb[4];
		HashState cts[4];
		NTE_UNROLL
		for (int ci = 0; ci < 4; ci++) {
			cts[ci] = hs;
			cb[ci] = 0;
			if ((u32)ci < n_cand) {
				hash_changelast(cts[ci], e.tab, draft_code, char_code(cand[ci]));
				cb[ci] = cts[ci].fh + cts[ci].rh;
			}
		}
		u32 there = 0;
		const bool gate_first = GATE && mode() != 2 && !mask();
		if (gate_first) {
			there = n_cand ? present_group<4>(cb, n_cand, true) : 0;
			if (!there) {
				return LANE_NONE;
			}
		}
		// step 2 (ntedit.cpp:1826-1873)
		u32 check_missing = 0;
		if (counting() || snv()) {
			const bool ok = lane_step2_counts(draft_char, check_missing, check_there, there_median);
			if (!snv() && (!ok || !(check_missing >= p.thr_missing || (counting() && there_median < p.min_thr)))) {
				return LANE_NONE;
			}
		} else if (!dirty) {
			for (u32 k = 0; k < p.k; k++) {
				if (win_i(k) == CODE_BAD) {
					return LANE_NONE;
				}
			}
			const u64 g = g_pos + 1;
			for (u32 k = 0; k < p.k; k += p.jump) {
				check_missing += bit_absent(e.bitmap, g + k) ? 1u : 0u;
			}
			if (check_missing < p.thr_missing) {
				return LANE_NONE;
			}
		} else {
			const SubsetResult r =
			    subset_scan<8>(hs, 0, p.k - 1, false, false, 0, 0, p.thr_missing, [&](u32 k, HashState& t) {
				    const u8 in = win_i(k);
				    hash_roll(t, e.tab, win_o(k), in);
				    return in != CODE_BAD;
			    });
			if (r.aborted || r.gave_up || r.total - r.present < p.thr_missing) {
				return LANE_NONE;
			}
		}
		if (GATE && !gate_first) {
			return LANE_EDIT; // -m 2 sweeps every candidate whether its k-mer is there or not; -a masks the base
		}
		// step 3: the candidates' support
		if (!gate_first) {
			there = n_cand ? present_group<4>(cb, n_cand, true) : 0;
		}
		NTE_UNROLL
		for (int ci = 0; ci < 4; ci++) {
			if ((u32)ci < n_cand && ((there >> ci) & 1)) {
				reverted = true;
				const u8 sub_code = char_code(cand[ci]);
				const u32 last = p.k - 1;
				const SubsetResult r =
				    subset_scan<8>(cts[ci], 0, last, true, false, 0, p.thr_edit, 0, [&](u32 k, HashState& t) {
					    hash_roll(t, e.tab, k == last ? sub_code : win_o(k), win_i(k));
					    return true;
				    });
				S[ci] = 0x80000000u | (r.gave_up ? 0u : r.present);
			}
		}
		Best b;
		return decide_position(S, draft_char, check_there, there_median, false, b);
	}

	// what a position leaves behind even without an edit: bit 0 the upper-cased draft base of the substitution revert
	// (ntedit.cpp:1975-1981; where the draft byte differs), bit 1 -- -s 1 -- a position that keeps its base but may have
	// supported alternatives to report (only a candidate that reached the support bar can become one)
	NTE_HD u32
	lane_leftovers(u32 state, const u32 (&S)[4], bool reverted, u8 draft_byte, u8 draft_char) const
	{
		u32 out = (reverted && draft_byte != draft_char) ? 1u : 0u;
		if (snv() && state == LANE_NONE) {
			for (int ci = 0; ci < 4; ci++) {
				if ((S[ci] >> 31) && (S[ci] & 0x7FFFFFFFu) >= p.thr_edit) {
					out |= 2u;
				}
			}
		}
		return out;
	}

	// k_assess / its host twin: can the clean-state assessment of the position whose k-mer starts at window offset
	// win_off (hash in hs) do anything at all?  The window holds k + win_len_in() accepted codes from there on;
	// draft_byte = the draft's own byte under the last base of the k-mer.
	NTE_HD bool
	assess_gate(u64 g_pos, u8 draft_byte)
	{
		u32 S[4];
		u8 dc = 0;
		u32 ct = 0, med = 0;
		bool rev = false;
		const u32 st = assess_lane<true>(g_pos, false, S, dc, ct, med, rev);
		return st != LANE_NONE || lane_leftovers(st, S, rev, draft_byte, dc) != 0;
	}

	// hash of the k-mer at window offset win_off (seed of ntedit.cpp:412-413 on the window codes)
	NTE_HD HashState
	seed_from_window() const
	{
		HashState s;
		s.fh = 0;
		s.rh = 0;
		for (u32 i = 0; i < p.k; i++) {
			s.fh = srol1(s.fh) ^ tab_f(e.tab, win_o(i));
		}
		for (u32 i = p.k; i > 0; i--) {
			s.rh = srol1(s.rh) ^ tab_r(e.tab, win_o(i - 1));
		}
		return s;
	}

	// One batch of positions from the cursors on.  false: nothing was done (the window cannot be used here), the
	// caller assesses this position the serial way.  true: the machine stands at the last position the batch dealt
	// with -- the one whose edit it applied (changed_seq), or the last of a stretch without any -- ready to roll on;
	// walked = positions it went through.
	NTE_HD bool
	run_lanes(u32& walked)
	{
		const u32 q = h_seq_i;
		NTE_PROF_SUB(7); // (time outside)
		u32 n = lanes_run_length(q);
		if (n == 0 || !lanes_fill_window(q, &n)) {
			return false;
		}
		NTE_PROF_SUB(0); // run length + window
		NTE_PROF_ADD(2, 1);
		NTE_PROF_ADD(3, n);
		win_ok = true;
		NTE_COUNT(lane_batches, 1);
		NTE_COUNT(lane_positions, n);
		PerLane<u64> l_fh, l_rh;
		PerLane<u32> l_state, l_s0, l_s1, l_s2, l_s3, l_there, l_median, l_out;
		const HashState keep_hs = hs;
		NTE_FOR_LANES(l, N_LANES)
		{
			l_state.at(l) = LANE_NONE;
			l_out.at(l) = 0;
			l_fh.at(l) = l_rh.at(l) = 0;
			l_s0.at(l) = l_s1.at(l) = l_s2.at(l) = l_s3.at(l) = 0;
			l_there.at(l) = l_median.at(l) = 0;
		}
		NTE_FOR_LANES(l, n)
		{
			win_off = l;
			const HashState s = seed_from_window();
			hs = s;
			u32 S[4];
			u8 dc = 0;
			u32 ct = 0, med = 0;
			bool rev = false;
			const u64 P = (u64)q + l;
			const u32 st = assess_lane(e.gbase + P, (int64_t)P <= last_sub_pos, S, dc, ct, med, rev);
			l_fh.at(l) = s.fh;
			l_rh.at(l) = s.rh;
			l_state.at(l) = st;
			l_s0.at(l) = S[0];
			l_s1.at(l) = S[1];
			l_s2.at(l) = S[2];
			l_s3.at(l) = S[3];
			l_there.at(l) = ct;
			l_median.at(l) = med;
			l_out.at(l) = lane_leftovers(st, S, rev, e.seq[P + p.k - 1], dc) | ((u32)dc << 8);
		}
		hs = keep_hs;
		NTE_PROF_SUB(2); // the lanes' own phases
		// ---- serial order: the first position that is not a plain "no edit"
		u32 f = n; // lane of the edit (n = none)
		Best b;
		b.edit_type = 0;
		u32 from = 0;
		while (from < n) {
			const u32 c = lanes_first(l_state, from, n);
			if (c >= n) {
				break;
			}
			u32 S[4];
			S[0] = lanes_get(l_s0, c);
			S[1] = lanes_get(l_s1, c);
			S[2] = lanes_get(l_s2, c);
			S[3] = lanes_get(l_s3, c);
			const u8 dc = (u8)(lanes_get(l_out, c) >> 8);
			hs.fh = lanes_get(l_fh, c);
			hs.rh = lanes_get(l_rh, c);
			win_off = c;
			const u32 st = decide_position(S, dc, lanes_get(l_there, c), lanes_get(l_median, c), true, b);
			if (st == LANE_EDIT) {
				f = c;
				break;
			}
			from = c + 1; // its sweeps found nothing
		}
		NTE_PROF_SUB(4); // replay + indel sweeps
		// ---- what the positions in front of the edit (and the edit's own revert) leave behind, in order
		const u32 upto = f < n ? f + 1 : n;
		u32 from_out = 0;
		while (from_out < upto) {
			PerLane<u32> flag;
			NTE_FOR_LANES(l, N_LANES) { flag.at(l) = l_out.at(l) & 3u; }
			const u32 c = lanes_first(flag, from_out, upto);
			if (c >= upto) {
				break;
			}
			const u32 o = lanes_get(l_out, c);
			const u8 dc = (u8)(o >> 8);
			if (o & 1u) {
				set_seq(q + c + p.k - 1, dc);
			}
			if ((o & 2u) && c != f) {
				u32 S[4];
				S[0] = lanes_get(l_s0, c);
				S[1] = lanes_get(l_s1, c);
				S[2] = lanes_get(l_s2, c);
				S[3] = lanes_get(l_s3, c);
				Best nb;
				decide_position(S, dc, lanes_get(l_there, c), lanes_get(l_median, c), false, nb);
				if (nb.edit_type == 0 && nb.altsupp1) {
					t_seq_i = q + c + p.k - 1; // (make_edit reports the tail position)
					make_edit(dc, nb);
				}
			}
			from_out = c + 1;
		}
		// ---- place the machine
		const u32 at = f < n ? f : n - 1;
		h_seq_i = q + at;
		t_seq_i = q + at + p.k - 1;
		hs.fh = lanes_get(l_fh, at);
		hs.rh = lanes_get(l_rh, at);
		win_off = at;
		walked = at + 1;
		NTE_COUNT(lane_walked, walked);
		changed_seq = false;
		if (f < n) {
			NTE_COUNT(lane_edits, 1);
			changed_seq = true;
			make_edit((u8)(lanes_get(l_out, f) >> 8), b);
		}
		NTE_PROF_SUB(3); // what the positions leave behind, placing the machine, applying
		return true;
	}

	// run one event that starts (clean) with its k-mer head at local position start
	// LANES: runs of failing positions go through run_lanes() (the wavefront-per-event kernel; the host build)
	template<bool LANES = false>
	NTE_HD void
	run(u32 start, u32& cover_end)
	{
		h_seq_i = start;
		t_seq_i = start + p.k - 1;
		h_node = t_node = 0;
		nbase = 0;
		nsize = 0;
		rope_touched = false;
		n_ov = 0;
		tmp_on = false;
		tmp_pos = 0;
		tmp_chr = 0;
		last_sub_pos = -1;
		win_ok = false;
		wc_valid = false;
		wc_pos0 = 0;
		wc_len = 0;
		win_off = 0;
		la_mask = la_n = la_i = 0;
		la_off = false;
		la_win = false;
		la_known = la_known_vals = 0;
		la_known_pos = NONE32;
		changed_seq = false;
		first_chunk = cur_chunk = NONE32;
		fill = 0;
		flags = 0;
		cover_end = start;

		Node root;
		root.type = 0;
		root.s_pos = 0;
		root.e_pos = e.len - 1;
		root.c = 0;
		root.support = 0;
		nput(0, root);

		// seed hash of the k-mer at start (ntedit.cpp:1778; all bases accepted)
		hs.fh = 0;
		hs.rh = 0;
		for (u32 i = 0; i < p.k; i++) {
			u8 code = char_code(e.seq[start + i]);
			hs.fh = srol1(hs.fh) ^ tab_f(e.tab, code);
		}
		for (u32 i = p.k; i > 0; i--) {
			u8 code = char_code(e.seq[start + i - 1]);
			hs.rh = srol1(hs.rh) ^ tab_r(e.tab, code);
		}
		u8 char_in = e.seq[t_seq_i];
		u8 char_out = 0;

		if (p.debug_stop == 1) {
			cover_end = e.len;
			return;
		}
		if ((u64)start + p.k == e.len) {
			// findFirstAcceptedKmer stops at i + k < size (ntedit.cpp:527): the reference never
			// SEEDS its main loop at the last k-mer start of a contig, it only gets there by
			// rolling on from an earlier accepted k-mer.  Without one this position is never
			// looked at (visible with -s 1, where every position is assessed).
			u32 good = p.k - 1; // start .. start+k-2 are accepted bases
			bool earlier = false;
			for (u32 i = start; i > 0;) {
				i--;
				good = char_code(e.seq[i]) != CODE_BAD ? good + 1 : 0;
				if (good >= p.k) {
					earlier = true;
					break;
				}
			}
			if (!earlier) {
				flags |= EV_TERMINAL;
				cover_end = e.len;
				return;
			}
		}
		bool first = true;
		u32 steps = 0;
		NTE_PROF_DECL;
		while (true) {
			NTE_PROF(first ? 0 : 5); // 0 = seeding, 5 = loop overhead
			if (p.event_budget && ++steps > p.event_budget && cur_chunk != NONE32) {
				// A run that does not come back to a clean state for this long is (almost
				// always) one the serial order will discard: park it.  The host re-runs it
				// without a budget if it turns out to be applied.
				flags |= EV_UNFINISHED;
				cover_end = e.len;
				break;
			}
			if ((u64)h_seq_i + p.k - 1 >= e.len) {
				flags |= EV_TERMINAL;
				cover_end = e.len;
				break;
			}
			if (flags & (EV_OVERFLOW | EV_ARENA_FULL | EV_DEFERRED)) {
				cover_end = e.len;
				break;
			}
			bool missing;
			const bool clean = is_clean();
			if (clean) {
				u64 g = e.gbase + h_seq_i;
				if (!first && (!bit_absent(e.runmap, g) || is_event_start(e.runmap, g, p.start_grid))) {
					cover_end = h_seq_i;
					break;
				}
			}
			bool in_lanes = false;
			if (LANES && p.lanes && mode() == 0 && !mask() && !p.debug_stop && e.win && lane_counts_fit() && (clean || (p.lanes > 1 && linear()))) {
				// this position and the ones behind it, one per lane
				u32 walked = 0;
				in_lanes = run_lanes(walked);
				if (in_lanes) {
					steps += walked - 1;
					la_n = la_i = 0;
					la_off = false;
					la_win = false;
				}
			}
			const bool was_first = first;
			first = false;
			if (in_lanes) {
				missing = true;
				NTE_PROF(was_first ? 2 : 3);
				NTE_PROF_COUNT(0);
				NTE_PROF_COUNT(1);
			} else {
			if (clean) {
				missing = true; // clean state: the screening bitmap already answered
			} else {
				if (snv()) {
					missing = true;
				} else {
					if (la_i >= la_n && !la_off) {
						build_lookahead();
					}
					if (la_i < la_n) {
						missing = !((la_mask >> la_i) & 1);
					} else {
						missing = screen_absent(hs);
					}
				}
			}
			NTE_PROF(1); // presence of the k-mer at the cursor (look-ahead included)
			NTE_PROF_COUNT(0);
			if (missing) {
				NTE_PROF_COUNT(1);
				changed_seq = false;
				process_missing(char_in);
				NTE_PROF(was_first ? 2 : 3); // first / later failing positions
				la_win = false; // (the failing position filled the window for itself)
				if (changed_seq) {
					la_n = la_i = 0; // the sequence has changed: look ahead afresh
					la_off = false;
				}
				// (an error nothing fixes fails at every k-mer that covers it: the k-mers ahead are the same ones)
				if (!LANES && e.defer_sweeps && p.defer_run && !changed_seq && clean && mode() == 0 && !mask() && p.lanes && lane_counts_fit() &&
				    !(flags & EV_DEFERRED)) {
					// A clean position that ends without an edit: if the absent run goes on, the following positions
					// are assessed one per lane by the wavefront-per-event launch instead of one after the other here.
					u32 more = 0;
					const u64 g1 = e.gbase + h_seq_i + 1;
					while (more < p.defer_run && bit_absent(e.runmap, g1 + more) && !is_event_start(e.runmap, g1 + more, p.start_grid)) {
						more++;
					}
					if (more >= p.defer_run) {
						flags |= EV_DEFERRED;
					}
				}
			}
			}
			if (p.debug_stop >= 2 && p.debug_stop < 8) {
				cover_end = e.len;
				break;
			}
			if (p.debug_stop >= 8 && was_first) {
				// 8: keep only events whose first position made an edit; 16: only the others
				const bool edited = last_sub_pos >= 0 || rope_touched;
				if ((p.debug_stop == 8 && !edited) || (p.debug_stop == 16 && edited)) {
					cover_end = e.len;
					break;
				}
			}
			// Behind an edit the next k-1 k-mers hold the new base(s); they were probed together (look-ahead)
			// and, as a rule, are all there: the reference rolls through them one position at a time doing
			// nothing.  That stretch of the walk is taken in one stride: the cursors are rolled without reading
			// characters (increment() only: rope nodes, no draft bytes), as long as the next position is
			// looked-ahead, present, inside the contig and still dirty -- where the machine would be clean the
			// main loop has to consult the screening bitmap itself -- and the hash is rolled from the window codes
			// the look-ahead was hashed from.
			NTE_PROF_SUB(7);
			if (!missing && la_i < la_n && la_win && !is_clean()) {
				u32 J = 0;
				u32 room = 0xFFFFFFFFu;
				if (p.event_budget) {
					room = steps < p.event_budget ? p.event_budget - steps : 0;
				}
				// (the nodes under the two cursors are kept in registers: a roll inside a node costs no rope access)
				Node nh = nget(h_node), nt = nget(t_node);
				u32 hs2 = h_seq_i, ts2 = t_seq_i, hn2 = h_node, tn2 = t_node;
				while (la_i + J + 1 < la_n && ((la_mask >> (la_i + J + 1)) & 1) && J < room) {
					// one roll of the cursors (roll(), ntedit.cpp:1216-1247) on copies
					u32 hs3 = hs2, ts3 = ts2, hn3 = hn2, tn3 = tn2;
					Node nh3 = nh, nt3 = nt;
					if (hs3 >= e.len || hn3 >= nsize) {
						break;
					}
					increment_cached(hs3, hn3, nh3);
					if (ts3 >= e.len || tn3 >= nsize) {
						break;
					}
					increment_cached(ts3, tn3, nt3);
					if (ts3 >= e.len || tn3 >= nsize || (u64)hs3 + p.k - 1 >= e.len) {
						break;
					}
					if (clean_at(hs3, ts3, hn3, tn3, nt3)) {
						break; // where the machine would be clean the main loop consults the screening bitmap itself
					}
					hs2 = hs3;
					ts2 = ts3;
					hn2 = hn3;
					tn2 = tn3;
					nh = nh3;
					nt = nt3;
					J++;
				}
				if (J) {
					h_seq_i = hs2;
					t_seq_i = ts2;
					h_node = hn2;
					t_node = tn2;
					for (u32 q = 0; q < J; q++) {
						hash_roll(hs, e.tab, win_o(la_i + q), win_i(la_i + q));
					}
					la_i += J;
					steps += J;
					char_in = get_character(t_seq_i, nget(t_node));
				}
			}
			NTE_PROF_SUB(5); // stride
			// advance; skip over k-mers containing a non-accepted base (ntedit.cpp:2119-2138)
			bool ended = false;
			int64_t target = -1;
			do {
				if (roll(h_seq_i, t_seq_i, h_node, t_node, char_out, char_in)) {
					la_i++;
					if (char_code(char_in) == CODE_BAD) {
						target = (int64_t)t_seq_i + (int64_t)p.k;
					}
					roll_hash(hs, char_out, char_in);
				} else {
					ended = true;
					break;
				}
			} while (target >= 0 && (int64_t)t_seq_i != target);
			if (ended) {
				flags |= EV_TERMINAL;
				cover_end = e.len;
				break;
			}
			NTE_PROF_SUB(6); // roll
			NTE_PROF(4); // advance
			housekeeping();
			NTE_PROF(6);
		}
		NTE_PROF(5);

		// stream out what is left of the rope (only if an indel touched it)
		if (rope_touched && !(flags & (EV_DEFERRED | EV_UNFINISHED))) {
			for (u32 i = nbase; i < nsize; i++) {
				Node n = nget(i);
				if (n.type == -1) {
					break; // unset slots behind the rope are spare capacity, not content
				}
				emit_node(n);
			}
		}
		NTE_PROF(7); // rope flush
		NTE_PROF_FLUSH;
	}

	// seal the chunk chain; returns the first chunk (NONE32 if nothing was emitted)
	NTE_HD u32
	finish(u32 start, u32 cover_end)
	{
		if (flags & (EV_OVERFLOW | EV_ARENA_FULL | EV_DEFERRED)) {
			return NONE32;
		}
		if (cur_chunk == NONE32) {
			return NONE32;
		}
		Item link;
		link.w[0] = NONE32;
		link.w[1] = fill;
		link.w[2] = link.w[3] = 0;
		e.arena[(u64)cur_chunk * CHUNK_ITEMS] = link;
		Item hdr;
		hdr.w[0] = e.contig;
		hdr.w[1] = start;
		hdr.w[2] = cover_end;
		hdr.w[3] = flags;
		e.arena[(u64)first_chunk * CHUNK_ITEMS + 1] = hdr;
		return first_chunk;
	}
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
