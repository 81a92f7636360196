// nte_common.h -- shared definitions for the MI355X ntEdit hot path:
// ntHash2 arithmetic, Bloom-filter probe, device-side parameter block and the
// event-record wire format.  Compiles under hipcc (device + host) and under a
// plain host compiler (used by the CPU-side logic tests of the event machine).
//
// Reference behaviour restated here (never copied):
//   btllib hashing_internals (call sites ntedit.cpp:412-415,428-431,444-451)
//   btllib KmerBloomFilter::contains (call site ntedit.cpp:368-371)
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#if defined(NTE_SOFT_INLINE)
#define NTE_HD __host__ __device__ inline // code-size experiments: let the compiler decide
#else
#define NTE_HD __host__ __device__ __forceinline__
#endif
#define NTE_UNROLL _Pragma("unroll")
#else
#define NTE_HD inline
#define NTE_UNROLL
#endif

namespace nte {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

// ---------------------------------------------------------------- constants
// ntHash2 seeds; SEED_N = 0.
constexpr u64 SEED_A = 0x3c8bfbb395c60474ULL;
constexpr u64 SEED_C = 0x3193c18562a02b4cULL;
constexpr u64 SEED_G = 0x20323ed082572324ULL;
constexpr u64 SEED_T = 0x295549f54be24456ULL;
constexpr u64 MULTISEED = 0x90b45d39fb6da1faULL;
constexpr unsigned MULTISHIFT = 27;
constexpr unsigned MAX_HASHES = 8;
// Bounds of the fixed-size per-lane arrays of the event machine (nte_machine_*.inc), checked where the parameters are
// folded (host/params.cpp: make_dev_params refuses anything beyond them) and asserted where the arrays are declared:
constexpr unsigned MAX_INSERTION = 5;  // -i (ntedit.cpp:2485-2488): inserted bases of a candidate
constexpr unsigned MAX_DELETION = 10;  // -d (ntedit.cpp:2489-2493)
constexpr unsigned MAX_CANDIDATES = 4; // substitution candidates of a position (ntedit.cpp:190-199: at most "ATCG")
constexpr unsigned INDEL_BYTES = 12;   // u8 ins[] / deleted[] / Best::indel: index base + inserted bases, or the deleted bases
static_assert(MAX_INSERTION + 1 <= INDEL_BYTES && MAX_DELETION <= INDEL_BYTES, "indel buffers hold the longest candidate");

// Character classes.  Every byte of the draft is mapped to a 4-bit code:
//   0..3   A C G T (either case)
//   4..13  the other "accepted" IUPAC codes R Y S W K M B D H V
//          (isAcceptedBase, ntedit.cpp:493-499; case-insensitive via toupper)
//   15     anything else (N, U, separators): breaks every k-mer it touches
constexpr u8 CODE_BAD = 15;

NTE_HD u8
char_code(u8 c)
{
	switch (c & 0xDF) { // folds a-z onto A-Z; no other byte lands on a letter
	case 'A':
		return 0;
	case 'C':
		return 1;
	case 'G':
		return 2;
	case 'T':
		return 3;
	case 'R':
		return 4;
	case 'Y':
		return 5;
	case 'S':
		return 6;
	case 'W':
		return 7;
	case 'K':
		return 8;
	case 'M':
		return 9;
	case 'B':
		return 10;
	case 'D':
		return 11;
	case 'H':
		return 12;
	case 'V':
		return 13;
	default:
		return CODE_BAD;
	}
}

// Forward seed of a code = SEED_TAB[letter]; only ACGT are non-zero.
NTE_HD u64
seed_fwd_of_code(u8 code)
{
	switch (code) {
	case 0:
		return SEED_A;
	case 1:
		return SEED_C;
	case 2:
		return SEED_G;
	case 3:
		return SEED_T;
	default:
		return 0;
	}
}

// Reverse-strand seed = SEED_TAB[letter & 7] (CP_OFF slots 1,3,4,5,7 hold the
// complement seeds).  For the IUPAC letters the slot is whatever (c & 7)
// lands on: Y->1(T) S->3(G) W->7(C) K->3(G) M->5(A) D->4(A), R/B/H/V -> 0.
NTE_HD u64
seed_rev_of_code(u8 code)
{
	switch (code) {
	case 0: // A -> T
		return SEED_T;
	case 1: // C -> G
		return SEED_G;
	case 2: // G -> C
		return SEED_C;
	case 3: // T -> A
		return SEED_A;
	case 5: // Y (0x59 & 7 = 1)
		return SEED_T;
	case 6: // S (0x53 & 7 = 3)
		return SEED_G;
	case 7: // W (0x57 & 7 = 7)
		return SEED_C;
	case 8: // K (0x4B & 7 = 3)
		return SEED_G;
	case 9: // M (0x4D & 7 = 5)
		return SEED_A;
	case 11: // D (0x44 & 7 = 4)
		return SEED_A;
	default: // R B H V and BAD
		return 0;
	}
}

// Seeds by raw byte, exactly SEED_TAB[c] / SEED_TAB[c & 7]: also defined for characters that
// are not accepted bases.  U/u hash like T, and a handful of other bytes pick up a
// complement-slot seed on the reverse strand through (c & 7).  The 4-bit codes above fold all
// of those into CODE_BAD (seed 0), which is exact for N/n and for separators; the raw forms
// are used wherever a k-mer that contains such a byte can actually be hashed.
NTE_HD u64
seed_fwd_raw(u8 c)
{
	switch (c) {
	case 'A':
	case 'a':
		return SEED_A;
	case 'C':
	case 'c':
		return SEED_C;
	case 'G':
	case 'g':
		return SEED_G;
	case 'T':
	case 't':
	case 'U':
	case 'u':
		return SEED_T;
	case 1:
		return SEED_T;
	case 3:
		return SEED_G;
	case 4:
	case 5:
		return SEED_A;
	case 7:
		return SEED_C;
	default:
		return 0;
	}
}

NTE_HD u64
seed_rev_raw(u8 c)
{
	switch (c & 7) {
	case 1:
		return SEED_T;
	case 3:
		return SEED_G;
	case 4:
	case 5:
		return SEED_A;
	case 7:
		return SEED_C;
	default:
		return 0;
	}
}

// a byte whose 4-bit code (CODE_BAD -> zero seeds) would NOT reproduce its raw seeds
NTE_HD bool
is_exotic(u8 c)
{
	return char_code(c) == CODE_BAD && (seed_fwd_raw(c) != 0 || seed_rev_raw(c) != 0);
}

// ---------------------------------------------------------------- rotations
// ntHash2 "split rotate": bits 0..32 and bits 33..63 rotate independently.
NTE_HD u64
srol1(u64 x)
{
	u64 m = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
	return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | m;
}

NTE_HD u64
sror1(u64 x)
{
	u64 m = ((x & 0x200000000ULL) << 30) | ((x & 1ULL) << 32);
	return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | m;
}

NTE_HD u64
sroln(u64 x, unsigned d)
{
	const u64 lo_mask = 0x1FFFFFFFFULL;
	u64 lo = x & lo_mask;
	u64 hi = x >> 33;
	unsigned dl = d % 33, dh = d % 31;
	if (dl) {
		lo = ((lo << dl) | (lo >> (33 - dl))) & lo_mask;
	}
	if (dh) {
		hi = ((hi << dh) | (hi >> (31 - dh))) & 0x7FFFFFFFULL;
	}
	return (hi << 33) | lo;
}

// ------------------------------------------------------------- seed tables
// Five 16-entry tables indexed by character code, rebuilt per k:
//   F   forward seed                      (enters the forward hash)
//   FK  forward seed rotated by k         (leaves the forward hash)
//   R   reverse seed                      (leaves the reverse hash)
//   RK  reverse seed rotated by k         (enters the reverse hash)
//   RK1 reverse seed rotated by k-1       (swap of the LAST base, ntedit.cpp:434-452)
// Interleaved by use: a roll needs {F, RK} of the base that enters and {FK, R} of the base that leaves, so each pair
// sits in 16 consecutive bytes -- two 16-byte LDS reads per roll instead of four 8-byte ones.
enum
{
	TAB_IN = 0,   // [2c] = F, [2c + 1] = RK
	TAB_OUT = 32, // [2c] = FK, [2c + 1] = R
	TAB_RK1 = 64,
	TAB_WORDS = 80
};

NTE_HD u64
tab_f(const u64* tab, u8 c)
{
	return tab[TAB_IN + 2 * c];
}

NTE_HD u64
tab_rk(const u64* tab, u8 c)
{
	return tab[TAB_IN + 2 * c + 1];
}

NTE_HD u64
tab_fk(const u64* tab, u8 c)
{
	return tab[TAB_OUT + 2 * c];
}

NTE_HD u64
tab_r(const u64* tab, u8 c)
{
	return tab[TAB_OUT + 2 * c + 1];
}

inline void
build_seed_tables(unsigned k, u64* tab)
{
	for (u8 c = 0; c < 16; c++) {
		u64 f = seed_fwd_of_code(c), r = seed_rev_of_code(c);
		tab[TAB_IN + 2 * c] = f;
		tab[TAB_IN + 2 * c + 1] = sroln(r, k);
		tab[TAB_OUT + 2 * c] = sroln(f, k);
		tab[TAB_OUT + 2 * c + 1] = r;
		tab[TAB_RK1 + c] = sroln(r, k - 1);
	}
}

struct HashState
{
	u64 fh, rh;
};

// roll one base: `out` leaves at the head, `in` enters at the tail
NTE_HD void
hash_roll(HashState& s, const u64* tab, u8 out, u8 in)
{
	s.fh = srol1(s.fh) ^ tab_f(tab, in) ^ tab_fk(tab, out);
	s.rh = sror1(s.rh ^ tab_rk(tab, in) ^ tab_r(tab, out));
}

// replace the last base of the current k-mer
NTE_HD void
hash_changelast(HashState& s, const u64* tab, u8 out, u8 in)
{
	s.fh ^= tab_f(tab, out) ^ tab_f(tab, in);
	s.rh ^= tab[TAB_RK1 + out] ^ tab[TAB_RK1 + in];
}

// raw-byte forms of the two updates (same arithmetic, seeds from the bytes themselves)
NTE_HD void
hash_roll_raw(HashState& s, unsigned k, u8 out, u8 in)
{
	s.fh = srol1(s.fh) ^ seed_fwd_raw(in) ^ sroln(seed_fwd_raw(out), k);
	s.rh = sror1(s.rh ^ sroln(seed_rev_raw(in), k) ^ seed_rev_raw(out));
}

NTE_HD void
hash_changelast_raw(HashState& s, unsigned k, u8 out, u8 in)
{
	s.fh ^= seed_fwd_raw(out) ^ seed_fwd_raw(in);
	s.rh ^= sroln(seed_rev_raw(out), k - 1) ^ sroln(seed_rev_raw(in), k - 1);
}

// --------------------------------------------------------------- the filter
struct Filter
{
	const u8* data; // plain BF: bit array, LSB-first within a byte; counting BF: 8-bit counters
	u64 bits;       // number of addressable slots: bits (plain) or counters (counting)
	u64 mask;       // bits - 1 when bits is a power of two, else 0
	u64 magic;      // floor(2^64 / bits) when bits is not a power of two (filter_set_size)
	u32 hash_num;
	u32 counting;   // 1 = btllib KmerCountingBloomFilter8 (contains() = min counter)
	u32 magic32;    // the same reciprocal when it fits 32 bits (2^32 < bits < 2^40, not a power of two): filter_slot's short form
	u32 pad_;
};

struct DevParams
{
	u32 k, h;
	u32 jump;
	u32 ins_tries;     // num_tries[max_insertions] (ntedit.cpp:172,1587)
	u32 max_deletions;
	u32 mode, mask, secbf;
	u32 insertion_cap;
	// the float comparisons of ntedit.cpp:1531-1535,1659-1663,1867-1872,1992-1997
	// folded on the host into "count >= integer" thresholds
	u32 thr_missing, thr_edit, thr_edit_del;
	u32 start_grid;    // extra event start every start_grid positions inside an absent run
	u32 node_window;   // live rope nodes kept per event thread
	u32 event_budget;  // positions an event may walk before it is parked as EV_UNFINISHED (0 = no limit)
	u32 inline_tries;  // a launch that postpones indel sweeps still tries this many candidates of one itself (see nte_machine.h)
	u32 debug_stop;    // timing ablations only (NTEDIT_HIP_MACHINE_DEBUG): 1 seed, 2 step 2, 4 first position
	u32 counting;      // primary filter is a counting filter
	u32 snv;           // -s 1: every position is re-assessed (ntedit.cpp:1806,1865)
	u32 min_thr, max_thr; // -p / -q (ntedit.cpp:131-132); only meaningful with a counting filter
	u32 lanes;         // wavefront-per-event launch: runs of failing positions are assessed one position per lane
	                   // (0 off, 1 in the clean state, 2 also while the window holds substituted bases; nte_machine.h)
	u32 defer_run;     // thread-per-event launch: an event whose clean position ends without an edit and whose absent run
	                   // goes on for at least this many positions is handed to the wavefront-per-event launch (0 = never)
	u32 defer_fail;    // thread-per-event launch: an event that has gone through this many failing positions (a chain of
	                   // edits in repetitive sequence: one lane at work, 63 waiting for it) goes there as well (0 = never)
	u32 defer_fail_snv; // the same with -s 1, where every position an event goes through counts (0 = never)
	u64 mul[MAX_HASHES]; // mul[i] = i ^ (k * MULTISEED), i >= 1
};

NTE_HD u64
hash_extend(u64 base, const DevParams& p, unsigned i)
{
	if (i == 0) {
		return base;
	}
	u64 t = base * p.mul[i];
	return t ^ (t >> MULTISHIFT);
}

// high 64 bits of a 64 x 64 product
NTE_HD u64
mulhi64(u64 a, u64 b)
{
#if defined(__HIP_DEVICE_COMPILE__)
	return __umul64hi(a, b);
#else
	return (u64)(((unsigned __int128)a * (unsigned __int128)b) >> 64);
#endif
}

// geometry of a filter with `slots` addressable bits / counters
inline void
filter_set_size(Filter& f, u64 slots)
{
	f.bits = slots;
	const bool pow2 = slots && (slots & (slots - 1)) == 0;
	f.mask = pow2 ? slots - 1 : 0;
	// floor(2^64 / slots) == floor((2^64 - 1) / slots) unless slots divides 2^64
	f.magic = (slots && !pow2) ? 0xFFFFFFFFFFFFFFFFULL / slots : 0;
	f.magic32 = (!pow2 && slots > (1ULL << 32) && slots < (1ULL << 40)) ? (u32)f.magic : 0;
	f.pad_ = 0;
}

NTE_HD u64
filter_slot(const Filter& f, u64 hv)
{
	if (f.mask) {
		return hv & f.mask;
	}
	if (f.magic32) {
		// Filters of 2^32 .. 2^40 slots (the reference tool's own size for a human genome: 3.7e10 bits): the quotient from
		// the HIGH words alone.  With H = hv >> 32 and M = floor(2^64 / bits) < 2^32, q' = floor(H * M / 2^32) falls short
		// of floor(hv / bits) by at most 2 (H * frac(2^64 / bits) / 2^32 < 1 and (hv mod 2^32) / bits < 1), so
		// hv - q' * bits < 3 * bits and two conditional subtractions finish it: four 32-bit multiplications where the
		// general form below takes eleven (a 64 x 64 -> 128 high half and a 64 x 64 product; 32-bit multiplies are
		// quarter rate, and the partition kernel does three of these per k-mer: 43.8 -> 4x.x ms per 3 Gbp at 4.64 GB).
		const u32 H = (u32)(hv >> 32);
		const u32 q = (u32)(((u64)H * f.magic32) >> 32);
		const u32 ml = (u32)f.bits, mh = (u32)(f.bits >> 32);
		const u64 qm = (u64)q * ml + ((u64)(q * mh) << 32); // (q * bits <= hv < 2^64: q * mh < 2^32)
		u64 r = hv - qm;
		if (r >= f.bits) {
			r -= f.bits;
		}
		if (r >= f.bits) {
			r -= f.bits;
		}
		return r;
	}
	// hv % bits without a division: q = floor(hv * magic / 2^64) is floor(hv / bits) or one
	// less (magic > 2^64 / bits - 1), so one conditional subtraction finishes it
	const u64 q = mulhi64(hv, f.magic);
	u64 r = hv - q * f.bits;
	if (r >= f.bits) {
		r -= f.bits;
	}
	return r;
}

// counting filter: contains() returns the smallest of the h counters
// (btllib CountingBloomFilter::contains; BFWrapper::get_count, ntedit.cpp:373-376)
NTE_HD u32
filter_min_count(const Filter& f, const DevParams& p, u64 base)
{
	u32 mn = 255;
	for (unsigned i = 0; i < f.hash_num; i++) {
		const u32 c = f.data[filter_slot(f, hash_extend(base, p, i))];
		mn = c < mn ? c : mn;
		if (mn == 0) {
			break;
		}
	}
	return mn;
}

// contains(): AND of hash_num bits, early exit on the first zero (plain);
// min counter > 0 (counting; BFWrapper::contains, ntedit.cpp:368-371)
NTE_HD bool
filter_contains(const Filter& f, const DevParams& p, const HashState& s)
{
	u64 base = s.fh + s.rh;
	if (f.counting) {
		return filter_min_count(f, p, base) > 0;
	}
	for (unsigned i = 0; i < f.hash_num; i++) {
		u64 n = filter_slot(f, hash_extend(base, p, i));
		if (!((f.data[n >> 3] >> (n & 7)) & 1)) {
			return false;
		}
	}
	return true;
}

// the main loop's test of a k-mer (ntedit.cpp:1806): not contained, or -- counting filter --
// seen fewer than min_thr (-p) times
NTE_HD bool
filter_screen_absent(const Filter& f, const DevParams& p, const HashState& s)
{
	if (f.counting) {
		const u32 c = filter_min_count(f, p, s.fh + s.rh);
		return c == 0 || c < p.min_thr;
	}
	return !filter_contains(f, p, s);
}

// ------------------------------------------------------- event wire format
// Event threads stream 16-byte items into 128-byte chunks of a global arena.
// chunk = 8 items; item 0 is the chunk link {next, n_items, -, -}; the first
// chunk of an event carries the event header in item 1.
constexpr u32 CHUNK_ITEMS = 8;
constexpr u32 NONE32 = 0xFFFFFFFFu;

enum ItemTag : u32
{
	TAG_NODE = 1,
	TAG_SUB = 2,
	TAG_MOD = 3,
	TAG_HDR = 4
};

enum EventFlags : u32
{
	EV_TERMINAL = 1,  // ran to the end of the contig
	EV_OVERFLOW = 2,  // ran out of node window: results invalid (retry with a larger window)
	EV_ARENA_FULL = 4, // output arena exhausted: results invalid (retry with a larger arena)
	EV_DEFERRED = 8,   // needs an indel sweep and the launch asked to postpone those (pass 1)
	EV_UNFINISHED = 16 // walked more positions than the launch's budget: what it emitted is void, its
	                   // cover_end is the contig end; re-run without a budget if it turns out to be applied
};

struct Item
{
	u32 w[4];
};

// rope node (ntedit.cpp:613-620), packed
struct Node
{
	u32 s_pos, e_pos;
	u16 support;
	int8_t type; // -1 unset, 0 position range, 1 character
	u8 c;
};

} // namespace nte
