// nte_kernels.hip -- gfx950 kernels of the ntEdit hot path.
//
//   k_screen        step 1 of the reference's per-contig loop for EVERY k-mer of
//                   a batch at once: rolling forward+reverse ntHash, h-way
//                   Bloom-filter probe, 1 "absent" bit per k-mer start
//                   (ntedit.cpp:1798-1807 + the roll at 2119-2138).  Also the
//                   filter build (insert) variant.
//   k_count_starts / k_scan_counts / k_write_starts
//                   turn the absent bitmap into the ordered list of event starts
//   k_machine       one thread per event: steps 2-5 + makeEdit (nte_machine.h)
//   k_popcount      occupancy of a filter (for get_fpr)
//   k_gather        random 1-byte gather micro-benchmark (roofline denominator)
//
// Roofline: all of this is integer hashing + random 1-byte gathers from a
// multi-GiB bit array => HBM random-read bound; no MFMA anywhere.
#include "nte_machine.h"
#include "nte_machine_launch.h"

#include <hip/hip_runtime.h>

namespace nte {

// ------------------------------------------------------------------ k_screen
// Tile: 256 threads x 64 consecutive k-mer starts per thread = 16384 starts per
// workgroup.  The tile's bases (+ k-1 halo) are loaded once, coalesced 16 B per
// lane, translated to 4-bit character codes through an LDS LUT and parked in
// LDS as one code per byte.  Rows of 64 codes are padded to 68 bytes so that
// the per-thread streams (lane stride = one row) hit 32 different banks.
constexpr int SCREEN_TPB = 256;
constexpr int SCREEN_L = 64;
constexpr int SCREEN_TILE = SCREEN_TPB * SCREEN_L;
constexpr int SCREEN_MAXK = 256;
constexpr int SCREEN_LDS_BYTES = ((SCREEN_TILE + SCREEN_MAXK + 63) / 64) * 68 + 16;

__device__ __forceinline__ u32
lds_phys(u32 x)
{
	return x + ((x >> 6) << 2);
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding
// global store (s_waitcnt vmcnt(0)); the scatter's stores are write-only streams nobody in the kernel
// reads back (measured: partition 45.1 ms against 46.1 ms per 3 Gbp with __syncthreads()).
__device__ __forceinline__ void
lds_barrier()
{
	asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// stage one tile of the batch as 4-bit codes in LDS (padded rows, see k_screen)
__device__ __forceinline__ void
stage_tile(const u8* __restrict__ seq, u64 n, u64 tile_base, u32 tile_len, u32 k, const u8* s_lut, u8* s_codes, u32 tid, u32 tpb)
{
	const u32 n_chunks = (tile_len + k - 1 + 15) / 16;
	for (u32 c = tid; c < n_chunks; c += tpb) {
		const u64 g = tile_base + (u64)c * 16;
		u32 w[4];
		if (g + 16 <= n) {
			const uint4 v = *reinterpret_cast<const uint4*>(seq + g);
			w[0] = v.x;
			w[1] = v.y;
			w[2] = v.z;
			w[3] = v.w;
		} else {
#pragma unroll
			for (int q = 0; q < 4; q++) {
				u32 x = 0;
#pragma unroll
				for (int b = 0; b < 4; b++) {
					const u64 gg = g + q * 4 + b;
					const u32 ch = gg < n ? seq[gg] : (u32)'\n';
					x |= ch << (8 * b);
				}
				w[q] = x;
			}
		}
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const u32 x = w[q];
			const u32 codes = (u32)s_lut[x & 0xFF] | ((u32)s_lut[(x >> 8) & 0xFF] << 8) |
			                  ((u32)s_lut[(x >> 16) & 0xFF] << 16) | ((u32)s_lut[x >> 24] << 24);
			*reinterpret_cast<u32*>(&s_codes[lds_phys(c * 16 + q * 4)]) = codes;
		}
	}
}


// ------------------------------------------------------------------ k_unpack
// A batch that crossed PCIe in the packed form (include/ntedit_hip.h: 4-bit codes + a case bit per base) back into
// the byte batch every kernel reads: 16 bases per thread -- 8 bytes of codes, 2 bytes of case bits in, one 16-byte
// store out.  5 bytes of HBM traffic per 16 bases on top of the 16 written: ~2 ms per 3 Gbp.
__global__ __launch_bounds__(256) void
k_unpack(const u8* __restrict__ codes, const u8* __restrict__ cases, u8* __restrict__ out, u64 first16, u64 n16)
{
	const u64 LET_LO = 0x5753595254474341ULL; // "ACGTRYSW" (codes 0..7), little endian
	const u64 LET_HI = 0x4E4E564844424D4BULL; // "KMBDHVNN" (codes 8..15)
	for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n16; i += (u64)gridDim.x * 256) {
		const u64 g = first16 + i; // group of 16 bases
		const u64 c = *reinterpret_cast<const u64*>(codes + g * 8);
		const u32 cs = *reinterpret_cast<const unsigned short*>(cases + g * 2);
		u64 w[2];
#pragma unroll
		for (int h = 0; h < 2; h++) {
			u64 v = 0;
#pragma unroll
			for (int b = 0; b < 8; b++) {
				const u32 code = (u32)(c >> (4 * (8 * h + b))) & 15u;
				u32 ch = (u32)(((code & 8u) ? LET_HI : LET_LO) >> (8 * (code & 7u))) & 0xFFu;
				ch |= ((cs >> (8 * h + b)) & 1u) << 5; // lower case
				v |= (u64)ch << (8 * b);
			}
			w[h] = v;
		}
		*reinterpret_cast<ulonglong2*>(out + g * 16) = make_ulonglong2(w[0], w[1]);
	}
}

template<int H, bool POW2, bool INSERT>
__global__ __launch_bounds__(SCREEN_TPB) void
k_screen(
    const u8* __restrict__ seq,
    u64 n,
    Filter f,
    DevParams p,
    const u64* __restrict__ tabs,
    u64* __restrict__ bitmap,
    u64 n_words,
    u64 first_tile)
{
	__shared__ __attribute__((aligned(16))) u64 s_tab[TAB_WORDS];
	__shared__ u8 s_lut[256];
	__shared__ __attribute__((aligned(16))) u8 s_codes[SCREEN_LDS_BYTES];
	extern __shared__ u8 s_occupancy_pad[]; // launch-time LDS pad: leaves CU room for k_machine

	const u32 tid = threadIdx.x;
	if (tid < TAB_WORDS) {
		s_tab[tid] = tabs[tid];
	}
	{
		u8 code = char_code((u8)tid);
		if (INSERT && code > 3) {
			code = CODE_BAD; // the filter build takes ACGT-only k-mers
		}
		s_lut[tid] = code;
	}
	__syncthreads();

	const u64 tile_base = (first_tile + blockIdx.x) * SCREEN_TILE;
	const u32 k = p.k;
	stage_tile(seq, n, tile_base, SCREEN_TILE, k, s_lut, s_codes, tid, SCREEN_TPB);
	__syncthreads();

	// ---- per-thread stream of 64 k-mer starts
	const u32 x0 = tid * SCREEN_L;
	HashState hs = { 0, 0 };
	u32 good = 0;
	for (u32 i = 0; i < k; i++) {
		const u8 in = s_codes[lds_phys(x0 + i)];
		hash_roll(hs, s_tab, CODE_BAD, in);
		good = in == CODE_BAD ? 0 : good + 1;
	}

	const u32 ksh = (k & 3) * 8;
	const u32 count_lo = p.min_thr > 1 ? p.min_thr : 1;
	u64 bits = 0;
	u32 in_lo = *reinterpret_cast<const u32*>(&s_codes[lds_phys((x0 + k) & ~3u)]);
	for (u32 j0 = 0; j0 < SCREEN_L; j0 += 4) {
		const u32 outw = *reinterpret_cast<const u32*>(&s_codes[lds_phys(x0 + j0)]);
		const u32 in_hi = *reinterpret_cast<const u32*>(&s_codes[lds_phys(((x0 + j0 + k) & ~3u) + 4)]);
		const u32 inw = ksh ? ((in_lo >> ksh) | (in_hi << (32 - ksh))) : in_lo;
		in_lo = in_hi;

		u64 slot[4][H > 0 ? H : 1];
		bool valid[4];
		u64 base4[4];
#pragma unroll
		for (int u = 0; u < 4; u++) {
			valid[u] = good >= k;
			const u64 base = hs.fh + hs.rh;
			base4[u] = base;
			if (H > 0) {
#pragma unroll
				for (int i = 0; i < (H > 0 ? H : 1); i++) {
					const u64 hv = hash_extend(base, p, i);
					slot[u][i] = POW2 ? (hv & f.mask) : filter_slot(f, hv);
				}
			}
			const u8 out = (outw >> (8 * u)) & 0xFF;
			const u8 in = (inw >> (8 * u)) & 0xFF;
			hash_roll(hs, s_tab, out, in);
			good = in == CODE_BAD ? 0 : good + 1;
		}

		if (INSERT) {
#pragma unroll
			for (int u = 0; u < 4; u++) {
				if (valid[u]) {
					if (H > 0) {
#pragma unroll
						for (int i = 0; i < (H > 0 ? H : 1); i++) {
							const u64 s = slot[u][i];
							atomicOr(
							    reinterpret_cast<u32*>(const_cast<u8*>(f.data)) + (s >> 5),
							    1u << (s & 31));
						}
					} else {
						for (u32 i = 0; i < f.hash_num; i++) {
							const u64 s = filter_slot(f, hash_extend(base4[u], p, i));
							atomicOr(
							    reinterpret_cast<u32*>(const_cast<u8*>(f.data)) + (s >> 5),
							    1u << (s & 31));
						}
					}
				}
			}
		} else if (p.snv) {
			// -s 1: every k-mer of accepted bases is re-assessed, no probe decides that
#pragma unroll
			for (int u = 0; u < 4; u++) {
				if (valid[u]) {
					bits |= 1ULL << (j0 + u);
				}
			}
		} else {
			if (H > 0) {
				u8 byte[4][H > 0 ? H : 1];
#pragma unroll
				for (int u = 0; u < 4; u++) {
#pragma unroll
					for (int i = 0; i < (H > 0 ? H : 1); i++) {
						// invalid k-mers read byte 0 instead of branching: keeps all
						// 4*H gathers independent and in flight together
						byte[u][i] = f.data[valid[u] ? (f.counting ? slot[u][i] : (slot[u][i] >> 3)) : 0];
					}
				}
#pragma unroll
				for (int u = 0; u < 4; u++) {
					u32 present = 1;
#pragma unroll
					for (int i = 0; i < (H > 0 ? H : 1); i++) {
						// plain filter: the bit; counting filter: every counter >= max(1, -p)
						present &= f.counting ? (u32)(byte[u][i] >= count_lo) : ((byte[u][i] >> (slot[u][i] & 7)) & 1);
					}
					if (valid[u] && !present) {
						bits |= 1ULL << (j0 + u);
					}
				}
			} else {
#pragma unroll
				for (int u = 0; u < 4; u++) {
					if (valid[u]) {
						const HashState hb = { base4[u], 0 };
						if (filter_screen_absent(f, p, hb)) {
							bits |= 1ULL << (j0 + u);
						}
					}
				}
			}
		}
	}
	if (!INSERT) {
		const u64 w = (first_tile + blockIdx.x) * SCREEN_TPB + tid;
		if (w < n_words) {
			bitmap[w] = bits;
		}
	}
}

// ------------------------------------------------------- binned screening
// The direct kernel above is bound by the L2-miss path: every 1-byte probe of a multi-GiB filter costs one
// 64-byte fabric request (~51 G requests/s measured, the same as a pure random-gather micro-benchmark),
// while gathers that HIT in an XCD's 4 MiB L2 run at up to ~270 G/s.  The binned pipeline makes the probes
// L2-resident by partitioning them by filter slice first:
//   k_wc_scatter_b (default) / k_wc_scatter   (nte_bin_wc.inc) write-combining partition of the h probes of every
//                  k-mer into {position, offset-in-slice} records, slice by slice, ONE pass over the draft
//   k_bin_probe    every XCD walks slices one after the other: the slice's 2-4 MiB of filter stay in that XCD's L2
//                  while its records stream by; a zero bit ORs the k-mer's bit into the absent bitmap
// Records are 8 bytes: (global k-mer position << slice_log2) | slot offset in slice.
// Measured (MI355X, 3 Gbp, 4 GiB filter, h = 3): 94 ms against 176 ms for the direct kernel.
constexpr u64 WC_EMPTY_REC = ~0ULL;             // padding of a run's last 64-byte group: no record
constexpr u64 WC_REC_POS_MASK = ~(1ULL << 63);  // bit 63 of a stored record is the ring generation bit

struct BinArgs
{
	const u8* seq;
	u64 n;           // total batch bytes (for the halo / tail)
	u64 chunk_begin; // first k-mer start of this chunk (multiple of SCREEN_TILE)
	u64 chunk_end;
	Filter f;
	DevParams p;
	const u64* tabs;
	u32 n_slices;
	u32 slice_log2;  // log2(bits per slice)
	u64* records;    // runs of `cap` records, one per (slice, partition workgroup) pair
};

// Probe stage.  A slice's 2-4 MiB of filter fit the 4 MiB L2 of ONE XCD, so all workgroups of an XCD work on the
// same slice at the same time: a workgroup asks the hardware which XCD it runs on (HW_REG_XCC_ID) and draws
// stretches of that XCD's CURRENT slice from the slice's counter, one 512-record step per wavefront.  Large
// workgroups (16 wavefronts, two per CU) keep the draws rare (a counter that everybody on an XCD draws from serves
// ~15 M draws/s) AND the work in flight small: when the XCD moves on to its next slice only 64 stretches of the old one
// are still being probed (two slices do not fit the L2 together; with a draw per wavefront half a slice was in flight
// at every switch and a quarter of all probes missed the L2).
// Slices are not tied to XCDs: the workgroup whose draw reaches the end of a slice takes the next unclaimed slice from a
// global counter and publishes it as its XCD's current one.  Every slice is therefore probed completely whatever the
// number of XCDs the device exposes (compute partitioning: 1, 2, 4 or 8), whichever of them receive workgroups, and
// however fast each one is; placement only affects speed, never the result.
// ctl[] (zeroed before the launch): [0] next unclaimed slice, [1..16] current slice of XCD x (0 = none yet,
// 1 = being claimed, s + 2 = slice s), [32 + s] records of slice s handed out so far.
// A slice's records: n_wg runs of `cap` records; in the counter's coordinates every run takes capr = cap rounded up to
// whole steps, so no step crosses runs.
// Filters beyond 1024 x 4 MiB: the partition kernel has rings for 1024 slices (its LDS), so the slices grow past what
// an XCD's L2 holds -- 8 MiB at 8 GiB, where half the gathers would miss.  Such a slice is probed in 2^plog PARTS of 4 MiB:
// the unit the XCDs claim is a (slice, part) pair, the slice's records stream by once per part and every pass probes the
// records whose slot lies in its part (ctl[] then has one "slice" entry per pair).  Re-reading the records costs less than
// missing the L2: 8 GiB 88 -> 6x ms, and a 16 GiB filter no longer falls back to the direct kernel (185 ms).
#ifndef NTE_PROBE_LOAD
#define NTE_PROBE_LOAD 1 // record loads: 0 = 8 bytes per lane, non-temporal; 1 = 16 bytes, non-temporal; 2 = 16 bytes, plain
#endif
#ifndef NTE_PROBE_TPB
#define NTE_PROBE_TPB 512
#endif
constexpr int PROBE_TPB = NTE_PROBE_TPB;
constexpr int PROBE_WAVES = PROBE_TPB / 64;
#ifndef NTE_PROBE_PER
#define NTE_PROBE_PER 8
#endif
constexpr int PROBE_PER = NTE_PROBE_PER;      // records per lane and step, all in flight together
constexpr int PROBE_STEP = 64 * PROBE_PER;    // records per step of a wavefront
constexpr u32 PROBE_DRAW = PROBE_STEP * PROBE_WAVES; // records per draw of a workgroup
constexpr u32 CTL_NEXT = 0, CTL_CUR = 1, CTL_WORK = 32;
constexpr u32 PROBE_XCC_ANY = 0xFFFFFFFFu;

struct ProbeArgs
{
	const u8* filter;
	const u64* records;
	const u32* fill; // [n_slices][n_wg]
	u32 n_slices;
	u32 slog;
	u32 n_wg;
	u32 cap;
	u32* ctl;
	u32* absent32;
	u32 force_xcc;   // PROBE_XCC_ANY, or the XCD id every workgroup pretends to run on (tests)
	u32 counting;    // the filter holds 8-bit counters: a slot is a byte, "absent" = counter < count_lo
	u32 count_lo;    // max(1, -p) (ntedit.cpp:1806)
	u32 plog;        // log2 of the parts a slice is probed in (0: whole slices)
	u32 mark_present; // 1: OR the bit of a record whose slot IS set (the candidate map of -s 1) instead of the absent ones
};

__device__ __forceinline__ u32
xcc_id()
{
	// s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
	return __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u;
}

__device__ __forceinline__ u32
ctl_load(const u32* p)
{
	return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One thread: the next stretch [p, p + PROBE_DRAW) of the XCD's current slice, moving on to further slices as they run
// out.  On entry {sl, p} is a draw already made from slice `sl` (or sl == NONE32: none yet).  Returns false when no
// slice is left.
__device__ __forceinline__ bool
probe_claim(u32* __restrict__ ctl, u32 xcd, u32 n_slices, u32 total, u32& sl, u32& p)
{
	u32* cur = ctl + CTL_CUR + xcd;
	for (;;) {
		if (sl != NONE32 && p < total) {
			if (total - p <= PROBE_DRAW) {
				// this draw reaches the end of the slice: open the next one for the whole XCD
				const u32 nx = atomicAdd(&ctl[CTL_NEXT], 1u);
				__hip_atomic_store(cur, nx + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
			return true;
		}
		// the XCD's current slice (another workgroup may be switching it right now)
		u32 v = ctl_load(cur);
		for (;;) {
			if (v == 0) {
				if (atomicCAS(cur, 0u, 1u) == 0u) {
					const u32 nx = atomicAdd(&ctl[CTL_NEXT], 1u);
					__hip_atomic_store(cur, nx + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
			} else if (v >= 2 && v - 2 != sl) {
				break;
			} else {
				__builtin_amdgcn_s_sleep(4);
			}
			v = ctl_load(cur);
		}
		sl = v - 2;
		if (sl >= n_slices) {
			return false;
		}
		p = atomicAdd(&ctl[CTL_WORK + sl], PROBE_DRAW);
	}
}

__global__ __launch_bounds__(PROBE_TPB) void
k_bin_probe(ProbeArgs a)
{
	__shared__ u32 s_draw[2][4]; // {ok, slice, start}, double-buffered: one barrier per draw
	const u32 lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const u64 off_mask = (1ULL << a.slog) - 1;
	const u32 xcd = a.force_xcc == PROBE_XCC_ANY ? xcc_id() : (a.force_xcc & 15u);
	const u32 capr = (a.cap + PROBE_STEP - 1) / PROBE_STEP * PROBE_STEP; // a run in the counter's coordinates
	const u32 total = a.n_wg * capr;
	const u32 bsh = a.counting ? 0u : 3u; // slots per byte, as a shift
	const u32 n_units = a.n_slices << a.plog; // (slice, part) pairs
	const u32 part_shift = a.slog - a.plog;
	u32 sl = NONE32, p = 0; // (thread 0: the draw made ahead; sl counts (slice, part) pairs)
	for (u32 it = 0;; it++) {
		u32* d = s_draw[it & 1];
		if (threadIdx.x == 0) {
			const bool ok = probe_claim(a.ctl, xcd, n_units, total, sl, p);
			d[0] = ok ? 1u : 0u;
			d[1] = sl;
			d[2] = p;
			if (ok) {
				// draw the stretch after this one now: its round trip hides behind the probes
				p = atomicAdd(&a.ctl[CTL_WORK + sl], PROBE_DRAW);
			}
		}
		__syncthreads();
		if (!d[0]) {
			return;
		}
		const u32 dsl = d[1];
		const u32 at = d[2] + wave * PROBE_STEP;
		if (at >= total) {
			continue;
		}
		const u32 rsl = dsl >> a.plog; // the slice of the record array
		const u32 part = dsl & ((1u << a.plog) - 1u);
		const u8* __restrict__ fs = a.filter + ((u64)rsl << (a.slog - bsh));
		const u32 w = at / capr;
		const u32 i0 = at - w * capr;
		const u32 fill = a.fill[(size_t)rsl * a.n_wg + w];
		if (i0 >= fill) {
			continue;
		}
		const u32 n = fill - i0 < (u32)PROBE_STEP ? fill - i0 : (u32)PROBE_STEP;
		const u64* __restrict__ src = a.records + ((u64)(rsl * a.n_wg + w) * a.cap + i0);
		u64 rec[PROBE_PER];
		u8 byte[PROBE_PER];
#if NTE_PROBE_LOAD == 0
#pragma unroll
		for (int q = 0; q < PROBE_PER; q++) {
			const u32 i = (u32)q * 64 + lane;
			// (non-temporal: the records are read once; the slice they probe should keep the XCD's L2)
			rec[q] = i < n ? __builtin_nontemporal_load(src + i) : WC_EMPTY_REC;
		}
#else
		// 16 bytes per lane, non-temporal (54.3 ms against 55.2 for 8 bytes per lane; plain loads: 60.3 ms, the record stream
		// then evicts the slice); runs are padded to whole groups of 8, so pairs never straddle their end
#pragma unroll
		for (int q = 0; q < PROBE_PER / 2; q++) {
			const u32 i = ((u32)q * 64 + lane) * 2;
			ulonglong2 v = make_ulonglong2(WC_EMPTY_REC, WC_EMPTY_REC);
			if (i < n) {
#if NTE_PROBE_LOAD == 1
				v.x = __builtin_nontemporal_load(src + i);
				v.y = __builtin_nontemporal_load(src + i + 1);
#else
				v = *reinterpret_cast<const ulonglong2*>(src + i);
#endif
			}
			rec[2 * q] = v.x;
			rec[2 * q + 1] = v.y;
		}
#endif
#pragma unroll
		for (int q = 0; q < PROBE_PER; q++) {
			const u32 off = (u32)(rec[q] & off_mask);
			// (a record of another part of the slice reads as "present": it is probed in that part's pass)
			byte[q] = rec[q] != WC_EMPTY_REC && (off >> part_shift) == part ? fs[off >> bsh] : (u8)0xFF;
		}
#pragma unroll
		for (int q = 0; q < PROBE_PER; q++) {
			const u32 off = (u32)(rec[q] & off_mask);
			// plain filter: the bit; counting filter: the counter must reach max(1, -p)
			const bool absent = a.counting ? (u32)byte[q] < a.count_lo : !((byte[q] >> (off & 7)) & 1);
			if (absent != (a.mark_present != 0) && rec[q] != WC_EMPTY_REC && (off >> part_shift) == part) {
				const u64 pos = (rec[q] & WC_REC_POS_MASK) >> a.slog;
				atomicOr(&a.absent32[pos >> 5], 1u << (pos & 31));
			}
		}
	}
}

#include "nte_bin_wc.inc"

// ------------------------------------------------------------ event starts
// bits of word w whose position lies in [pos_lo, pos_hi)
__device__ __forceinline__ u64
range_mask(u64 w, u64 pos_lo, u64 pos_hi)
{
	const u64 b = w * 64;
	u64 m = ~0ULL;
	if (b < pos_lo) {
		const u64 d = pos_lo - b;
		m = d >= 64 ? 0 : (m << d);
	}
	if (b + 64 > pos_hi) {
		const u64 d = pos_hi > b ? pos_hi - b : 0;
		m &= d >= 64 ? ~0ULL : ((1ULL << d) - 1);
	}
	return m;
}

__device__ __forceinline__ u64
start_mask(const u64* __restrict__ bitmap, u64 w, u64 grid_lo, u32 grid)
{
	const u64 m = bitmap[w];
	if (m == 0) {
		return 0;
	}
	const u64 prev = w ? (bitmap[w - 1] >> 63) : 0;
	const u64 shifted = (m << 1) | prev;
	u64 gm = grid_lo;
	if (grid >= 64) {
		gm = ((w * 64) % grid) == 0 ? 1ULL : 0ULL;
	}
	return m & (~shifted | gm);
}

constexpr int ST_TPB = 256;

// A range of the batch [pos_lo, pos_hi) (one pipeline chunk = whole contigs) is turned into
// its ordered event list: words [w0, w0 + gridDim*256).
// block_counts[b] = starts in block b (from `bitmap`: the screening bitmap or the run map), block_counts[gridDim + b] = absent
// k-mers of block b (always the set bits of the screening bitmap: ntedit_hip_stats::absent_kmers)
__global__ __launch_bounds__(ST_TPB) void
k_count_starts(
    const u64* __restrict__ bitmap,
    u64 w0,
    u64 w1,
    u64 pos_lo,
    u64 pos_hi,
    u64 grid_lo,
    u32 grid,
    u32* __restrict__ block_counts,
    const u64* __restrict__ absent_bitmap) // the screening bitmap (= bitmap unless the starts come from the run map)
{
	__shared__ u32 s_cnt[ST_TPB / 64];
	__shared__ u32 s_abs[ST_TPB / 64];
	const u64 w = w0 + (u64)blockIdx.x * ST_TPB + threadIdx.x;
	u32 c = 0, a = 0;
	if (w < w1) {
		const u64 rm = range_mask(w, pos_lo, pos_hi);
		c = __popcll(start_mask(bitmap, w, grid_lo, grid) & rm);
		a = __popcll(absent_bitmap[w] & rm);
	}
	for (int off = 32; off > 0; off >>= 1) {
		c += __shfl_down(c, off, 64);
		a += __shfl_down(a, off, 64);
	}
	if ((threadIdx.x & 63) == 0) {
		s_cnt[threadIdx.x >> 6] = c;
		s_abs[threadIdx.x >> 6] = a;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		u32 tc = 0, ta = 0;
		for (int i = 0; i < ST_TPB / 64; i++) {
			tc += s_cnt[i];
			ta += s_abs[i];
		}
		block_counts[blockIdx.x] = tc;
		// (the absent k-mers of the block go behind the counts: 183 k blocks bumping ONE counter took 2 ms per 3 Gbp;
		// k_scan_counts adds them up)
		block_counts[gridDim.x + blockIdx.x] = ta;
	}
}

// exclusive scan of block_counts (single workgroup); counters[1] = total
__global__ __launch_bounds__(1024) void
k_scan_counts(
    const u32* __restrict__ block_counts,
    u64 n_blocks,
    unsigned long long* __restrict__ block_offsets,
    unsigned long long* __restrict__ counters)
{
	__shared__ unsigned long long s_part[1024];
	__shared__ unsigned long long s_carry;
	__shared__ unsigned long long s_abs[16];
	if (threadIdx.x == 0) {
		s_carry = 0;
	}
	__syncthreads();
	unsigned long long absent = 0; // block_counts[n_blocks + b]: absent k-mers of block b (k_count_starts)
	for (u64 base = 0; base < n_blocks; base += 1024) {
		const u64 i = base + threadIdx.x;
		const unsigned long long v = i < n_blocks ? block_counts[i] : 0;
		absent += i < n_blocks ? block_counts[n_blocks + i] : 0;
		s_part[threadIdx.x] = v;
		__syncthreads();
		// Hillis-Steele inclusive scan
		for (int off = 1; off < 1024; off <<= 1) {
			unsigned long long t = 0;
			if ((int)threadIdx.x >= off) {
				t = s_part[threadIdx.x - off];
			}
			__syncthreads();
			s_part[threadIdx.x] += t;
			__syncthreads();
		}
		const unsigned long long incl = s_part[threadIdx.x];
		const unsigned long long carry = s_carry;
		if (i < n_blocks) {
			block_offsets[i] = carry + incl - v;
		}
		__syncthreads();
		if (threadIdx.x == 1023) {
			s_carry = carry + incl;
		}
		__syncthreads();
	}
	for (int off = 32; off > 0; off >>= 1) {
		absent += __shfl_down(absent, off, 64);
	}
	if ((threadIdx.x & 63) == 0) {
		s_abs[threadIdx.x >> 6] = absent;
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		unsigned long long a = 0;
		for (int i = 0; i < 16; i++) {
			a += s_abs[i];
		}
		counters[0] += a; // (this chunk's; the counter is zeroed per attempt)
		counters[1] = s_carry;
	}
}

__global__ __launch_bounds__(ST_TPB) void
k_write_starts(
    const u64* __restrict__ bitmap,
    u64 w0,
    u64 w1,
    u64 pos_lo,
    u64 pos_hi,
    u64 grid_lo,
    u32 grid,
    const unsigned long long* __restrict__ block_offsets,
    u64* __restrict__ events)
{
	__shared__ u32 s_scan[ST_TPB];
	const u64 w = w0 + (u64)blockIdx.x * ST_TPB + threadIdx.x;
	u64 m = 0;
	if (w < w1) {
		m = start_mask(bitmap, w, grid_lo, grid) & range_mask(w, pos_lo, pos_hi);
	}
	const u32 c = __popcll(m);
	s_scan[threadIdx.x] = c;
	__syncthreads();
	for (int off = 1; off < ST_TPB; off <<= 1) {
		u32 t = 0;
		if ((int)threadIdx.x >= off) {
			t = s_scan[threadIdx.x - off];
		}
		__syncthreads();
		s_scan[threadIdx.x] += t;
		__syncthreads();
	}
	u64 o = block_offsets[blockIdx.x] + (s_scan[threadIdx.x] - c);
	while (m) {
		const int b = __ffsll((long long)m) - 1;
		events[o++] = w * 64 + b;
		m &= m - 1;
	}
}

// ------------------------------------------------------------ event rounds
// Events are speculative: one that starts inside an earlier event's serial run is discarded by the
// serial-order filter.  With a Bloom filter's false positives an absent run is often broken into two or three
// pieces, every piece starts an event, and all but the first of them are, as a rule, overtaken by the first one's
// run -- after having cost as much machine time as the useful events together (measured: 26 % of the events, 42 %
// of the machine work).  The launch driver therefore runs the events in rounds:
//   round A   PRIMARY events: no other event starts within `gap` positions in front of them
//   select    a SECONDARY event is skipped iff it starts inside the run of its cluster's primary P
//             (events[i] < cover[P], P ran to its end) and P itself is certainly applied: no event that has run
//             so far reaches P (max cover of all earlier run events <= start of P; k_ev_prefix_max)
//   round B   the secondaries that are not skipped
//   verify    the "certainly applied" test again with round B's runs; any violation -> the skipped events run too
// A skipped event is exactly one the serial order discards (cover only grows along a contig), so the records
// the host gets are the same set of APPLIED events as before; first_chunk = NONE32 makes it skip the others.
constexpr u8 EVC_PRIMARY = 0x20, EVC_SKIPPED = 0x40, EVC_RAN = 0x80; // (+ EV_UNFINISHED = 0x10 from the machine)
constexpr int EVR_TPB = 256;

// appends the values of the lanes with `pred` to list[*count...], in thread order within the workgroup (EVR_TPB threads,
// every thread of the workgroup calls it): ONE bump of the counter per workgroup
__device__ __forceinline__ void
wave_append(u32* list, u32* count, bool pred, u32 value)
{
	__shared__ u32 s_n[EVR_TPB / 64];
	__shared__ u32 s_base;
	const u64 m = __ballot(pred);
	const u32 lane = __lane_id(), wave = threadIdx.x >> 6;
	if (lane == 0) {
		s_n[wave] = (u32)__popcll(m);
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		u32 t = 0;
		for (int i = 0; i < EVR_TPB / 64; i++) {
			t += s_n[i];
		}
		s_base = t ? atomicAdd(count, t) : 0u;
	}
	__syncthreads();
	if (pred) {
		u32 off = s_base;
		for (u32 i = 0; i < wave; i++) {
			off += s_n[i];
		}
		list[off + (u32)__popcll(m & ((1ULL << lane) - 1))] = value;
	}
}

__global__ __launch_bounds__(EVR_TPB) void
k_ev_primaries(const u64* __restrict__ events, u32 n, u32 gap, u8* __restrict__ flags, u32* __restrict__ list, u32* __restrict__ count)
{
	const u32 i = blockIdx.x * EVR_TPB + threadIdx.x;
	bool prim = false;
	if (i < n) {
		prim = i == 0 || events[i] - events[i - 1] > gap;
		flags[i] = prim ? EVC_PRIMARY : 0;
	}
	wave_append(list, count, prim, i);
}

// exclusive prefix maximum of cover[] (0 for events that have not run): stage 1, per block
__global__ __launch_bounds__(1024) void
k_ev_prefix_max_1(const u64* __restrict__ cover, u32 n, u64* __restrict__ before, u64* __restrict__ block_max)
{
	__shared__ u64 s[1024];
	const u32 i = blockIdx.x * 1024 + threadIdx.x;
	const u64 v = i < n ? cover[i] : 0;
	s[threadIdx.x] = v;
	__syncthreads();
	for (int off = 1; off < 1024; off <<= 1) {
		u64 t = 0;
		if ((int)threadIdx.x >= off) {
			t = s[threadIdx.x - off];
		}
		__syncthreads();
		if (t > s[threadIdx.x]) {
			s[threadIdx.x] = t;
		}
		__syncthreads();
	}
	if (i < n) {
		before[i] = threadIdx.x ? s[threadIdx.x - 1] : 0; // exclusive, within the block
	}
	if (threadIdx.x == 1023) {
		block_max[blockIdx.x] = s[1023];
	}
}

// stage 2: fold in the maxima of all earlier blocks
__global__ __launch_bounds__(1024) void
k_ev_prefix_max_2(u32 n, u64* __restrict__ before, const u64* __restrict__ block_max)
{
	__shared__ u64 s_red[16];
	u64 carry = 0;
	for (u32 b = threadIdx.x; b < blockIdx.x; b += 1024) {
		const u64 v = block_max[b];
		carry = v > carry ? v : carry;
	}
	for (int off = 32; off > 0; off >>= 1) {
		const u64 o = __shfl_down(carry, off, 64);
		carry = o > carry ? o : carry;
	}
	if ((threadIdx.x & 63) == 0) {
		s_red[threadIdx.x >> 6] = carry;
	}
	__syncthreads();
	carry = 0;
	for (int w = 0; w < 16; w++) {
		carry = s_red[w] > carry ? s_red[w] : carry;
	}
	const u32 i = blockIdx.x * 1024 + threadIdx.x;
	if (i < n && carry > before[i]) {
		before[i] = carry;
	}
}

// the primary of event i's cluster (walking back over at most 64 secondaries), or NONE32
__device__ __forceinline__ u32
cluster_primary(const u8* __restrict__ flags, u32 i)
{
	for (u32 step = 1; step <= 64 && step <= i; step++) {
		if (flags[i - step] & EVC_PRIMARY) {
			return i - step;
		}
	}
	return NONE32;
}

// MODE 0: decide the secondaries (skip or append to `list`).  MODE 1: verify the skipped ones (count violations in
// *count and append them to `list`).
template<int MODE>
__global__ __launch_bounds__(EVR_TPB) void
k_ev_select(
    const u64* __restrict__ events,
    u32 n,
    const u64* __restrict__ cover,
    const u64* __restrict__ before,
    u8* __restrict__ flags,
    u32* __restrict__ first_chunk,
    u32* __restrict__ list,
    u32* __restrict__ count)
{
	const u32 i = blockIdx.x * EVR_TPB + threadIdx.x;
	bool take = false;
	if (i < n) {
		const u8 f = flags[i];
		const bool candidate = MODE == 0 ? !(f & EVC_PRIMARY) : (f & EVC_SKIPPED) != 0;
		if (candidate) {
			const u32 P = cluster_primary(flags, i);
			bool covered = false;
			if (P != NONE32) {
				const u8 pf = flags[P];
				const bool p_done = (pf & EVC_RAN) && !(pf & EV_UNFINISHED);
				covered = p_done && before[P] <= events[P] && events[i] < cover[P];
			}
			if (MODE == 0) {
				if (covered) {
					flags[i] = (u8)(f | EVC_SKIPPED);
					first_chunk[i] = NONE32;
				} else {
					take = true;
				}
			} else if (!covered) {
				flags[i] = (u8)(f & ~EVC_SKIPPED);
				take = true;
			}
		}
	}
	wave_append(list, count, take, i);
}

// (k_machine lives in its own translation units, nte_machine_thread.hip / nte_machine_wave.hip:
// the fully inlined state machine is by far the largest kernel and the two variants compile in
// parallel; interface in nte_machine_launch.h)

// ---------------------------------------------------------------- k_popcount
// set bits of a filter (plain) / non-zero counters (counting): the occupancy behind
// btllib's get_fpr() = (occupied / slots)^hash_num, printed by ntedit-make-genome-bf
// (src/ntedit_make_genome_bf.cpp:159).  Streaming read, one atomic per wavefront.
__global__ __launch_bounds__(256) void
k_popcount(const u64* __restrict__ words, u64 n_words, int counting, unsigned long long* __restrict__ total)
{
	u64 acc = 0;
	for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (u64)gridDim.x * blockDim.x) {
		const u64 w = words[i];
		if (counting) {
			// bytes that are not zero: fold every byte's bits onto its lowest bit
			u64 t = w | (w >> 4);
			t |= t >> 2;
			t |= t >> 1;
			acc += (u64)__popcll(t & 0x0101010101010101ULL);
		} else {
			acc += (u64)__popcll(w);
		}
	}
	for (int off = 32; off > 0; off >>= 1) {
		acc += __shfl_down(acc, off, 64);
	}
	if ((threadIdx.x & 63) == 0 && acc) {
		atomicAdd(total, (unsigned long long)acc);
	}
}

// ------------------------------------------------------------------ k_gather
// Uniform random 1-byte gathers, 12 independent loads in flight per lane
// (the same depth k_screen uses at h=3): the random-read roofline reference.
__global__ __launch_bounds__(256) void
k_gather(const u8* __restrict__ data, u64 mask, u64 per_thread, u32* __restrict__ sink)
{
	u64 x = ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
	u32 acc = 0;
	for (u64 it = 0; it < per_thread; it += 12) {
		u64 a[12];
#pragma unroll
		for (int i = 0; i < 12; i++) {
			x ^= x << 13;
			x ^= x >> 7;
			x ^= x << 17;
			a[i] = (x * 0x2545F4914F6CDD1DULL) & mask;
		}
		u8 b[12];
#pragma unroll
		for (int i = 0; i < 12; i++) {
			b[i] = data[a[i] >> 3];
		}
#pragma unroll
		for (int i = 0; i < 12; i++) {
			acc += (b[i] >> (a[i] & 7)) & 1;
		}
	}
	if (acc == 0xFFFFFFFFu) {
		sink[0] = acc;
	}
	if (threadIdx.x == 0 && blockIdx.x == 0) {
		sink[1] = acc;
	}
}

} // namespace nte
