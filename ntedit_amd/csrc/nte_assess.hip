// nte_assess.hip -- k_assess: which absent positions can do anything at all?
//
// Where the serial program stands in its clean state the assessment of a failing position is a function
// of draft and filter alone, and it begins with two tests that most positions of some workloads never get past: step 2
// (ntedit.cpp:1826-1873) and the presence of a substitution candidate's own k-mer (ntedit.cpp:1916-1934) -- without
// such a candidate there is no substitution, no indel sweep (its index base is that candidate), no upper-cased revert
// and no -s 1 report: the position is a no-op.  With -s 1 EVERY position is assessed and one in a thousand gets that
// far; with a counting filter and -p 2 a quarter of the k-mers count as absent.  This kernel applies the two tests
// to every position of the absent bitmap, one position per lane, and writes the "run map": the positions that
// survive.  The event machine starts, goes on and stops by the run map (a cleared position is to it what a present
// k-mer is) and still reads the absent bitmap where step 2 asks for it, so nothing changes but the work.
// The tests are Machine::assess_lane<GATE> itself (nte_machine.h) on a character window shared by the workgroup;
// positions the window cannot serve (a non-accepted character or the end of the contig within k bases behind the
// k-mer) are asked the gate's first question only, which reads the k-mer's own codes; where a tile holds bytes the
// 4-bit codes cannot express they keep their bit.
#include "nte_machine_launch.h"

#include <hip/hip_runtime.h>

namespace nte {

constexpr int ASSESS_TPB = 256;
constexpr int ASSESS_L = 16;                          // consecutive positions per lane (a quarter of a bitmap word)
constexpr int ASSESS_TILE = ASSESS_TPB * ASSESS_L;    // positions per workgroup tile
constexpr int ASSESS_MAX_WIN = ASSESS_TILE + 2 * 200 + 10 + 1 + 32; // k <= 200, max_deletions <= 10 (params.cpp)

// A lane walks ASSESS_L consecutive positions: it seeds the hash of its first k-mer from the window and ROLLS from
// there on (2 table look-ups per position instead of 2k), keeps count of the accepted codes in front of it (the window
// of a position has to hold k + win_len_in() of them), and asks Machine::assess_gate() where the absent bitmap has the
// position.  Four lanes make one word of the run map.
#ifndef NTE_ASSESS_MIN_BLOCKS
#define NTE_ASSESS_MIN_BLOCKS 5 // blocks of 256 per CU the register allocation leaves room for (round 6, the kernel in two phases:
                                // 4 / 5 / 6 / 7 blocks -> -s 1 19.0 / 18.5 / 19.3 / 19.3 ms, counting filter 30.6 / 28.9 / 28.8 / 28.1)
#endif
__global__ __launch_bounds__(ASSESS_TPB, NTE_ASSESS_MIN_BLOCKS) void
k_assess(AssessArgs a)
{
	__shared__ __attribute__((aligned(16))) u64 s_tab[TAB_WORDS];
	__shared__ __attribute__((aligned(16))) u8 s_win[ASSESS_MAX_WIN];
	__shared__ u32 s_list[ASSESS_TILE]; // second phase: (window offset << 4) | candidates there
	__shared__ u32 s_keep[ASSESS_TPB];  // the lanes' keep bits, completed by the second phase
	__shared__ u32 s_n_list;
	if (threadIdx.x < TAB_WORDS) {
		s_tab[threadIdx.x] = a.tabs[threadIdx.x];
	}
	const u32 k = a.p.k;
	// accepted codes the gate reads at a position: its k-mer and the k rolls of step 2 / the substitution walks (the gate
	// never sweeps; round 6: until then the whole window of a failing position, k + max_del + 1 rolls, was asked for)
	const u32 need = 2 * k;
	const u32 span = ASSESS_TILE + need;     // codes a tile's lanes read
	EventEnv env;
	env.seq = a.seq;
	env.batch_end = a.seq + a.n_bytes;
	env.len = 0;
	env.contig = 0;
	env.gbase = 0;
	env.bitmap = a.bitmap;
	env.runmap = a.bitmap;
	env.tab = s_tab;
	env.p = &a.p;
	env.bloom = a.bloom;
	env.rep = a.rep;
	env.nodes = nullptr;
	env.ov_pos = nullptr;
	env.ov_chr = nullptr;
	env.win = s_win;
	env.win_stride = 1;
	env.prev = nullptr;
	env.lps = nullptr;
	env.arena = nullptr;
	env.arena_next = nullptr;
	env.arena_chunks = 0;
	env.defer_sweeps = false;
	env.wave_size = 1;
	const u32 x0 = threadIdx.x * ASSESS_L;
	if (threadIdx.x == 0) {
		s_n_list = 0;
	}
	for (u64 tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
		const u64 base = a.pos_begin + tile * ASSESS_TILE; // (a multiple of 64)
		__syncthreads(); // (the window of the previous tile is no longer read)
		// this lane's quarter of a bitmap word
		const u64 p0 = base + x0;
		u32 abits = 0;
		if (p0 < a.pos_end) {
			abits = (u32)(a.bitmap[p0 >> 6] >> (p0 & 63)) & 0xFFFFu;
			if (a.pos_end - p0 < ASSESS_L) {
				abits &= (1u << (a.pos_end - p0)) - 1;
			}
		}
		if (!__syncthreads_or(abits != 0)) {
			// nothing absent in the whole tile
			if ((threadIdx.x & 3) == 0 && p0 < a.pos_end) {
				a.runmap[p0 >> 6] = 0;
			}
			continue;
		}
		bool exotic = false;
		for (u32 i = threadIdx.x; i < span; i += ASSESS_TPB) {
			u8 code = CODE_BAD;
			if (base + i < a.n_bytes) {
				const u8 ch = a.seq[base + i];
				code = char_code(ch);
				exotic |= code == CODE_BAD && is_exotic(ch);
			}
			s_win[i] = code;
		}
		// (a byte the 4-bit codes cannot express anywhere in the tile: every absent position keeps its bit)
		const bool plain = !__syncthreads_or(exotic ? 1 : 0);
		u32 keep = abits;
		// (the candidate map applies where the gate's first question comes first: not with -m 2 / -a, plain filters only)
		const bool use_cmap = a.cand_map != nullptr && a.p.mode != 2 && !a.p.mask && !a.p.counting;
		u64 cmap = 0;
		if (use_cmap && abits) {
			cmap = *reinterpret_cast<const u64*>(a.cand_map + (p0 >> 3)); // 16 positions x 4 bits (p0 is a multiple of 16)
		}
		if (plain && abits) {
			MachineT<0> m(env); // (the general machine: an instantiation for -m 0 / no secondary filter / power-of-two sizes
			                    // was built in round 5 and ran SLOWER, 40.5 against 35.7 ms per 250 Mbp with -s 1)
			m.win_ok = true;
			m.win_off = x0;
			HashState hs = m.seed_from_window();
			// accepted codes in a row ending at the far end of the first position's window
			u32 good = 0;
			for (u32 i = 0; i < need; i++) {
				good = s_win[x0 + i] != CODE_BAD ? good + 1 : 0;
			}
			keep = 0;
			// Two phases (round 6).  One position in 27 gets past the gate's first question, but of the 64 positions a
			// wavefront assesses together nearly always one does -- and the others wait while it runs step 2 and its
			// candidates' support walks, twenty dependent probe rounds, on ONE lane: 32 of the kernel's 36 ms per 250 Mbp
			// with -s 1.  So the lanes ask the first question only, as they roll along their positions, and put the
			// positions that pass on a list in the LDS; the rest of the assessment then runs on that list, every lane a
			// survivor.  (-m 2 / -a: the first question is not the first, assess_gate as before.)
			const bool two_phase = a.p.mode != 2 && !a.p.mask;
			for (u32 j = 0; j < (u32)ASSESS_L; j++) {
				if ((abits >> j) & 1) {
					bool kp = true;
					m.win_off = x0 + j;
					if (use_cmap) {
						// the candidate map: which of the position's candidates have the first bit of their k-mer set.  None:
						// no candidate's k-mer is there, the position is a no-op -- without a single gather.
						m.cand_l1 = (u32)((cmap >> (4 * j)) & 15u);
					}
					if (use_cmap && m.cand_l1 == 0) {
						kp = false;
					} else if (two_phase) {
						m.hs = hs;
						const u32 there = m.gate_candidates();
						kp = there != 0;
						if (kp && good >= need) {
							// (a position without a full window -- a contig ends, a non-accepted character is ahead -- keeps
							// its bit on the first question alone; the others go on the list)
							kp = false;
							const u32 at = atomicAdd(&s_n_list, 1u);
							s_list[at] = ((x0 + j) << 4) | there;
						}
					} else if (good >= need) {
						m.hs = hs;
						kp = m.assess_gate(p0 + j, a.seq[p0 + j + k - 1]);
					} else {
						m.hs = m.seed_from_window();
						kp = m.assess_gate_kmer_only();
					}
					keep |= kp ? 1u << j : 0u;
				}
				// on to the next position
				hash_roll(hs, s_tab, s_win[x0 + j], s_win[x0 + j + k]);
				good = s_win[x0 + j + need] != CODE_BAD ? good + 1 : 0;
			}
			// (Probing the candidates of a lane's 16 positions together -- 16 gathers per filter level in flight instead of
			// three or four -- was built and measured: 70 instead of 35 ms per 250 Mbp with -s 1.  The wavefronts in flight
			// already keep the memory system busy; the unrolled 16-wide bookkeeping only added instructions.  Round 5, same
			// question at a smaller scale: the gate alone for TWO positions at a time (6-8 gathers per level) 38.9 against
			// 35.9 ms; 6 / 8 waves per SIMD (80 / 64 registers) 41.1 / 43.5 ms; an instantiation of the machine for the
			// configuration 40.5 ms.  The kernel's TCPs wait for misses 87 % of the time at 29 G fabric requests/s
			// (profiles/r5_assess_counters.txt): what it lacks is not requests in flight.)
		}
		// ---- second phase: the listed positions, one per lane
		if (plain) {
			s_keep[threadIdx.x] = keep;
		}
		__syncthreads();
		if (plain) {
			const u32 n_list = s_n_list;
			for (u32 q = threadIdx.x; q < n_list; q += ASSESS_TPB) {
				const u32 ent = s_list[q];
				const u32 x = ent >> 4;
				MachineT<0> m(env);
				m.win_ok = true;
				m.win_off = x;
				m.hs = m.seed_from_window();
				m.there_known = ent & 15u;
				if (m.assess_gate(base + x, a.seq[base + x + k - 1])) {
					atomicOr(&s_keep[x / ASSESS_L], 1u << (x % ASSESS_L));
				}
			}
			__syncthreads();
			keep = s_keep[threadIdx.x];
			if (threadIdx.x == 0) {
				s_n_list = 0;
			}
		}
		// four lanes = one word
		const u32 k1 = (u32)__shfl_down((int)keep, 1, 64), k2 = (u32)__shfl_down((int)keep, 2, 64), k3 = (u32)__shfl_down((int)keep, 3, 64);
		if ((threadIdx.x & 3) == 0 && p0 < a.pos_end) {
			a.runmap[p0 >> 6] = (u64)keep | ((u64)k1 << 16) | ((u64)k2 << 32) | ((u64)k3 << 48);
		}
	}
}

void
launch_k_assess(unsigned blocks, hipStream_t stream, const AssessArgs& a)
{
	hipLaunchKernelGGL(k_assess, dim3(blocks), dim3(ASSESS_TPB), 0, stream, a);
}

int
assess_tile()
{
	return ASSESS_TILE;
}

} // namespace nte
