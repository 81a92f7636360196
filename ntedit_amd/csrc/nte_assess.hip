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
// positions the window cannot serve (a non-accepted character within k + max_del + 1 bases behind the k-mer, the
// end of a contig, bytes the 4-bit codes cannot express) keep their bit.
#include "nte_machine_launch.h"

#include <hip/hip_runtime.h>

namespace nte {

constexpr int ASSESS_TPB = 256;
constexpr int ASSESS_MAX_WIN = ASSESS_TPB + 2 * 200 + 10 + 1 + 16; // k <= 200, max_deletions <= 10 (params.cpp)

__global__ __launch_bounds__(ASSESS_TPB) void
k_assess(AssessArgs a)
{
	__shared__ u64 s_tab[TAB_WORDS];
	__shared__ __attribute__((aligned(16))) u8 s_win[ASSESS_MAX_WIN];
	__shared__ u32 s_exotic;
	if (threadIdx.x < TAB_WORDS) {
		s_tab[threadIdx.x] = a.tabs[threadIdx.x];
	}
	const u32 K = a.p.k + a.p.max_deletions + 1; // Machine::win_len_in()
	const u32 span = ASSESS_TPB + a.p.k + K;     // codes a tile's lanes read
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
	for (u64 tile = blockIdx.x; tile < a.n_tiles; tile += gridDim.x) {
		const u64 base = a.pos_begin + tile * ASSESS_TPB; // (a multiple of 64)
		__syncthreads();
		if (threadIdx.x == 0) {
			s_exotic = 0;
		}
		__syncthreads();
		// does the tile hold an absent k-mer at all?
		const bool in_range = base + (threadIdx.x & ~63u) < a.pos_end; // (this wavefront's bitmap word exists)
		const u64 word = in_range ? a.bitmap[(base >> 6) + (threadIdx.x >> 6)] : 0;
		const bool mine = base + threadIdx.x < a.pos_end && ((word >> (threadIdx.x & 63)) & 1);
		if (!__syncthreads_or(mine ? 1 : 0)) {
			if ((threadIdx.x & 63) == 0 && in_range) {
				a.runmap[(base >> 6) + (threadIdx.x >> 6)] = 0;
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
		if (exotic) {
			s_exotic = 1; // (benign race: every writer stores 1)
		}
		__syncthreads();
		bool keep = mine;
		if (mine && !s_exotic) {
			// the window of this lane: k + K accepted codes from its offset on, or the bit stays
			bool clear = true;
			for (u32 i = 0; i < a.p.k + K; i++) {
				clear = clear && s_win[threadIdx.x + i] != CODE_BAD;
			}
			if (clear) {
				MachineT<0> m(env);
				m.win_off = threadIdx.x;
				m.win_ok = true;
				m.hs = m.seed_from_window();
				keep = m.assess_gate(base + threadIdx.x, a.seq[base + threadIdx.x + a.p.k - 1]);
			}
		}
		const u64 out = __ballot(keep);
		if ((threadIdx.x & 63) == 0 && in_range) {
			a.runmap[(base >> 6) + (threadIdx.x >> 6)] = out;
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
	return ASSESS_TPB;
}

} // namespace nte
