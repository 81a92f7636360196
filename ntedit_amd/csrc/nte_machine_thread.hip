// k_machine<false, NTE_CFG>: one thread per event.  Compiled once per machine configuration (Makefile: -DNTE_CFG=...).
#include "nte_machine_kernel.inc"
#include <cstdio>
#include <cstdlib>

#ifndef NTE_CFG
#define NTE_CFG 0
#endif
#define NTE_CAT2(a, b) a##b
#define NTE_CAT(a, b) NTE_CAT2(a, b)

namespace nte {

void
NTE_CAT(launch_k_machine_thread_cfg, NTE_CFG)(unsigned blocks, size_t dyn_lds, hipStream_t stream, const MachineArgs& a)
{
	hipLaunchKernelGGL((k_machine<false, NTE_CFG>), dim3(blocks), dim3(MACHINE_TPB), dyn_lds, stream, a);
}

// filter gathers counted by a -DNTE_PROFILE build (0 otherwise); reading resets the counter
unsigned long long
NTE_CAT(machine_thread_gathers_cfg, NTE_CFG)()
{
	unsigned long long v = 0;
#if defined(NTE_PROFILE)
	unsigned long long all[64];
	unsigned long long zero[64] = { 0 };
	(void)hipMemcpyFromSymbol(all, HIP_SYMBOL(g_prof), sizeof all);
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof zero);
	v = all[15];
#endif
	return v;
}

#if NTE_CFG == 0
// The candidate tables of ntedit.cpp:176-348 as the DEVICE code holds them (MachineT::candidate_bases /
// insertion_candidate run by one GPU thread), raw: out[0 .. 30 * 8): per {snv, letter of "ATCGRYSWKMBDHVN"} the
// count and the candidates; then per index base of "ACGT" 341 entries of 8 bytes {length, characters}.
// ntedit_hip_device_tables() turns them into the text tests/tools/reference_tables.py hashes.
__global__ void
k_tables(u8* out)
{
	const char* letters = "ATCGRYSWKMBDHVN";
	for (int snv = 0; snv < 2; snv++) {
		for (int l = 0; l < 15; l++) {
			u8 cand[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
			const u32 nc = MachineT<0>::candidate_bases((u8)letters[l], snv != 0, cand);
			u8* o = out + (snv * 15 + l) * 8;
			o[0] = (u8)nc;
			for (int q = 0; q < 7; q++) {
				o[1 + q] = cand[q];
			}
		}
	}
	const char* acgt = "ACGT";
	for (int b = 0; b < 4; b++) {
		for (u32 i = 0; i < 341; i++) {
			u8 ins[INDEL_BYTES] = { 0 };
			const u32 m = MachineT<0>::insertion_candidate((u8)acgt[b], i, ins);
			u8* o = out + 30 * 8 + ((size_t)b * 341 + i) * 8;
			o[0] = (u8)m;
			for (u32 q = 0; q < 7; q++) {
				o[1 + q] = q < m ? ins[q] : 0;
			}
		}
	}
}

void
launch_k_tables(hipStream_t stream, u8* out)
{
	hipLaunchKernelGGL(k_tables, dim3(1), dim3(1), 0, stream, out);
}
#endif

// events of more than 2^22 ticks logged by a -DNTE_PROFILE build (nte_machine.h: g_evlog); reading resets the log.
// Returns the entries written to out[4 * cap] ({position, begin tick, end tick, covered | flags << 32}).
unsigned
NTE_CAT(machine_thread_evlog_cfg, NTE_CFG)(unsigned long long* out, unsigned cap)
{
	unsigned n = 0;
#if defined(NTE_PROFILE)
	static unsigned long long all[4 + 4 * NTE_EVLOG_CAP];
	(void)hipMemcpyFromSymbol(all, HIP_SYMBOL(g_evlog), sizeof all);
	unsigned long long zero = 0;
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_evlog), &zero, sizeof zero);
	{
		// (profile build) filter bytes gathered per 4 Mbase region of the batch, printed when NTEDIT_HIP_REGIONS is set
		static unsigned long long reg[NTE_REGIONS], zr[NTE_REGIONS];
		(void)hipMemcpyFromSymbol(reg, HIP_SYMBOL(g_region), sizeof reg);
		(void)hipMemcpyToSymbol(HIP_SYMBOL(g_region), zr, sizeof zr);
		if (getenv("NTEDIT_HIP_REGIONS")) {
			for (unsigned i = 0; i < NTE_REGIONS; i++) {
				if (reg[i] > 2000000ull) {
					fprintf(stderr, "[ntedit_hip] region thread %u: %llu gathers (cfg %d)\n", i, reg[i], (int)NTE_CFG);
				}
			}
		}
	}
	n = (unsigned)(all[0] < NTE_EVLOG_CAP ? all[0] : NTE_EVLOG_CAP);
	n = n < cap ? n : cap;
	for (unsigned i = 0; i < 4 * n; i++) {
		out[i] = all[4 + i];
	}
#else
	(void)out;
	(void)cap;
#endif
	return n;
}

} // namespace nte
