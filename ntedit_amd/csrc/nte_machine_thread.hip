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
