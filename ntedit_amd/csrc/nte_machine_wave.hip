// k_machine<true, NTE_CFG>: one wavefront per event.  Compiled once per machine configuration (Makefile: -DNTE_CFG=...).
#include "nte_machine_kernel.inc"

#ifndef NTE_CFG
#define NTE_CFG 0
#endif
#define NTE_CAT2(a, b) a##b
#define NTE_CAT(a, b) NTE_CAT2(a, b)

namespace nte {

void
NTE_CAT(launch_k_machine_wave_cfg, NTE_CFG)(unsigned blocks, size_t dyn_lds, hipStream_t stream, const MachineArgs& a)
{
	hipLaunchKernelGGL((k_machine<true, NTE_CFG>), dim3(blocks), dim3(MACHINE_TPB), dyn_lds, stream, a);
}

// phase timers (all zero unless built with -DNTE_PROFILE); reading resets them
void
NTE_CAT(machine_wave_profile_cfg, NTE_CFG)(unsigned long long out[64])
{
	for (int i = 0; i < 64; i++) {
		out[i] = 0;
	}
#if defined(NTE_PROFILE)
	unsigned long long zero[64] = { 0 };
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof zero);
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof zero);
#endif
}

#if NTE_CFG == 0
int
machine_wave_group()
{
	return NTE_WAVE_GROUP;
}
#endif

} // namespace nte
