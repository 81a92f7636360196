// k_machine<false, NTE_CFG>: one thread per event.  Compiled once per machine configuration (Makefile: -DNTE_CFG=...).
#include "nte_machine_kernel.inc"

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

} // namespace nte
