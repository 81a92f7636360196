// k_machine<true>: one wavefront per event.
#include "nte_machine_kernel.inc"

namespace nte {

void
launch_k_machine_wave(unsigned blocks, size_t dyn_lds, hipStream_t stream, const MachineArgs& a)
{
	hipLaunchKernelGGL(k_machine<true>, dim3(blocks), dim3(MACHINE_TPB), dyn_lds, stream, a);
}

// phase timers (all zero unless built with -DNTE_PROFILE); reading resets them
void
machine_wave_profile(unsigned long long out[24])
{
	for (int i = 0; i < 24; i++) {
		out[i] = 0;
	}
#if defined(NTE_PROFILE)
	unsigned long long zero[24] = { 0 };
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof zero);
	(void)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), zero, sizeof zero);
#endif
}

int
machine_wave_group()
{
	return NTE_WAVE_GROUP;
}

} // namespace nte
