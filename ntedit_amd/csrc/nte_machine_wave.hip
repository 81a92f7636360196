// k_machine<true>: one wavefront per event.
#include "nte_machine_kernel.inc"

namespace nte {

void
launch_k_machine_wave(unsigned blocks, size_t dyn_lds, hipStream_t stream, const MachineArgs& a)
{
	hipLaunchKernelGGL(k_machine<true>, dim3(blocks), dim3(MACHINE_TPB), dyn_lds, stream, a);
}

int
machine_wave_group()
{
	return NTE_WAVE_GROUP;
}

} // namespace nte
