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

} // namespace nte
