// nte_lanes.h -- the one place where the event machine's source differs between the device build and the host build.
//
// The machine (nte_machine.h and its parts) is ONE source for the HIP kernels and for the test-only host simulation
// (tests/hostsim).  Where a wavefront spreads work over its lanes -- the candidates of an indel sweep, the positions of a
// run of failing positions -- the host build runs the lanes as a loop.  Everything that needs to know which of the two
// it is lives here: group-wide ballots / shuffles / reductions, the per-lane value wrapper and the lane loop macro.
// `group` = the lanes that share one event (EventEnv::wave_size: 1 in the thread-per-event kernel, 64 in the
// wavefront-per-event kernel; the host build always passes 1).
#pragma once
#include "nte_common.h"

namespace nte {

constexpr u32 N_LANES = 64; // positions of a run of failing positions that are assessed together (MachineT::run_lanes)

#if defined(__HIP_DEVICE_COMPILE__)

__device__ __forceinline__ u32
group_lane(u32 group)
{
	return threadIdx.x & (group - 1u);
}

// the lanes of this event's group for which pred holds (a wavefront may carry 64 / group events)
__device__ __forceinline__ u64
group_ballot(bool pred, u32 group)
{
	if (group > 1) {
		const u64 all = __ballot(pred);
		const u32 shift = threadIdx.x & 63u & ~(group - 1u);
		return group >= 64 ? all : ((all >> shift) & ((1ull << group) - 1ull));
	}
	return pred ? 1ull : 0ull;
}

__device__ __forceinline__ u32
group_shfl(u32 v, u32 src, u32 group)
{
	return group > 1 ? (u32)__shfl((int)v, (int)src, (int)group) : v;
}

// one bump of *counter for the whole group (its first lane does it, everybody learns the old value)
__device__ __forceinline__ u32
group_take(u32* counter, u32 group)
{
	if (group > 1) {
		u32 c = 0;
		if ((threadIdx.x & (group - 1u)) == 0) {
			c = atomicAdd(counter, 1u);
		}
		return (u32)__shfl((int)c, 0, (int)group);
	}
	return atomicAdd(counter, 1u);
}

// smallest v over the wavefront (groups of 64 only; a lone lane keeps its own)
__device__ __forceinline__ u32
group_min(u32 v, u32 group)
{
	if (group > 1) {
		for (u32 off = 32; off > 0; off >>= 1) {
			const u32 o = (u32)__shfl_xor((int)v, (int)off, 64);
			v = o < v ? o : v;
		}
	}
	return v;
}

// a value every lane holds for its own position
template<typename T>
struct PerLane
{
	T v;
	__device__ __forceinline__ T& at(u32) { return v; }
};
// body once, for this lane's index l, if l < n
#define NTE_FOR_LANES(l, n) for (u32 l = ::nte::group_lane(64), nte_once_ = 1; nte_once_ && l < (n); nte_once_ = 0)

// value of lane `src`
__device__ __forceinline__ u32
lanes_get(PerLane<u32>& x, u32 src)
{
	return (u32)__shfl((int)x.v, (int)src, 64);
}

__device__ __forceinline__ u64
lanes_get(PerLane<u64>& x, u32 src)
{
	const u32 lo = (u32)__shfl((int)(u32)x.v, (int)src, 64);
	const u32 hi = (u32)__shfl((int)(u32)(x.v >> 32), (int)src, 64);
	return ((u64)hi << 32) | lo;
}

// lowest lane in [from, n) whose value is not zero; n if there is none
__device__ __forceinline__ u32
lanes_first(PerLane<u32>& x, u32 from, u32 n)
{
	const u32 l = group_lane(64);
	const u64 m = __ballot(l >= from && l < n && x.v != 0);
	return m ? (u32)__builtin_ctzll(m) : n;
}

// how many lanes from lane 0 on hold a non-zero value (all N_LANES if every one does)
__device__ __forceinline__ u32
lanes_leading(PerLane<u32>& x)
{
	const u64 m = ~__ballot(x.v != 0);
	return m ? (u32)__builtin_ctzll(m) : N_LANES;
}

// stores of one lane become visible to the others of its wavefront
__device__ __forceinline__ void
lanes_sync()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

#else // ---- host build: one "lane" runs everything, the lanes of a batch of positions are a loop

inline u32
group_lane(u32)
{
	return 0;
}

inline u64
group_ballot(bool pred, u32)
{
	return pred ? 1ull : 0ull;
}

inline u32
group_shfl(u32 v, u32, u32)
{
	return v;
}

inline u32
group_take(u32* counter, u32)
{
	return (*counter)++;
}

inline u32
group_min(u32 v, u32)
{
	return v;
}

template<typename T>
struct PerLane
{
	T v[N_LANES];
	T& at(u32 l) { return v[l]; }
};
#define NTE_FOR_LANES(l, n) for (u32 l = 0; l < (n); l++)

inline u32
lanes_get(PerLane<u32>& x, u32 src)
{
	return x.v[src];
}

inline u64
lanes_get(PerLane<u64>& x, u32 src)
{
	return x.v[src];
}

inline u32
lanes_first(PerLane<u32>& x, u32 from, u32 n)
{
	for (u32 l = from; l < n; l++) {
		if (x.v[l]) {
			return l;
		}
	}
	return n;
}

inline u32
lanes_leading(PerLane<u32>& x)
{
	u32 n = 0;
	while (n < N_LANES && x.v[n]) {
		n++;
	}
	return n;
}

inline void
lanes_sync()
{
}

#endif

} // namespace nte
