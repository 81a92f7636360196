// gather_flavors.hip -- micro-benchmark (tools only, not part of the library): how fast can a CU gather single
// bytes/words from an L2-resident buffer, by load flavour?  Decides the inner load of k_bin_probe.
//   hipcc --offload-arch=gfx950 -O3 -o gather_flavors gather_flavors.hip && ./gather_flavors
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { F_U8 = 0, F_U32, F_U8_NT, F_U32_SC0, F_U32_SC1, F_U32_SC01, F_U32_NT, F_ATOMIC_OR, F_ATOMIC_ADD, F_U8_SHARE4, F_U8_SHARE16, N_FLAVORS };
static const char* NAMES[] = { "u8", "u32", "u8 nontemporal", "u32 sc0", "u32 sc1", "u32 sc0 sc1", "u32 nt", "atomic or 0 (rtn)", "atomic add 0 (rtn)", "u8, 4 lanes share a line", "u8, 16 lanes share a line" };

template<int FL>
__device__ __forceinline__ u32
probe(const u8* data, u64 bit)
{
	const u64 byte = bit >> 3;
	if (FL == F_U8 || FL == F_U8_SHARE4 || FL == F_U8_SHARE16) {
		return data[byte];
	} else if (FL == F_U8_NT) {
		return __builtin_nontemporal_load(data + byte);
	} else if (FL == F_U32) {
		return *reinterpret_cast<const u32*>(data + (byte & ~3ULL));
	} else if (FL == F_ATOMIC_OR) {
		return __hip_atomic_fetch_or(reinterpret_cast<u32*>(const_cast<u8*>(data + (byte & ~3ULL))), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	} else if (FL == F_ATOMIC_ADD) {
		return __hip_atomic_fetch_add(reinterpret_cast<u32*>(const_cast<u8*>(data + (byte & ~3ULL))), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	} else {
		const u8* p = data + (byte & ~3ULL);
		u32 v;
		if (FL == F_U32_SC0) {
			asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
		} else if (FL == F_U32_SC1) {
			asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
		} else if (FL == F_U32_SC01) {
			asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
		} else {
			asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
		}
		return v;
	}
}

template<int FL, int DEPTH>
__global__ __launch_bounds__(256) void
k_gather(const u8* __restrict__ data, u64 mask, u64 per_thread, u32* __restrict__ sink)
{
	u64 x = ((u64)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
	const u32 lane = threadIdx.x & 63;
	u32 acc = 0;
	for (u64 it = 0; it < per_thread; it += DEPTH) {
		u64 a[DEPTH];
#pragma unroll
		for (int i = 0; i < DEPTH; i++) {
			x ^= x << 13;
			x ^= x >> 7;
			x ^= x << 17;
			u64 v = (x * 0x2545F4914F6CDD1DULL) & mask;
			if (FL == F_U8_SHARE4 || FL == F_U8_SHARE16) {
				// lanes in groups of 4 / 16 read different bytes of the same 128-byte line
				const int g = FL == F_U8_SHARE4 ? 4 : 16;
				const u64 lead = __shfl(v, lane & ~(g - 1), 64);
				v = ((lead & ~1023ULL) | ((u64)(lane & (g - 1)) * 64 + (v & 63))) & mask;
			}
			a[i] = v;
		}
		u32 b[DEPTH];
#pragma unroll
		for (int i = 0; i < DEPTH; i++) {
			b[i] = probe<FL>(data, a[i]);
		}
		if (FL >= F_U32_SC0 && FL <= F_U32_NT) {
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		}
#pragma unroll
		for (int i = 0; i < DEPTH; i++) {
			acc += (b[i] >> (a[i] & 7)) & 1;
		}
	}
	if (acc == 0xFFFFFFFFu) {
		sink[0] = acc;
	}
}

template<int FL, int DEPTH>
static double
run(const u8* buf, u64 nbytes, u32* sink, int wg_per_cu, int cus, u64 probes)
{
	const u64 threads = (u64)cus * wg_per_cu * 256;
	u64 per_thread = (probes + threads - 1) / threads;
	per_thread = (per_thread + DEPTH - 1) / DEPTH * DEPTH;
	hipEvent_t e0, e1;
	CK(hipEventCreate(&e0));
	CK(hipEventCreate(&e1));
	hipLaunchKernelGGL((k_gather<FL, DEPTH>), dim3((unsigned)(threads / 256)), dim3(256), 0, 0, buf, nbytes * 8 - 1, (u64)DEPTH * 4, sink);
	CK(hipEventRecord(e0, 0));
	hipLaunchKernelGGL((k_gather<FL, DEPTH>), dim3((unsigned)(threads / 256)), dim3(256), 0, 0, buf, nbytes * 8 - 1, per_thread, sink);
	CK(hipEventRecord(e1, 0));
	CK(hipEventSynchronize(e1));
	float ms = 0;
	CK(hipEventElapsedTime(&ms, e0, e1));
	CK(hipEventDestroy(e0));
	CK(hipEventDestroy(e1));
	return (double)(per_thread * threads) / (ms * 1e-3) / 1e9;
}

template<int FL>
static void
flavor(const u8* buf, u32* sink, int cus)
{
	const u64 sizes[] = { 1ULL << 20, 1ULL << 21, 1ULL << 22, 1ULL << 23, 1ULL << 32 };
	for (u64 nbytes : sizes) {
		if ((FL == F_ATOMIC_OR || FL == F_ATOMIC_ADD) && nbytes > (1ULL << 23)) {
			continue;
		}
		const u64 probes = nbytes > (1ULL << 24) ? 2000000000ULL : 6000000000ULL;
		printf("%-28s %8.1f MiB  depth4 x8wg %.1f  depth8 x8wg %.1f  depth12 x8wg %.1f  depth12 x4wg %.1f  depth16 x8wg %.1f  G/s\n", NAMES[FL], nbytes / 1048576.0,
		       run<FL, 4>(buf, nbytes, sink, 8, cus, probes), run<FL, 8>(buf, nbytes, sink, 8, cus, probes), run<FL, 12>(buf, nbytes, sink, 8, cus, probes),
		       run<FL, 12>(buf, nbytes, sink, 4, cus, probes), run<FL, 16>(buf, nbytes, sink, 8, cus, probes));
		fflush(stdout);
	}
}

int
main()
{
	hipDeviceProp_t prop;
	CK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("device %s, %d CUs, L2 %d KiB\n", prop.name, cus, prop.l2CacheSize / 1024);
	u8* buf;
	u32* sink;
	CK(hipMalloc((void**)&buf, 1ULL << 32));
	CK(hipMalloc((void**)&sink, 64));
	CK(hipMemset(buf, 0x5A, 1ULL << 32));
	flavor<F_U8>(buf, sink, cus);
	flavor<F_U32>(buf, sink, cus);
	flavor<F_U8_NT>(buf, sink, cus);
	flavor<F_U32_SC0>(buf, sink, cus);
	flavor<F_U32_SC1>(buf, sink, cus);
	flavor<F_U32_SC01>(buf, sink, cus);
	flavor<F_U32_NT>(buf, sink, cus);
	flavor<F_ATOMIC_OR>(buf, sink, cus);
	flavor<F_ATOMIC_ADD>(buf, sink, cus);
	flavor<F_U8_SHARE4>(buf, sink, cus);
	flavor<F_U8_SHARE16>(buf, sink, cus);
	return 0;
}
