// Micro-benchmark: does the HBM ceiling of a 3R:2W streaming kernel depend on how the wavefronts'
// streams are laid out? (a) every wavefront owns a contiguous slice of each stream (the decode
// kernel's per-wave arenas), (b) the wavefronts interleave at 4 KiB granularity (level-major
// layout), each with non-temporal or default-policy accesses.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4 *p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// R = 3 input streams, W = 2 output streams; chunk = 64 lanes x 4 vectors = 4 KiB per stream per iteration
template <bool NT, bool INTERLEAVED>
__global__ void __launch_bounds__(64) k(const u32x4 *in, u32x4 *out, size_t per_stream)
{
	const size_t chunks = per_stream / 256, nb = gridDim.x, b = blockIdx.x, mine = chunks / nb;
	for (size_t c = 0; c < mine; ++c) {
		size_t chunk = INTERLEAVED ? c * nb + b : b * mine + c;
		size_t t = chunk * 256 + threadIdx.x;
		u32x4 x[3][4];
#pragma unroll
		for (int r = 0; r < 3; ++r)
#pragma unroll
			for (int u = 0; u < 4; ++u)
				x[r][u] = ld<NT>(in + r * per_stream + t + 64 * u);
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			u32x4 acc = x[0][u] ^ x[1][u] ^ x[2][u];
			st<NT>(out + t + 64 * u, acc);
			acc.x += 1;
			st<NT>(out + per_stream + t + 64 * u, acc);
		}
	}
}

template <bool NT, bool IL> static void run(const char *name, const u32x4 *in, u32x4 *out, size_t per_stream)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	hipLaunchKernelGGL((k<NT, IL>), dim3(2048), dim3(64), 0, 0, in, out, per_stream);
	CK(hipEventRecord(a));
	for (int i = 0; i < 3; ++i)
		hipLaunchKernelGGL((k<NT, IL>), dim3(2048), dim3(64), 0, 0, in, out, per_stream);
	CK(hipEventRecord(b));
	CK(hipEventSynchronize(b));
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	printf("%-34s %.2f TB/s\n", name, 3.0 * 5 * per_stream * 16 / ms / 1e9);
}

int main()
{
	const size_t per_stream = (size_t)1 << 28; // 4 GiB
	u32x4 *in, *out;
	CK(hipMalloc(&in, 3 * per_stream * 16));
	CK(hipMalloc(&out, 2 * per_stream * 16));
	CK(hipMemset(in, 1, 3 * per_stream * 16));
	run<true, false>("contiguous per wave, nt", in, out, per_stream);
	run<false, false>("contiguous per wave, default", in, out, per_stream);
	run<true, true>("interleaved at 4 KiB, nt", in, out, per_stream);
	run<false, true>("interleaved at 4 KiB, default", in, out, per_stream);
	return 0;
}
