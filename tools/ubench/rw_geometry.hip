// Micro-benchmark: 3R:2W streaming throughput by launch geometry -- workgroup size, resident
// wavefronts per CU and 16-byte loads in flight per lane -- to see what the decode kernel's shape
// (8 single-wave workgroups per CU, 16 loads in flight per lane) leaves on the table.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// every wavefront streams its own contiguous slice; U vectors in flight per lane and input stream
template <int U>
__global__ void k(const u32x4 *in, u32x4 *out, size_t per_stream, size_t waves)
{
	const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, lane = threadIdx.x & 63;
	const size_t slice = per_stream / waves, base = slice * wave;
	for (size_t t = lane; t + 64 * (U - 1) < slice; t += 64 * U) {
		u32x4 x[3][U];
#pragma unroll
		for (int r = 0; r < 3; ++r)
#pragma unroll
			for (int u = 0; u < U; ++u)
				x[r][u] = __builtin_nontemporal_load(in + r * per_stream + base + t + 64 * u);
#pragma unroll
		for (int u = 0; u < U; ++u) {
			u32x4 acc = x[0][u] ^ x[1][u] ^ x[2][u];
			__builtin_nontemporal_store(acc, out + base + t + 64 * u);
			acc.x += 1;
			__builtin_nontemporal_store(acc, out + per_stream + base + t + 64 * u);
		}
	}
}

template <int U> static void run(const u32x4 *in, u32x4 *out, size_t per_stream, int block, int waves_per_cu)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	size_t waves = (size_t)256 * waves_per_cu;
	dim3 grid((unsigned)(waves * 64 / block)), blk(block);
	hipLaunchKernelGGL((k<U>), grid, blk, 0, 0, in, out, per_stream, waves);
	CK(hipEventRecord(a));
	for (int i = 0; i < 3; ++i)
		hipLaunchKernelGGL((k<U>), grid, blk, 0, 0, in, out, per_stream, waves);
	CK(hipEventRecord(b));
	CK(hipEventSynchronize(b));
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	printf("block %4d, %2d waves/CU, %2d x 16 B in flight per lane and stream: %.2f TB/s\n", block, waves_per_cu, U, 3.0 * 5 * per_stream * 16 / ms / 1e9);
}

int main()
{
	const size_t per_stream = (size_t)1 << 28;
	u32x4 *in, *out;
	CK(hipMalloc(&in, 3 * per_stream * 16));
	CK(hipMalloc(&out, 2 * per_stream * 16));
	CK(hipMemset(in, 1, 3 * per_stream * 16));
	for (int block : {64, 256})
		for (int wpc : {4, 8, 16, 32}) {
			run<1>(in, out, per_stream, block, wpc);
			run<4>(in, out, per_stream, block, wpc);
			if (wpc <= 16)
				run<8>(in, out, per_stream, block, wpc);
		}
	return 0;
}
