// Micro-benchmark: HBM throughput of streaming kernels by read:write mix (R input streams, W output
// streams, 16-byte non-temporal accesses, persistent single-wave workgroups like the decode kernel).
//   hipcc --offload-arch=gfx950 -O3 -o rw_mix rw_mix.hip && ./rw_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int R, int W>
__global__ void __launch_bounds__(64) mix_kernel(const u32x4 *in, u32x4 *out, size_t per_stream)
{
	// stream s of the kernel = per_stream vectors; block b owns a contiguous slice of every stream
	size_t slice = per_stream / gridDim.x, base = slice * blockIdx.x;
	for (size_t t = threadIdx.x; t + 64 * 3 < slice; t += 64 * 4) {
		u32x4 x[R][4];
#pragma unroll
		for (int r = 0; r < R; ++r)
#pragma unroll
			for (int u = 0; u < 4; ++u)
				x[r][u] = __builtin_nontemporal_load(in + r * per_stream + base + t + 64 * u);
#pragma unroll
		for (int u = 0; u < 4; ++u) {
			u32x4 acc = x[0][u];
#pragma unroll
			for (int r = 1; r < R; ++r)
				acc ^= x[r][u];
#pragma unroll
			for (int w = 0; w < W; ++w) {
				acc.x += w;
				__builtin_nontemporal_store(acc, out + w * per_stream + base + t + 64 * u);
			}
		}
	}
}

template <int R, int W> static void run(const u32x4 *in, u32x4 *out, size_t per_stream)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	for (int blocks : {2048, 4096}) {
		hipLaunchKernelGGL((mix_kernel<R, W>), dim3(blocks), dim3(64), 0, 0, in, out, per_stream);
		CK(hipEventRecord(a));
		for (int k = 0; k < 3; ++k)
			hipLaunchKernelGGL((mix_kernel<R, W>), dim3(blocks), dim3(64), 0, 0, in, out, per_stream);
		CK(hipEventRecord(b));
		CK(hipEventSynchronize(b));
		float ms;
		CK(hipEventElapsedTime(&ms, a, b));
		double bytes = 3.0 * (R + W) * per_stream * 16;
		printf("R:W = %d:%d  %4d waves: %.2f TB/s total (%.2f read, %.2f written)\n", R, W, blocks, bytes / ms / 1e9, bytes / ms / 1e9 * R / (R + W), bytes / ms / 1e9 * W / (R + W));
	}
}

int main()
{
	const size_t per_stream = (size_t)1 << 28; // 4 GiB per stream
	u32x4 *in, *out;
	CK(hipMalloc(&in, 5 * per_stream * 16));
	CK(hipMalloc(&out, 3 * per_stream * 16));
	CK(hipMemset(in, 1, 5 * per_stream * 16));
	run<1, 0>(in, out, per_stream);
	run<1, 1>(in, out, per_stream);
	run<4, 3>(in, out, per_stream);
	run<3, 2>(in, out, per_stream);
	run<2, 1>(in, out, per_stream);
	run<3, 1>(in, out, per_stream);
	run<4, 1>(in, out, per_stream);
	run<5, 1>(in, out, per_stream);
	return 0;
}
