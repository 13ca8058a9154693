// Micro-benchmark: how fast can wavefronts that each own 64 consecutive frames (64-byte row
// segments) stream an [N][F] byte matrix, vs a linear read of the same bytes?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// each wave owns G bytes of every row (G = 4*Q); per instruction: 64 lanes x 16 B
template <int Q, int U>
__global__ void __launch_bounds__(64) rows_kernel(const unsigned char *in, size_t stride, int n, u32 groups, u32 *out)
{
	u32 acc = 0;
	int lane = threadIdx.x;
	for (u32 grp = blockIdx.x; grp < groups; grp += gridDim.x) {
		const unsigned char *p = in + (size_t)grp * 4 * Q;
		int nd = (Q * n) / U; // dwords per stream
		for (int t = lane * 4; t < nd; t += 256) {
			u32x4 x[U];
#pragma unroll
			for (int j = 0; j < U; ++j) {
				int tt = t + j * nd, i = tt / Q, q = tt % Q;
				x[j] = *(const u32x4 *)(p + (size_t)i * stride + 4 * q);
			}
#pragma unroll
			for (int j = 0; j < U; ++j)
				acc ^= x[j].x ^ x[j].y ^ x[j].z ^ x[j].w;
		}
	}
	if (acc == 0x12345678)
		out[0] = acc;
}

template <int U>
__global__ void __launch_bounds__(64) linear_kernel(const unsigned char *in, size_t bytes, u32 *out)
{
	u32 acc = 0;
	size_t per = bytes / gridDim.x;
	const u32x4 *p = (const u32x4 *)(in + per * blockIdx.x);
	size_t n4 = per / 16;
	for (size_t t = threadIdx.x; t + 64 * (U - 1) < n4; t += 64 * U) {
		u32x4 x[U];
#pragma unroll
		for (int j = 0; j < U; ++j)
			x[j] = p[t + 64 * j];
#pragma unroll
		for (int j = 0; j < U; ++j)
			acc ^= x[j].x ^ x[j].y ^ x[j].z ^ x[j].w;
	}
	if (acc == 0x12345678)
		out[0] = acc;
}

template <class F> static float timeit(F f)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
	f();
	CK(hipEventRecord(a));
	for (int r = 0; r < 3; ++r) f();
	CK(hipEventRecord(b));
	CK(hipEventSynchronize(b));
	float ms; CK(hipEventElapsedTime(&ms, a, b));
	return ms / 3;
}

int main()
{
	const int n = 65536;
	u32 *out; CK(hipMalloc(&out, 4));
	size_t strides[] = {229376, 262144, 262144 + 256, 1048576, 1048576 + 256, 1048576 + 4096 + 256};
	for (size_t stride : strides) {
		size_t frames = stride > 1048576 ? 1048576 : (stride > 262144 ? 262144 : stride);
		if (stride == 229376) frames = 229376;
		size_t bytes = (size_t)n * stride;
		unsigned char *in; CK(hipMalloc(&in, bytes)); CK(hipMemset(in, 1, bytes));
		double gb = (double)n * frames / 1e9;
		for (int blocks : {1792, 3840}) {
			float a = timeit([&] { hipLaunchKernelGGL((rows_kernel<16, 8>), dim3(blocks), dim3(64), 0, 0, in, stride, n, (u32)(frames / 64), out); });
			float b = timeit([&] { hipLaunchKernelGGL((rows_kernel<16, 2>), dim3(blocks), dim3(64), 0, 0, in, stride, n, (u32)(frames / 64), out); });
			float c = timeit([&] { hipLaunchKernelGGL((rows_kernel<64, 8>), dim3(blocks), dim3(64), 0, 0, in, stride, n, (u32)(frames / 256), out); });
			float d = timeit([&] { hipLaunchKernelGGL((linear_kernel<8>), dim3(blocks), dim3(64), 0, 0, in, (size_t)n * frames, out); });
			printf("stride %8zu frames %8zu blocks %4d: rows G=64 U=8 %.2f TB/s, U=2 %.2f TB/s, G=256 U=8 %.2f TB/s, linear %.2f TB/s\n", stride, frames, blocks, gb / a, gb / b, gb / c, gb / d);
		}
		CK(hipFree(in));
	}
	return 0;
}
