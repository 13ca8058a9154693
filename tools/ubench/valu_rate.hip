// Micro-benchmark: issue rate of the decoder's VALU instruction mix (packed 16-bit saturating
// adds, v_perm_b32, v_bfi_b32) for 1, 2, 4 and 8 resident wavefronts per SIMD. Answers "how many
// cycles does one wave64 VALU instruction cost on gfx950, and how many waves does it take to get
// there" -- the VALU-bound floor of the decode kernel depends on it.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// KIND 0: 16 independent chains of v_pk_add_i16 clamp; 1: v_perm_b32; 2: the f kernel (prod4) on 8 independent dwords
template <int KIND>
__global__ void __launch_bounds__(64) rate_kernel(u32 *out, int iters, unsigned long long *cyc)
{
	u32 a[16];
#pragma unroll
	for (int j = 0; j < 16; ++j)
		a[j] = threadIdx.x * 2654435761u + j * 40503u + blockIdx.x;
	u32 k = 0x01000100u + (threadIdx.x & 1);
	unsigned long long t0 = __builtin_amdgcn_s_memtime();
	for (int i = 0; i < iters; ++i) {
#pragma unroll
		for (int r = 0; r < 4; ++r) {
#pragma unroll
			for (int j = 0; j < 16; ++j) {
				if (KIND == 0)
					asm volatile("v_pk_add_i16 %0, %0, %1 clamp" : "+v"(a[j]) : "v"(k));
				else if (KIND == 1)
					asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(k), "v"(0x07030501u));
				else
					asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a[j]) : "v"(k), "v"(0x0F0F0F0Fu));
			}
		}
	}
	unsigned long long t1 = __builtin_amdgcn_s_memtime();
	u32 s = 0;
#pragma unroll
	for (int j = 0; j < 16; ++j)
		s ^= a[j];
	if (s == 0x12345678u)
		out[0] = s;
	if (threadIdx.x == 0 && blockIdx.x == 0)
		cyc[0] = t1 - t0;
}

template <int KIND> static void run(const char *name, int cus)
{
	u32 *out;
	unsigned long long *cyc, h;
	CK(hipMalloc(&out, 4));
	CK(hipMalloc(&cyc, 8));
	const int iters = 20000;
	for (int per_simd : {1, 2, 4, 8}) {
		int blocks = cus * 4 * per_simd;
		hipEvent_t e0, e1;
		CK(hipEventCreate(&e0));
		CK(hipEventCreate(&e1));
		hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, out, 100, cyc);
		CK(hipEventRecord(e0));
		hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(64), 0, 0, out, iters, cyc);
		CK(hipEventRecord(e1));
		CK(hipEventSynchronize(e1));
		float ms;
		CK(hipEventElapsedTime(&ms, e0, e1));
		CK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
		double insts = (double)iters * 64; // per wave
		printf("%-14s %d waves/SIMD: %.3f ms, wave 0: %.2f cycles per instruction (own clock), chip: %.2f G wave-instr/s per SIMD\n", name, per_simd, ms,
			(double)h / insts, insts * per_simd / (ms * 1e-3) / 1e9);
	}
	CK(hipFree(out));
	CK(hipFree(cyc));
}

int main()
{
	hipDeviceProp_t p;
	CK(hipGetDeviceProperties(&p, 0));
	printf("%s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
	run<0>("v_pk_add_i16", p.multiProcessorCount);
	run<1>("v_perm_b32", p.multiProcessorCount);
	run<2>("v_bfi_b32", p.multiProcessorCount);
	return 0;
}
