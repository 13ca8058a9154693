// tests/cpp/dropin_device_test.cc -- the drop-in functors of include/polar_amd/polar.hh on buffers
// the caller keeps in device memory (hipMalloc): same call sequence as dropin_test.cc, the
// functors find out where the buffers live (polar_hip_where). Built with hipcc (host code only).
// Exit code 0 and "dropin device ok" on success. Needs a gfx950 device.
#include <hip/hip_runtime.h>
#include <stdio.h>

#include <vector>

#include "polar_amd/polar.hh"

#define HIP_OK(expr) \
	do { \
		hipError_t e_ = (expr); \
		if (e_ != hipSuccess) { \
			printf("%s: %s\n", #expr, hipGetErrorString(e_)); \
			return 1; \
		} \
	} while (0)

int main()
{
	const int M = 12, N = 1 << M, K = N / 2, W = 256;
	typedef GPU<int8_t, W> simd_type;
	std::vector<uint8_t> frozen(N), program(2 * N + 1);
	PolarCodeConst0<M> freeze;
	freeze(frozen.data(), M, K, std::exp(-1.L));
	PolarCompiler compile;
	compile(program.data(), frozen.data(), M);
	std::vector<int8_t> message(W * K), decoded(W * K), codeword(W * N);
	unsigned lcg = 7;
	for (auto &m : message)
		m = 1 - 2 * ((lcg = lcg * 1664525u + 1013904223u) >> 31);
	int8_t *d_msg, *d_cw, *d_dec;
	HIP_OK(hipMalloc((void **)&d_msg, message.size()));
	HIP_OK(hipMalloc((void **)&d_cw, codeword.size()));
	HIP_OK(hipMalloc((void **)&d_dec, decoded.size()));
	if (polar_hip_where(d_cw) != POLAR_DEVICE || polar_hip_where(codeword.data()) != POLAR_HOST) {
		printf("polar_hip_where misjudges a buffer\n");
		return 1;
	}
	HIP_OK(hipMemcpy(d_msg, message.data(), message.size(), hipMemcpyHostToDevice));
	PolarSysEnc<simd_type, M> sysenc;
	sysenc(reinterpret_cast<simd_type *>(d_cw), reinterpret_cast<simd_type *>(d_msg), frozen.data());
	HIP_OK(hipMemcpy(codeword.data(), d_cw, codeword.size(), hipMemcpyDeviceToHost));
	for (int i = 0; i < W * N; ++i) // strong LLRs, a few sign flips per frame
		codeword[i] = (int8_t)(codeword[i] * ((i % 89) == 3 ? -7 : 20));
	HIP_OK(hipMemcpy(d_cw, codeword.data(), codeword.size(), hipMemcpyHostToDevice));
	PolarDecoder<simd_type, M> decode;
	decode(reinterpret_cast<simd_type *>(d_dec), reinterpret_cast<simd_type *>(d_cw), program.data());
	PolarEncoder<simd_type, M> encode; // testbench.cc:177-183: re-encode the decoded u, gather below
	encode(reinterpret_cast<simd_type *>(d_cw), reinterpret_cast<simd_type *>(d_dec), frozen.data());
	HIP_OK(hipMemcpy(codeword.data(), d_cw, codeword.size(), hipMemcpyDeviceToHost));
	long errors = 0;
	for (int i = 0, j = 0; i < N; ++i)
		for (int k = 0; k < W; ++k)
			if (!frozen[i])
				errors += codeword[W * i + k] * message[j++] <= 0;
	printf("uncorrected errors = %ld\n", errors);
	if (errors)
		return 1;
	(void)hipFree(d_msg);
	(void)hipFree(d_cw);
	(void)hipFree(d_dec);
	printf("dropin device ok\n");
	return 0;
}
