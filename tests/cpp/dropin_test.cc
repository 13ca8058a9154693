// tests/cpp/dropin_test.cc -- the reference bench's call sequence (testbench.cc:83-100, 126-192)
// written against include/polar_amd/polar.hh with TYPE = GPU<int8_t, W>: construct, compile,
// systematic encode, noiseless + flipped-sign LLRs, decode, re-encode, gather, compare.
// Exit code 0 and "dropin ok" on success. Needs a gfx950 device.
#include <stdio.h>
#include <string.h>

#include <vector>

#include "polar_amd/polar.hh"

int main()
{
	const int M = 11, N = 1 << M, K = N / 2, W = 72;
	typedef GPU<int8_t, W> simd_type;
	std::vector<uint8_t> frozen(N), program(2 * N + 1);
	PolarCodeConst0<M> freeze;
	freeze(frozen.data(), M, K, std::exp(-1.L));
	PolarCompiler compile;
	int length = compile(program.data(), frozen.data(), M);
	std::vector<int8_t> message(W * K), decoded(W * K), codeword(W * N), orig(W * N);
	unsigned lcg = 1;
	for (auto &m : message)
		m = 1 - 2 * ((lcg = lcg * 1664525u + 1013904223u) >> 31);
	PolarSysEnc<simd_type, M> sysenc;
	sysenc(reinterpret_cast<simd_type *>(codeword.data()), reinterpret_cast<simd_type *>(message.data()), frozen.data());
	for (int i = 0, j = 0; i < N; ++i)
		for (int k = 0; k < W; ++k)
			if (!frozen[i] && codeword[W * i + k] != message[j++]) {
				printf("systematic positions differ\n");
				return 1;
			}
	orig = codeword;
	for (int i = 0; i < W * N; ++i) // strong LLRs, a few sign flips per frame
		codeword[i] = (int8_t)(orig[i] * ((i % 97) == 5 ? -9 : 24));
	PolarDecoder<simd_type, M> decode;
	decode(reinterpret_cast<simd_type *>(decoded.data()), reinterpret_cast<simd_type *>(codeword.data()), program.data());
	PolarEncoder<simd_type, M> encode;
	encode(reinterpret_cast<simd_type *>(codeword.data()), reinterpret_cast<simd_type *>(decoded.data()), frozen.data());
	long errors = 0;
	for (int i = 0, j = 0; i < N; ++i)
		for (int k = 0; k < W; ++k)
			if (!frozen[i])
				errors += codeword[W * i + k] * message[j++] <= 0;
	printf("program length = %d, uncorrected errors = %ld\n", length, errors);
	if (errors)
		return 1;
	// the same frames, tiled over page-locked buffers (instead of the bench's aligned_alloc,
	// testbench.cc:72,92-93): several chunks' worth, so the call overlaps its PCIe copies with the
	// decoding; every frame must decode as its twin did above
	setenv("POLAR_HIP_HOST_CHUNK", "1024", 1); // tests only: chunks of 1024 frames instead of 1 GiB
	const int W2 = 3 * 1024 + 40;
	typedef GPU<int8_t, W2> wide_type;
	for (int i = 0; i < W * N; ++i)
		codeword[i] = (int8_t)(orig[i] * ((i % 97) == 5 ? -9 : 24));
	decode(reinterpret_cast<simd_type *>(decoded.data()), reinterpret_cast<simd_type *>(codeword.data()), program.data());
	int8_t *llr2 = (int8_t *)polar_hip_host_alloc((size_t)W2 * N), *dec2 = (int8_t *)polar_hip_host_alloc((size_t)W2 * K);
	if (!llr2 || !dec2) {
		printf("polar_hip_host_alloc failed: %s\n", polar_hip_last_error());
		return 1;
	}
	for (int i = 0; i < N; ++i)
		for (int k = 0; k < W2; ++k)
			llr2[(size_t)W2 * i + k] = codeword[W * i + k % W];
	memset(dec2, 0, (size_t)W2 * K);
	PolarDecoder<wide_type, M> decode_wide;
	decode_wide(reinterpret_cast<wide_type *>(dec2), reinterpret_cast<wide_type *>(llr2), program.data());
	for (int j = 0; j < K; ++j)
		for (int k = 0; k < W2; ++k)
			if (dec2[(size_t)W2 * j + k] != decoded[W * j + k % W]) {
				printf("page-locked batch differs at row %d frame %d\n", j, k);
				return 1;
			}
	polar_hip_host_free(llr2);
	polar_hip_host_free(dec2);
	printf("dropin ok\n");
	return 0;
}
