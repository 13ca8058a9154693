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
	printf("dropin ok\n");
	return 0;
}
