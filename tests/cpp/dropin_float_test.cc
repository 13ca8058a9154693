// tests/cpp/dropin_float_test.cc -- the reference bench's call sequence with `typedef float
// code_type` (testbench.cc:49-53, 83-100, 126-192) written against include/polar_amd/polar.hh with
// TYPE = GPU<float, W>: construct, compile, systematic encode, scaled + flipped LLRs, float decode,
// re-encode, gather, compare; then the same frames through TYPE = GPU<int8_t, W> -- with strong
// LLRs both decoders must return the transmitted message.
// Exit code 0 and "dropin float ok" on success. Needs a gfx950 device.
#include <stdio.h>

#include <vector>

#include "polar_amd/polar.hh"

int main()
{
	const int M = 10, N = 1 << M, K = N / 2, W = 200;
	typedef float code_type;
	typedef GPU<code_type, W> simd_type;
	static_assert(sizeof(simd_type) == W * sizeof(float), "lane bundle layout");
	std::vector<uint8_t> frozen(N), program(2 * N + 1);
	PolarCodeConst0<M> freeze;
	freeze(frozen.data(), M, K, std::exp(-1.L));
	PolarCompiler compile;
	compile(program.data(), frozen.data(), M);
	std::vector<code_type> message(W * K), decoded(W * K), codeword(W * N), orig(W * N);
	unsigned lcg = 7;
	for (auto &m : message)
		m = 1.f - 2.f * (float)((lcg = lcg * 1664525u + 1013904223u) >> 31);
	PolarSysEnc<simd_type, M> sysenc;
	sysenc(reinterpret_cast<simd_type *>(codeword.data()), reinterpret_cast<simd_type *>(message.data()), frozen.data());
	for (int i = 0, j = 0; i < N; ++i)
		for (int k = 0; k < W; ++k) {
			float c = codeword[W * i + k];
			if (c != 1.f && c != -1.f) {
				printf("codeword symbol %g at %d\n", c, W * i + k);
				return 1;
			}
			if (!frozen[i] && c != message[j++]) {
				printf("systematic positions differ\n");
				return 1;
			}
		}
	orig = codeword;
	for (int i = 0; i < W * N; ++i) // strong fractional LLRs, a few sign flips per frame
		codeword[i] = orig[i] * ((i % 89) == 3 ? -1.375f : 5.0625f + (float)(i % 7) * 0.125f);
	PolarDecoder<simd_type, M> decode;
	decode(reinterpret_cast<simd_type *>(decoded.data()), reinterpret_cast<simd_type *>(codeword.data()), program.data());
	std::vector<code_type> again(W * N);
	PolarEncoder<simd_type, M> encode;
	encode(reinterpret_cast<simd_type *>(again.data()), reinterpret_cast<simd_type *>(decoded.data()), frozen.data());
	long errors = 0;
	for (int i = 0, j = 0; i < N; ++i)
		for (int k = 0; k < W; ++k)
			if (!frozen[i])
				errors += again[W * i + k] * message[j++] <= 0;
	printf("float path: uncorrected errors = %ld\n", errors);
	if (errors)
		return 1;
	// the int8 functors on the same frames (quantised as the bench does, polar_helper.hh:194-198)
	typedef GPU<int8_t, W> simd8_type;
	std::vector<int8_t> llr8(W * N), dec8(W * K);
	for (int i = 0; i < W * N; ++i) {
		float v = std::nearbyint(codeword[i] * 8.f);
		llr8[i] = (int8_t)(v > 127.f ? 127.f : v < -128.f ? -128.f : v);
	}
	PolarDecoder<simd8_type, M> decode8;
	decode8(reinterpret_cast<simd8_type *>(dec8.data()), reinterpret_cast<simd8_type *>(llr8.data()), program.data());
	for (int i = 0; i < W * K; ++i)
		if ((float)dec8[i] != decoded[i]) {
			printf("int8 and float decisions differ at %d: %d vs %g\n", i, dec8[i], decoded[i]);
			return 1;
		}
	printf("dropin float ok\n");
	return 0;
}
