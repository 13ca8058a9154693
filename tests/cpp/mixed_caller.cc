// tests/cpp/mixed_caller.cc -- compile-only: a caller shaped like the reference bench after the
// switch described in INTEGRATION.md: testbench.cc:16-17 (simd.hh, polar_helper.hh) stay, because the
// bench still uses PolarHelper<code_type>::quant (testbench.cc:165) and PolarTransform (:23-42);
// testbench.cc:18-21 (compiler, decoder, encoder, freezer headers) are replaced by polar_amd/polar.hh.
// Built with -fsyntax-only against the reference headers where they lie (tests/test_host_logic.py).
#include <limits>
#include <stdlib.h>
#include <stdint.h>
#include <cassert>
#include <cmath>
#include <algorithm>
#include <functional>
#include "simd.hh"
#include "polar_helper.hh"
#include "polar_amd/polar.hh"

int main()
{
	const int M = 10, N = 1 << M, K = N / 2, FRAMES = 64;
	typedef int8_t code_type;
	typedef GPU<code_type, FRAMES> simd_type; // instead of SIMD<code_type, 32> (testbench.cc:59-65)
	auto frozen = new uint8_t[N];
	auto program = new uint8_t[N];
	PolarCodeConst0<M> freeze;
	freeze(frozen, M, K, std::exp(-1.L));
	PolarCompiler compile;
	int length = compile(program, frozen, M);
	auto codeword = reinterpret_cast<code_type *>(aligned_alloc(64, sizeof(simd_type) * N));
	auto message = reinterpret_cast<code_type *>(aligned_alloc(64, sizeof(simd_type) * K));
	auto decoded = reinterpret_cast<code_type *>(aligned_alloc(64, sizeof(simd_type) * K));
	for (int i = 0; i < FRAMES * K; ++i)
		message[i] = 1 - 2 * (i & 1);
	PolarSysEnc<simd_type, M> sysenc;
	sysenc(reinterpret_cast<simd_type *>(codeword), reinterpret_cast<simd_type *>(message), frozen);
	double fact = 2.0;
	for (int i = 0; i < FRAMES * N; ++i) // the reference's own quantiser on the channel output (testbench.cc:164-165)
		codeword[i] = PolarHelper<code_type>::quant(fact * codeword[i]);
	PolarDecoder<simd_type, M> decode;
	decode(reinterpret_cast<simd_type *>(decoded), reinterpret_cast<simd_type *>(codeword), program);
	PolarEncoder<simd_type, M> encode;
	encode(reinterpret_cast<simd_type *>(codeword), reinterpret_cast<simd_type *>(decoded), frozen);
	return length > 0 ? 0 : 1;
}
