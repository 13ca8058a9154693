/*
 * oracle/ref_shim.cc -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Thin extern "C" wrapper that *includes the unmodified reference headers where they lie*
 * (-I/root/reference, see oracle/Makefile) and exposes the reference functors to ctypes so
 * that (1) the C restatement in oracle/polar_oracle.c can be pinned bit-for-bit against the
 * real thing, (2) tests/golden/ fixtures can be generated, and (3) bench.py can time the
 * genuine reference SIMD path on the GPU box's host cores (cpu_baseline.kind="reference").
 * Only the built oracle/_ref/libpolar_ref.so travels; no reference source is copied here.
 *
 * Reference interfaces wrapped:
 *   PolarCodeConst0<MAX_M>::operator()   polar_freezer.hh:49
 *   PolarFreezer::operator()             polar_freezer.hh:23
 *   PolarCompiler::operator()            polar_compiler.hh:51
 *   PolarEncoder<TYPE,M>::operator()     polar_encoder.hh:15
 *   PolarSysEnc<TYPE,M>::operator()      polar_encoder.hh:36
 *   PolarDecoder<TYPE,MAX_M>::operator() polar_decoder.hh:131
 * with TYPE = SIMD<int8_t, REF_W> (REF_W = 32 under AVX2, testbench.cc:59-65).
 *
 * Buffer layout at this shim's boundary is the reference's own: element-major /
 * frame-minor, buf[frames*i + k] = element i of frame k (testbench.cc:134-139), for any
 * frame count; the shim re-bundles frames into groups of REF_W lanes.
 */
#include <limits>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <cassert>
#include <cmath>
#include <algorithm>
#include <functional>
#include <chrono>
#include "simd.hh"
#include "polar_helper.hh"
#include "polar_compiler.hh"
#include "polar_decoder.hh"
#include "polar_encoder.hh"
#include "polar_freezer.hh"

#ifdef __AVX2__
static const int REF_W = 32;
#else
static const int REF_W = 16;
#endif
typedef SIMD<int8_t, REF_W> simd_type;

template <int M> struct Enc {
	static void run(int8_t *cw, const int8_t *msg, const uint8_t *frozen, int sys)
	{
		if (sys) {
			PolarSysEnc<simd_type, M> e;
			e(reinterpret_cast<simd_type *>(cw), reinterpret_cast<const simd_type *>(msg), frozen);
		} else {
			PolarEncoder<simd_type, M> e;
			e(reinterpret_cast<simd_type *>(cw), reinterpret_cast<const simd_type *>(msg), frozen);
		}
	}
};

static void enc_dispatch(int level, int8_t *cw, const int8_t *msg, const uint8_t *frozen, int sys)
{
	switch (level) {
#define C(M) case M: Enc<M>::run(cw, msg, frozen, sys); break;
	C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17) C(18) C(19) C(20)
#undef C
	default: abort();
	}
}

template <int MAX_M> static void dec_run(void *obj, int8_t *msg, const int8_t *cw, const uint8_t *prog)
{
	(*reinterpret_cast<PolarDecoder<simd_type, MAX_M> *>(obj))(reinterpret_cast<simd_type *>(msg), reinterpret_cast<const simd_type *>(cw), prog);
}

struct RefDecoder {
	int max_m;
	void *obj;
	void (*run)(void *, int8_t *, const int8_t *, const uint8_t *);
};

static void *xaligned(size_t bytes)
{
	size_t a = sizeof(simd_type);
	bytes = (bytes + a - 1) / a * a;
	void *p = aligned_alloc(a, bytes);
	if (!p) abort();
	memset(p, 0, bytes);
	return p;
}

static const int REF_WF = REF_W / 4;
typedef SIMD<float, REF_WF> simd_float;
template <int M> static void dec_run_f32(float *message, const float *codeword, const uint8_t *program)
{
	typedef PolarDecoder<simd_float, M> D;
	D *d = (D *)xaligned(sizeof(D));
	(*d)(reinterpret_cast<simd_float *>(message), reinterpret_cast<const simd_float *>(codeword), program);
	free(d);
}


extern "C" {

int ref_simd_width() { return REF_W; }

void ref_const0(uint8_t *frozen, int level, int K, long double p)
{
	auto f = new PolarCodeConst0<20>;
	(*f)(frozen, level, K, p);
	delete f;
}

/* probability exactly as testbench.cc:74-88 derives it for rate 1/2 */
void ref_const0_testbench(uint8_t *frozen, int level, int K)
{
	long double erasure_probability = 0.5;
	double design_SNR = 10 * std::log10(-std::log(erasure_probability));
	double better_SNR = design_SNR + 1.59175;
	long double probability = std::exp(-pow(10.0, better_SNR / 10));
	ref_const0(frozen, level, K, probability);
}

int ref_freezer(uint8_t *frozen, int level, long double p, long double th)
{
	PolarFreezer f;
	return f(frozen, level, p, th);
}

int ref_compile(uint8_t *program, const uint8_t *frozen, int level)
{
	PolarCompiler c;
	return c(program, frozen, level);
}

/* message: K x frames, codeword: N x frames, both [elem][frame] */
void ref_encode(int8_t *codeword, const int8_t *message, const uint8_t *frozen, int level, int K, long frames, int systematic)
{
	long N = 1L << level;
	int8_t *cw = (int8_t *)xaligned((size_t)N * REF_W);
	int8_t *ms = (int8_t *)xaligned((size_t)(K ? K : 1) * REF_W);
	for (long f0 = 0; f0 < frames; f0 += REF_W) {
		long w = std::min<long>(REF_W, frames - f0);
		for (long i = 0; i < K; ++i)
			for (long k = 0; k < REF_W; ++k)
				ms[i * REF_W + k] = k < w ? message[i * frames + f0 + k] : 1;
		enc_dispatch(level, cw, ms, frozen, systematic);
		for (long i = 0; i < N; ++i)
			for (long k = 0; k < w; ++k)
				codeword[i * frames + f0 + k] = cw[i * REF_W + k];
	}
	free(cw);
	free(ms);
}

void *ref_decoder_create(int level)
{
	RefDecoder *d = new RefDecoder;
#define D(M) if (level <= M) { d->max_m = M; d->obj = xaligned(sizeof(PolarDecoder<simd_type, M>)); d->run = dec_run<M>; return d; }
	D(6) D(8) D(10) D(12) D(14) D(16) D(20)
#undef D
	delete d;
	return nullptr;
}

void ref_decoder_destroy(void *h)
{
	RefDecoder *d = (RefDecoder *)h;
	free(d->obj);
	delete d;
}

/* one reference call on one already-bundled group of REF_W frames ([elem][REF_W], aligned) */
void ref_decode_bundle(void *h, int8_t *message, const int8_t *codeword, const uint8_t *program)
{
	RefDecoder *d = (RefDecoder *)h;
	d->run(d->obj, message, codeword, program);
}

/* message: K x frames out, codeword: N x frames in, [elem][frame]; K = message symbols per frame */
void ref_decode(int8_t *message, const int8_t *codeword, const uint8_t *program, int K, long frames)
{
	int level = program[0];
	long N = 1L << level;
	RefDecoder *d = (RefDecoder *)ref_decoder_create(level);
	int8_t *cw = (int8_t *)xaligned((size_t)N * REF_W);
	int8_t *ms = (int8_t *)xaligned((size_t)N * REF_W);
	for (long f0 = 0; f0 < frames; f0 += REF_W) {
		long w = std::min<long>(REF_W, frames - f0);
		for (long i = 0; i < N; ++i)
			for (long k = 0; k < REF_W; ++k)
				cw[i * REF_W + k] = k < w ? codeword[i * frames + f0 + k] : 0;
		d->run(d->obj, ms, cw, program);
		for (long i = 0; i < K; ++i)
			for (long k = 0; k < w; ++k)
				message[i * frames + f0 + k] = ms[i * REF_W + k];
	}
	free(cw);
	free(ms);
	ref_decoder_destroy(d);
}

/*
 * CPU baseline: time `reps` reference decoder calls (decode only, steady_clock) on one
 * bundle of REF_W frames; returns seconds. Mirrors the timed region testbench.cc:170-175.
 */
double ref_decode_time(const int8_t *codeword_bundle, const uint8_t *program, int reps)
{
	int level = program[0];
	long N = 1L << level;
	RefDecoder *d = (RefDecoder *)ref_decoder_create(level);
	int8_t *cw = (int8_t *)xaligned((size_t)N * REF_W);
	int8_t *ms = (int8_t *)xaligned((size_t)N * REF_W);
	memcpy(cw, codeword_bundle, (size_t)N * REF_W);
	d->run(d->obj, ms, cw, program);
	auto t0 = std::chrono::steady_clock::now();
	for (int r = 0; r < reps; ++r)
		d->run(d->obj, ms, cw, program);
	auto t1 = std::chrono::steady_clock::now();
	volatile int8_t sink = ms[0];
	(void)sink;
	free(cw);
	free(ms);
	ref_decoder_destroy(d);
	return std::chrono::duration<double>(t1 - t0).count();
}

/* single-op tables for per-op KATs (polar_helper.hh:113-173), lane 0 of a bundle */
int ref_op(int which, int a, int b, int c)
{
	typedef PolarHelper<simd_type> PH;
	simd_type A = vdup<simd_type>((int8_t)a), B = vdup<simd_type>((int8_t)b), Cc = vdup<simd_type>((int8_t)c), R;
	switch (which) {
	case 0: R = PH::prod(A, B); break;
	case 1: R = PH::madd(A, B, Cc); break;
	case 2: R = PH::qadd(A, B); break;
	case 3: R = PH::qmul(A, B); break;
	case 4: R = PH::qabs(A); break;
	case 5: R = PH::signum(A); break;
	case 6: R = PH::decide(A); break;
	case 7: R = PH::qmin(A, B); break;
	case 8: R = PH::flip(A, B, Cc, vdup<simd_type>((int8_t)(c >> 8))); break;
	default: abort();
	}
	return R.v[0];
}

int ref_quant(double x)
{
	return PolarHelper<int8_t>::quant(x);
}

/*
 * float decoder: PolarDecoder<SIMD<float, 8>, M> (testbench.cc:49-53 with `typedef float code_type`,
 * AVX2 width 32 / sizeof(float)). message: K x frames, codeword: N x frames floats, [elem][frame].
 */
int ref_decode_f32(float *message, const float *codeword, const uint8_t *program, int K, long frames)
{
	int level = program[0];
	if (level > 16)
		return -1;
	long N = 1L << level;
	float *cw = (float *)xaligned(sizeof(float) * (size_t)N * REF_WF);
	float *ms = (float *)xaligned(sizeof(float) * (size_t)N * REF_WF);
	for (long f0 = 0; f0 < frames; f0 += REF_WF) {
		long w = std::min<long>(REF_WF, frames - f0);
		for (long i = 0; i < N; ++i)
			for (long k = 0; k < REF_WF; ++k)
				cw[i * REF_WF + k] = k < w ? codeword[i * frames + f0 + k] : 0.f;
		if (level <= 8)
			dec_run_f32<8>(ms, cw, program);
		else if (level <= 12)
			dec_run_f32<12>(ms, cw, program);
		else
			dec_run_f32<16>(ms, cw, program);
		for (long i = 0; i < K; ++i)
			for (long k = 0; k < w; ++k)
				message[i * frames + f0 + k] = ms[i * REF_WF + k];
	}
	free(cw);
	free(ms);
	return 0;
}

}
