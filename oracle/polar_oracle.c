/*
 * oracle/polar_oracle.c -- TEST INFRASTRUCTURE ONLY (see polar_oracle.h).
 *
 * Scalar plain-C restatement of the reference's int8 SIMD semantics, written from the
 * specification of each function (file:line cited per function); the frame axis is the
 * innermost loop so gcc can vectorise it, which also makes this usable as a CPU baseline
 * ("port") next to the genuine reference build in oracle/_ref.
 */
#include "polar_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

/* ---- arithmetic policy: PolarHelper<SIMD<int8_t,W>> polar_helper.hh:113-173 ---- */

static inline int clamp8(int x)
{
	return x < -128 ? -128 : x > 127 ? 127 : x;
}

static inline int sgn(int x)
{
	return (x > 0) - (x < 0);
}

/* vqabs simd.hh:292-299: |max(a,-127)| */
static inline int abs8(int a)
{
	if (a < -127)
		a = -127;
	return a < 0 ? -a : a;
}

/* prod polar_helper.hh:153-160: sign(a) sign(b) min(|a|,|b|) with the saturating abs */
static inline int prod8(int a, int b)
{
	int m = abs8(a) < abs8(b) ? abs8(a) : abs8(b);
	return sgn(a) * sgn(b) * m;
}

/* madd polar_helper.hh:161-168: vqadd(vsign(vmax(a,-127), h), b) */
static inline int madd8(int h, int a, int b)
{
	if (a < -127)
		a = -127;
	return clamp8(sgn(h) * a + b);
}

/* qmul on x86 polar_helper.hh:145-152 -> vsign(a,b) = a * sign(b) (simd.hh:1354-1361) */
static inline int qmul8(int a, int b)
{
	return (int8_t)(a * sgn(b));
}

int8_t oracle_qabs(int8_t a) { return (int8_t)abs8(a); }
int8_t oracle_signum(int8_t a) { return (int8_t)sgn(a); }
/* decide polar_helper.hh:129-132: (+1 | (a<0 ? 0xFF : 0)) -> +1 for a>=0, -1 for a<0 */
int8_t oracle_decide(int8_t a) { return a < 0 ? -1 : 1; }
int8_t oracle_qmin(int8_t a, int8_t b) { return a < b ? a : b; }
int8_t oracle_qadd(int8_t a, int8_t b) { return (int8_t)clamp8(a + b); }
int8_t oracle_qmul(int8_t a, int8_t b) { return (int8_t)qmul8(a, b); }
int8_t oracle_prod(int8_t a, int8_t b) { return (int8_t)prod8(a, b); }
int8_t oracle_madd(int8_t h, int8_t a, int8_t b) { return (int8_t)madd8(h, a, b); }
/* flip polar_helper.hh:169-172 */
int8_t oracle_flip(int8_t a, int8_t b, int8_t c, int8_t d) { return c == d ? (int8_t)qmul8(a, b) : a; }

/* quant polar_helper.hh:194-198: clamp(nearbyint(x), -128, 127), default rounding mode */
int8_t oracle_quant(double x)
{
	double r = nearbyint(x);
	if (r < -128.0)
		r = -128.0;
	if (r > 127.0)
		r = 127.0;
	return (int8_t)r;
}

/* ---- code construction: polar_freezer.hh ---- */

/* BEC recursion polar_freezer.hh:13-21 / 37-45: the pe(2-pe) child takes the low half */
static void bec_split(long double *prob, long double pe, long i, long h)
{
	if (!h) {
		prob[i] = pe;
		return;
	}
	bec_split(prob, pe * (2 - pe), i, h / 2);
	bec_split(prob, pe * pe, i + h, h / 2);
}

static const long double *sort_key;

static int by_prob_then_index(const void *pa, const void *pb)
{
	int a = *(const int *)pa, b = *(const int *)pb;
	if (sort_key[a] < sort_key[b])
		return -1;
	if (sort_key[a] > sort_key[b])
		return 1;
	return (a > b) - (a < b);
}

int oracle_const0(uint8_t *frozen, int level, int K, long double erasure_probability)
{
	long n = 1L << level;
	long double *prob = malloc(sizeof(long double) * (size_t)n);
	int *order = malloc(sizeof(int) * (size_t)n);
	bec_split(prob, erasure_probability, 0, n / 2);
	for (long i = 0; i < n; ++i)
		order[i] = (int)i;
	sort_key = prob;
	qsort(order, (size_t)n, sizeof(int), by_prob_then_index);
	/* the K most reliable positions carry information (polar_freezer.hh:56-60) */
	for (long i = 0; i < n; ++i)
		frozen[order[i]] = i >= K;
	int ambiguous = K > 0 && K < n && prob[order[K - 1]] == prob[order[K]];
	free(order);
	free(prob);
	return ambiguous;
}

int oracle_freezer(uint8_t *frozen, int level, long double erasure_probability, long double freezing_threshold)
{
	long n = 1L << level;
	long double *prob = malloc(sizeof(long double) * (size_t)n);
	bec_split(prob, erasure_probability, 0, n / 2);
	int K = 0;
	for (long i = 0; i < n; ++i) {
		frozen[i] = prob[i] > freezing_threshold;
		K += !frozen[i];
	}
	free(prob);
	return K;
}

long double oracle_testbench_probability(void)
{
	/* testbench.cc:74-87, evaluated in the same types */
	long double erasure_probability = 0.5;
	double design_SNR = 10 * log10l(-logl(erasure_probability));
	double better_SNR = design_SNR + 1.59175;
	long double probability = exp(-pow(10.0, better_SNR / 10));
	return probability;
}

/* ---- program compiler: polar_compiler.hh ---- */

enum { OP_LEFT, OP_RIGHT, OP_COMB, OP_RATE0, OP_RATE1, OP_REP, OP_SPC, OP_RATE0_RIGHT, OP_RATE0_COMB, OP_RATE1_COMB, OP_END = 255 };

/* rule order polar_compiler.hh:26-48; cnt[] = prefix sums of frozen[] */
static int emit_node(uint8_t **out, const uint8_t *frozen, const long *cnt, long base, int level)
{
	if (level <= 0)
		return -1;
	long half = 1L << (level - 1);
	long lfro = cnt[base + half] - cnt[base];
	long rfro = cnt[base + 2 * half] - cnt[base + half];
	if (lfro == half && rfro == half) {
		*(*out)++ = OP_RATE0;
	} else if (!lfro && !rfro) {
		*(*out)++ = OP_RATE1;
	} else if (lfro == half && rfro == half - 1 && !frozen[base + 2 * half - 1]) {
		*(*out)++ = OP_REP;
	} else if (lfro == 1 && !rfro && frozen[base]) {
		*(*out)++ = OP_SPC;
	} else if (lfro == half) {
		*(*out)++ = OP_RATE0_RIGHT;
		if (emit_node(out, frozen, cnt, base + half, level - 1))
			return -1;
		*(*out)++ = OP_RATE0_COMB;
	} else if (!rfro) {
		*(*out)++ = OP_LEFT;
		if (emit_node(out, frozen, cnt, base, level - 1))
			return -1;
		*(*out)++ = OP_RATE1_COMB;
	} else {
		*(*out)++ = OP_LEFT;
		if (emit_node(out, frozen, cnt, base, level - 1))
			return -1;
		*(*out)++ = OP_RIGHT;
		if (emit_node(out, frozen, cnt, base + half, level - 1))
			return -1;
		*(*out)++ = OP_COMB;
	}
	return 0;
}

int oracle_compile(uint8_t *program, const uint8_t *frozen, int level)
{
	long n = 1L << level;
	long *cnt = malloc(sizeof(long) * (size_t)(n + 1));
	cnt[0] = 0;
	for (long i = 0; i < n; ++i)
		cnt[i + 1] = cnt[i] + (frozen[i] != 0);
	uint8_t *out = program;
	*out++ = (uint8_t)level;
	int bad = emit_node(&out, frozen, cnt, 0, level);
	*out++ = OP_END;
	free(cnt);
	return bad ? -1 : (int)(out - program);
}

/* ---- encoders: polar_encoder.hh ---- */

/* in-place butterfly x = u G in the multiplicative domain, polar_encoder.hh:17-26 */
static void butterfly(int8_t *x, long n, size_t frames)
{
	for (long h = 1; h < n; h *= 2)
		for (long i = 0; i < n; i += 2 * h)
			for (long j = i; j < i + h; ++j) {
				int8_t *a = x + (size_t)j * frames;
				const int8_t *b = x + (size_t)(j + h) * frames;
				for (size_t k = 0; k < frames; ++k)
					a[k] = (int8_t)qmul8(a[k], b[k]);
			}
}

void oracle_encode(int8_t *codeword, const int8_t *message, const uint8_t *frozen, int level, size_t frames, int systematic)
{
	long n = 1L << level;
	size_t row = 0;
	for (long i = 0; i < n; ++i) {
		int8_t *c = codeword + (size_t)i * frames;
		if (frozen[i])
			memset(c, 1, frames);
		else
			memcpy(c, message + frames * row++, frames);
	}
	butterfly(codeword, n, frames);
	if (!systematic)
		return;
	/* polar_encoder.hh:48-57: re-freeze, transform again */
	for (long i = 0; i < n; ++i)
		if (frozen[i])
			memset(codeword + (size_t)i * frames, 1, frames);
	butterfly(codeword, n, frames);
}

void oracle_systematic_extract(int8_t *decoded, const uint8_t *frozen, int level, size_t frames)
{
	long n = 1L << level;
	int8_t *cw = malloc((size_t)n * frames);
	oracle_encode(cw, decoded, frozen, level, frames, 0);
	size_t row = 0;
	for (long i = 0; i < n; ++i)
		if (!frozen[i])
			memcpy(decoded + frames * row++, cw + (size_t)i * frames, frames);
	free(cw);
}

void oracle_count(int64_t counters[4], const int8_t *noisy, const int8_t *orig, size_t n_code,
	const int8_t *decoded, const int8_t *message, size_t n_mesg)
{
	for (size_t i = 0; i < n_code; ++i) {
		counters[0] += noisy[i] * orig[i] < 0;
		counters[1] += !noisy[i];
	}
	for (size_t i = 0; i < n_mesg; ++i) {
		counters[2] += decoded[i] * message[i] <= 0;
		counters[3] += !decoded[i];
	}
}

/* ---- decoder: polar_decoder.hh ---- */

/* trans polar_decoder.hh:15-27 on `len` rows of FB frames; out may alias inp */
static void transform_rows(int8_t *out, const int8_t *inp, long len, size_t fb)
{
	if (out != inp)
		memcpy(out, inp, (size_t)len * fb);
	butterfly(out, len, fb);
}

#define FB 64 /* frames per block: keeps 3N*FB bytes of scratch cache-friendly */

/* decode one block of fb <= FB frames; cw/msg have row stride `stride` */
static long decode_block(int8_t *msg, const int8_t *cw, const uint8_t *program, size_t stride, size_t fb, int8_t *soft, int8_t *hard, int8_t *tmp)
{
	const uint8_t *pc = program;
	int top = *pc++;
	int lvl = top;
	long n = 1L << top;
	long hrd = 0, out = 0;
	if (top < 1 || top > 30)
		return -1;
#define S(i) (soft + (size_t)(i) * fb)
#define H(i) (hard + (size_t)(hrd + (i)) * fb)
	for (long i = 0; i < n; ++i)
		memcpy(S(n + i), cw + (size_t)i * stride, fb);
	for (;;) {
		int op = *pc++;
		if (op == OP_END)
			break;
		switch (op) {
		case OP_LEFT: { /* polar_decoder.hh:28-35 */
			if (lvl < 2) return -1;
			long len = 1L << lvl, h = len / 2;
			for (long i = 0; i < h; ++i) {
				int8_t *d = S(h + i);
				const int8_t *a = S(len + i), *b = S(len + h + i);
				for (size_t k = 0; k < fb; ++k)
					d[k] = (int8_t)prod8(a[k], b[k]);
			}
			--lvl;
		} break;
		case OP_RIGHT: { /* polar_decoder.hh:36-43; operates on the level above */
			if (lvl + 1 > top) return -1;
			long len = 1L << (lvl + 1), h = len / 2;
			for (long i = 0; i < h; ++i) {
				int8_t *d = S(h + i);
				const int8_t *a = S(len + i), *b = S(len + h + i), *p = H(i);
				for (size_t k = 0; k < fb; ++k)
					d[k] = (int8_t)madd8(p[k], a[k], b[k]);
			}
			hrd += 1L << lvl;
		} break;
		case OP_COMB: { /* polar_decoder.hh:52-59 */
			if (lvl + 1 > top) return -1;
			hrd -= 1L << lvl;
			++lvl;
			long h = 1L << (lvl - 1);
			for (long i = 0; i < h; ++i) {
				int8_t *a = H(i);
				const int8_t *b = H(i + h);
				for (size_t k = 0; k < fb; ++k)
					a[k] = (int8_t)qmul8(a[k], b[k]);
			}
		} break;
		case OP_RATE0: { /* polar_decoder.hh:68-75 */
			long len = 1L << lvl;
			memset(H(0), 1, (size_t)len * fb);
		} break;
		case OP_RATE1: { /* polar_decoder.hh:85-93 */
			long len = 1L << lvl;
			for (long i = 0; i < len; ++i) {
				int8_t *p = H(i);
				const int8_t *a = S(len + i);
				for (size_t k = 0; k < fb; ++k)
					p[k] = (int8_t)sgn(a[k]);
			}
			transform_rows(tmp, H(0), len, fb);
			for (long i = 0; i < len; ++i)
				memcpy(msg + (size_t)(out + i) * stride, tmp + (size_t)i * fb, fb);
			out += len;
		} break;
		case OP_REP: { /* polar_decoder.hh:94-106: pairwise-halving saturating sums */
			long len = 1L << lvl;
			for (long h = len; h; h /= 2)
				for (long i = 0; i < h / 2; ++i) {
					int8_t *d = S(h / 2 + i);
					const int8_t *a = S(h + i), *b = S(h + h / 2 + i);
					for (size_t k = 0; k < fb; ++k)
						d[k] = (int8_t)clamp8(a[k] + b[k]);
				}
			int8_t *m = msg + (size_t)out * stride;
			const int8_t *s1 = S(1);
			for (size_t k = 0; k < fb; ++k)
				m[k] = (int8_t)sgn(s1[k]);
			for (long i = 0; i < len; ++i)
				memcpy(H(i), m, fb);
			++out;
		} break;
		case OP_SPC: { /* polar_decoder.hh:107-127 */
			long len = 1L << lvl;
			int8_t parity[FB], weak[FB];
			for (size_t k = 0; k < fb; ++k) {
				parity[k] = 1;
				weak[k] = 127;
			}
			for (long i = 0; i < len; ++i) {
				int8_t *p = H(i), *m = S(i);
				const int8_t *a = S(len + i);
				for (size_t k = 0; k < fb; ++k) {
					p[k] = a[k] < 0 ? -1 : 1;
					parity[k] = (int8_t)(parity[k] * p[k]);
					m[k] = (int8_t)abs8(a[k]);
					if (m[k] < weak[k])
						weak[k] = m[k];
				}
			}
			/* every position tied for the minimum is multiplied by the parity */
			for (long i = 0; i < len; ++i) {
				int8_t *p = H(i);
				const int8_t *m = S(i);
				for (size_t k = 0; k < fb; ++k)
					if (m[k] == weak[k])
						p[k] = (int8_t)(p[k] * parity[k]);
			}
			transform_rows(S(0), H(0), len, fb);
			for (long i = 0; i < len - 1; ++i)
				memcpy(msg + (size_t)(out + i) * stride, S(i + 1), fb);
			out += len - 1;
		} break;
		case OP_RATE0_RIGHT: { /* polar_decoder.hh:44-51: g with hard == +1, no -127 clamp */
			if (lvl < 2) return -1;
			long len = 1L << lvl, h = len / 2;
			for (long i = 0; i < h; ++i) {
				int8_t *d = S(h + i);
				const int8_t *a = S(len + i), *b = S(len + h + i);
				for (size_t k = 0; k < fb; ++k)
					d[k] = (int8_t)clamp8(a[k] + b[k]);
			}
			--lvl;
			hrd += 1L << lvl;
		} break;
		case OP_RATE0_COMB: { /* polar_decoder.hh:60-67 */
			if (lvl + 1 > top) return -1;
			hrd -= 1L << lvl;
			++lvl;
			long h = 1L << (lvl - 1);
			memcpy(H(0), H(h), (size_t)h * fb);
		} break;
		case OP_RATE1_COMB: { /* polar_decoder.hh:76-84 */
			if (lvl + 1 > top) return -1;
			++lvl;
			long h = 1L << (lvl - 1);
			for (long i = 0; i < h; ++i) {
				int8_t *p = H(i), *q = H(i + h);
				const int8_t *a = S(2 * h + i), *b = S(3 * h + i);
				for (size_t k = 0; k < fb; ++k) {
					q[k] = (int8_t)sgn(madd8(p[k], a[k], b[k]));
					p[k] = (int8_t)qmul8(p[k], q[k]);
				}
			}
			transform_rows(tmp, H(h), h, fb);
			for (long i = 0; i < h; ++i)
				memcpy(msg + (size_t)(out + i) * stride, tmp + (size_t)i * fb, fb);
			out += h;
		} break;
		default:
			return -1;
		}
	}
#undef S
#undef H
	return lvl == top ? out : -1;
}

long oracle_decode(int8_t *message, const int8_t *codeword, const uint8_t *program, size_t frames)
{
	int top = program[0];
	if (top < 1 || top > 30)
		return -1;
	size_t n = (size_t)1 << top;
	int8_t *soft = malloc(2 * n * FB), *hard = malloc(n * FB), *tmp = malloc(n * FB);
	long rows = 0;
	for (size_t f0 = 0; f0 < frames && rows >= 0; f0 += FB) {
		size_t fb = frames - f0 < FB ? frames - f0 : FB;
		rows = decode_block(message + f0, codeword + f0, program, frames, fb, soft, hard, tmp);
	}
	if (!frames) { /* still validate the program and report its row count */
		int8_t *cw = calloc(n, 1), *ms = malloc(n);
		rows = decode_block(ms, cw, program, 1, 1, soft, hard, tmp);
		free(cw);
		free(ms);
	}
	free(soft);
	free(hard);
	free(tmp);
	return rows;
}

double oracle_decode_time(const int8_t *codeword, const uint8_t *program, size_t frames, int reps)
{
	size_t n = (size_t)1 << program[0];
	int8_t *msg = malloc(n * (frames ? frames : 1));
	struct timespec t0, t1;
	oracle_decode(msg, codeword, program, frames);
	clock_gettime(CLOCK_MONOTONIC, &t0);
	for (int r = 0; r < reps; ++r)
		oracle_decode(msg, codeword, program, frames);
	clock_gettime(CLOCK_MONOTONIC, &t1);
	free(msg);
	return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* ---- message source and channel (include/polar_hip.h; testbench.cc:112-117, 126-127, 151-165) ---- */

static uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

void oracle_philox4x32_10(uint32_t out[4], const uint32_t counter[4], const uint32_t key[2])
{
	uint32_t c[4] = {counter[0], counter[1], counter[2], counter[3]}, k[2] = {key[0], key[1]};
	for (int round = 0; round < 10; ++round) {
		if (round) { /* Weyl sequence on the key between rounds */
			k[0] += 0x9E3779B9u;
			k[1] += 0xBB67AE85u;
		}
		uint32_t hi0 = mulhi32(0xD2511F53u, c[0]), lo0 = 0xD2511F53u * c[0];
		uint32_t hi1 = mulhi32(0xCD9E8D57u, c[2]), lo1 = 0xCD9E8D57u * c[2];
		uint32_t n[4] = {hi1 ^ c[1] ^ k[0], lo1, hi0 ^ c[3] ^ k[1], lo0};
		memcpy(c, n, sizeof c);
	}
	memcpy(out, c, sizeof c);
}

static void oracle_draw(uint32_t out[4], uint64_t seed, uint32_t stream, uint32_t row, uint64_t quad)
{
	uint32_t ctr[4] = {(uint32_t)quad, (uint32_t)(quad >> 32), row, stream}, key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
	oracle_philox4x32_10(out, ctr, key);
}

void oracle_random_message(int8_t *message, size_t rows, size_t frames, uint64_t seed, uint64_t first_frame)
{
	for (size_t i = 0; i < rows; ++i)
		for (size_t f = 0; f < frames; ++f) {
			uint64_t g = first_frame + f;
			uint32_t w[4];
			oracle_draw(w, seed, 1, (uint32_t)i, g >> 2);
			message[i * frames + f] = (int8_t)(1 - 2 * (int)(w[g & 3] & 1));
		}
}

void oracle_channel(int8_t *llr, const int8_t *codeword, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t first_frame)
{
	double sigma = sqrt(1.0 / (2.0 * pow(10.0, snr_db / 10.0))); /* testbench.cc:112-114 */
	double fact = 2.0 / (sigma * sigma);                            /* testbench.cc:162-163 */
	for (size_t i = 0; i < rows; ++i)
		for (size_t f = 0; f < frames; ++f) {
			uint64_t g = first_frame + f;
			uint32_t w[4];
			oracle_draw(w, seed, 0, (uint32_t)i, g >> 2);
			int pair = (int)(g & 2);
			double u1 = ((double)w[pair] + 0.5) / 4294967296.0, u2 = ((double)w[pair + 1] + 0.5) / 4294967296.0;
			double r = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925286766559 * u2;
			double n = (g & 1) ? r * sin(t) : r * cos(t);
			double y = (double)codeword[i * frames + f] + sigma * n; /* testbench.cc:154-158 */
			llr[i * frames + f] = oracle_quant(fact * y);           /* testbench.cc:164-165 */
		}
}

/* the float bench's channel (code_type = float, testbench.cc:49-53): the same draws, quant = identity
 * (polar_helper.hh:28-32), the double product narrowed to float on assignment (testbench.cc:164-165) */
void oracle_channel_f32(float *llr, const int8_t *codeword, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t first_frame)
{
	double sigma = sqrt(1.0 / (2.0 * pow(10.0, snr_db / 10.0)));
	double fact = 2.0 / (sigma * sigma);
	for (size_t i = 0; i < rows; ++i)
		for (size_t f = 0; f < frames; ++f) {
			uint64_t g = first_frame + f;
			uint32_t w[4];
			oracle_draw(w, seed, 0, (uint32_t)i, g >> 2);
			int pair = (int)(g & 2);
			double u1 = ((double)w[pair] + 0.5) / 4294967296.0, u2 = ((double)w[pair + 1] + 0.5) / 4294967296.0;
			double r = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925286766559 * u2;
			double n = (g & 1) ? r * sin(t) : r * cos(t);
			llr[i * frames + f] = (float)(fact * ((double)codeword[i * frames + f] + sigma * n));
		}
}

/* ---- float decoder: PolarDecoder<SIMD<float,W>,M> with PolarHelper<SIMD<VALUE,WIDTH>>
 *      (polar_helper.hh:62-111), the instantiation the reference bench selects with
 *      `typedef float code_type` (testbench.cc:49-53). TEST INFRASTRUCTURE ONLY.
 *      Per lane, in the AVX2 forms of the reference (avx2.hh:514-521 vsignum: +0 for +-0,
 *      else copysign(1, a); avx2.hh:602-610 vcopysign; vmin(a, b) = a < b ? a : b). Signed
 *      zeros matter: prod can yield -0, and decide(-0) = -1. ---- */

static float f_signum(float a) { return a == 0.0f ? 0.0f : copysignf(1.0f, a); }
static float f_decide(float a) { return copysignf(1.0f, a); }
static float f_min(float a, float b) { return a < b ? a : b; }
static float f_prod(float a, float b)
{
	volatile float s = f_signum(a) * f_signum(b); /* vmul(vmul(sa, sb), vmin(|a|, |b|)), polar_helper.hh:98-101 */
	return s * f_min(fabsf(a), fabsf(b));
}
static float f_madd(float h, float a, float b)
{
	volatile float p = h * a; /* vadd(vmul(a, b), c): two roundings, polar_helper.hh:102-105 */
	return p + b;
}

static void f_trans(float *out, const float *inp, int length)
{
	for (int i = 0; i < length; i += 2) {
		out[i] = inp[i] * inp[i + 1];
		out[i + 1] = inp[i + 1];
	}
	for (int h = 2; h < length; h *= 2)
		for (int i = 0; i < length; i += 2 * h)
			for (int j = i; j < i + h; ++j)
				out[j] = out[j] * out[j + h];
}

/* one frame; message: rows out (stride `stride` floats), codeword: N in (stride `stride`) */
static long f_decode_frame(float *message, const float *codeword, const uint8_t *program, size_t stride, float *soft, float *hard, float *tmp)
{
	int top = program[0], lvl = top;
	size_t n = (size_t)1 << top;
	float *hrd = hard;
	long row = 0;
	for (size_t i = 0; i < n; ++i)
		soft[n + i] = codeword[i * stride];
	for (const uint8_t *pc = program + 1; *pc != 255; ++pc) {
		int length;
		switch (*pc) {
		case 0: /* left, polar_decoder.hh:28-35 */
			length = 1 << lvl;
			for (int i = 0; i < length / 2; ++i)
				soft[i + length / 2] = f_prod(soft[i + length], soft[i + length / 2 + length]);
			--lvl;
			break;
		case 1: /* right :36-43 */
			length = 1 << (lvl + 1);
			for (int i = 0; i < length / 2; ++i)
				soft[i + length / 2] = f_madd(hrd[i], soft[i + length], soft[i + length / 2 + length]);
			hrd += 1 << lvl;
			break;
		case 2: /* comb :52-59 */
			hrd -= 1 << lvl;
			++lvl;
			length = 1 << lvl;
			for (int i = 0; i < length / 2; ++i)
				hrd[i] = hrd[i] * hrd[i + length / 2];
			break;
		case 3: /* rate0 :68-75 */
			for (int i = 0; i < (1 << lvl); ++i)
				hrd[i] = 1.0f;
			break;
		case 4: /* rate1 :85-93 */
			length = 1 << lvl;
			for (int i = 0; i < length; ++i)
				hrd[i] = f_signum(soft[i + length]);
			f_trans(tmp, hrd, length);
			for (int i = 0; i < length; ++i)
				message[(size_t)(row + i) * stride] = tmp[i];
			row += length;
			break;
		case 5: /* rep :94-106 */
			length = 1 << lvl;
			for (int h = length; h; h /= 2)
				for (int i = 0; i < h / 2; ++i)
					soft[i + h / 2] = soft[i + h] + soft[i + h / 2 + h];
			{
				float d = f_signum(soft[1]);
				message[(size_t)row * stride] = d;
				for (int i = 0; i < length; ++i)
					hrd[i] = d;
			}
			++row;
			break;
		case 6: /* spc :107-127 */
			length = 1 << lvl;
			{
				for (int i = 0; i < length; ++i)
					hrd[i] = f_decide(soft[i + length]);
				float parity = hrd[0];
				for (int i = 1; i < length; ++i)
					parity = parity * hrd[i];
				for (int i = 0; i < length; ++i)
					soft[i] = fabsf(soft[i + length]);
				float weak = soft[0];
				for (int i = 1; i < length; ++i)
					weak = f_min(weak, soft[i]);
				for (int i = 0; i < length; ++i)
					hrd[i] = weak == soft[i] ? hrd[i] * parity : hrd[i];
				f_trans(soft, hrd, length);
				for (int i = 0; i < length - 1; ++i)
					message[(size_t)(row + i) * stride] = soft[i + 1];
			}
			row += length - 1;
			break;
		case 7: /* rate0_right :44-51 */
			length = 1 << lvl;
			for (int i = 0; i < length / 2; ++i)
				soft[i + length / 2] = soft[i + length] + soft[i + length / 2 + length];
			--lvl;
			hrd += 1 << lvl;
			break;
		case 8: /* rate0_comb :60-67 */
			hrd -= 1 << lvl;
			++lvl;
			length = 1 << lvl;
			for (int i = 0; i < length / 2; ++i)
				hrd[i] = hrd[i + length / 2];
			break;
		case 9: /* rate1_comb :76-84 */
			++lvl;
			length = 1 << (lvl - 1);
			for (int i = 0; i < length; ++i) {
				float r = f_signum(f_madd(hrd[i], soft[i + 2 * length], soft[i + 3 * length]));
				hrd[i + length] = r;
				hrd[i] = hrd[i] * r;
			}
			f_trans(tmp, hrd + length, length);
			for (int i = 0; i < length; ++i)
				message[(size_t)(row + i) * stride] = tmp[i];
			row += length;
			break;
		default:
			return -1;
		}
	}
	return lvl == top ? row : -1;
}

/* message: rows x frames out, codeword: N x frames in, [elem][frame] floats. Returns rows. */
long oracle_decode_f32(float *message, const float *codeword, const uint8_t *program, size_t frames)
{
	int top = program[0];
	if (top < 1 || top > 24)
		return -1;
	size_t n = (size_t)1 << top;
	float *soft = malloc(2 * n * sizeof(float)), *hard = malloc(n * sizeof(float)), *tmp = malloc(n * sizeof(float));
	long rows = 0;
	for (size_t f = 0; f < frames && rows >= 0; ++f)
		rows = f_decode_frame(message + f, codeword + f, program, frames, soft, hard, tmp);
	free(soft);
	free(hard);
	free(tmp);
	return rows;
}
