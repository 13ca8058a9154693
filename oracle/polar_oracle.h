/*
 * oracle/polar_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's int8 SIMD polar-code path (xdsopl/polar).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this;
 * the product (polar_amd/, include/) never links, imports or calls it.
 *
 * Parity status: PINNED. Every function here is checked bit-for-bit against the genuine
 * reference headers compiled in the build container (oracle/ref_shim.cc ->
 * oracle/_ref/libpolar_ref.so) by tests/test_oracle_vs_ref.py, and against the committed
 * golden vectors under tests/golden/ (generated from that same reference build by
 * tests/golden/make_golden.py) by tests/test_oracle_golden.py.
 *
 * Buffer layout everywhere: element-major / frame-minor, buf[frames*i + k] = element i of
 * frame k -- the reference's SIMD<int8_t,W> layout with W = frames (testbench.cc:134-139).
 * Symbols: +1 = bit 0, -1 = bit 1, 0 = erasure.
 */
#ifndef POLAR_ORACLE_H
#define POLAR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* polar_helper.hh:113-173 (PolarHelper<SIMD<int8_t,W>>), one lane */
int8_t oracle_qabs(int8_t a);
int8_t oracle_signum(int8_t a);
int8_t oracle_decide(int8_t a);
int8_t oracle_qmin(int8_t a, int8_t b);
int8_t oracle_qadd(int8_t a, int8_t b);
int8_t oracle_qmul(int8_t a, int8_t b);
int8_t oracle_prod(int8_t a, int8_t b);
int8_t oracle_madd(int8_t h, int8_t a, int8_t b);
int8_t oracle_flip(int8_t a, int8_t b, int8_t c, int8_t d);
/* polar_helper.hh:194-198 */
int8_t oracle_quant(double x);

/* polar_freezer.hh:49-61; returns 0, or 1 if a probability tie straddles the K boundary
 * (then the reference's result depends on std::nth_element internals) */
int oracle_const0(uint8_t *frozen, int level, int K, long double erasure_probability);
/* polar_freezer.hh:23-31; returns K */
int oracle_freezer(uint8_t *frozen, int level, long double erasure_probability, long double freezing_threshold);
/* erasure probability the bench uses for rate 1/2, testbench.cc:74-87 */
long double oracle_testbench_probability(void);

/* polar_compiler.hh:51-58; returns program length in bytes (<= N + 2), or -1 if the
 * reference would trip assert(level > 0) (polar_compiler.hh:23) */
int oracle_compile(uint8_t *program, const uint8_t *frozen, int level);

/* polar_encoder.hh:15-27 (systematic = 0) and :36-58 (systematic = 1) */
void oracle_encode(int8_t *codeword, const int8_t *message, const uint8_t *frozen, int level, size_t frames, int systematic);

/* polar_decoder.hh:131-464; message gets one row per emitted symbol (K rows for a
 * compiled program). Returns the number of message rows written, or -1 on a malformed
 * program (where the reference would assert). */
long oracle_decode(int8_t *message, const int8_t *codeword, const uint8_t *program, size_t frames);

/* testbench.cc:177-183: re-encode decoded u (non-systematic encoder) and gather the
 * info positions; in-place on `decoded` (K x frames). */
void oracle_systematic_extract(int8_t *decoded, const uint8_t *frozen, int level, size_t frames);

/* testbench.cc:185-192 counters; counters[4] += {awgn_errors, quantization_erasures,
 * uncorrected_errors, ambiguity_erasures} */
void oracle_count(int64_t counters[4], const int8_t *noisy, const int8_t *orig, size_t n_code,
	const int8_t *decoded, const int8_t *message, size_t n_mesg);

/* Message source and channel of the bench as include/polar_hip.h defines them for the device
 * (counter-based, so they shard): Philox-4x32-10 restated from the published algorithm (Salmon,
 * Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; Random123 1.14
 * philox.h), pinned by that library's known-answer vectors in tests/test_oracle_golden.py. */
void oracle_philox4x32_10(uint32_t out[4], const uint32_t counter[4], const uint32_t key[2]);
/* testbench.cc:126-127 */
void oracle_random_message(int8_t *message, size_t rows, size_t frames, uint64_t seed, uint64_t first_frame);
/* testbench.cc:112-114, 151-165 with quant of polar_helper.hh:194-198 */
void oracle_channel(int8_t *llr, const int8_t *codeword, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t first_frame);

/* wall-clock seconds for `reps` decode calls on `frames` frames (cpu_baseline "port") */
double oracle_decode_time(const int8_t *codeword, const uint8_t *program, size_t frames, int reps);

#ifdef __cplusplus
}
#endif

/* float decoder, PolarDecoder<SIMD<float,W>,M> semantics (polar_helper.hh:62-111), [elem][frame] floats */
long oracle_decode_f32(float *message, const float *codeword, const uint8_t *program, size_t frames);

#endif
