/*
 * polar_hip.h -- C ABI of libpolar_hip.so, the MI355X (gfx950) polar-code path.
 *
 * Drop-in boundary for the hot path of xdsopl/polar: every entry point below replaces one
 * functor call of the reference (cited per function, paths relative to the reference
 * checkout). Plain pointers and sizes only; no C++ or torch types cross this boundary.
 *
 * Buffer layout is the reference's SIMD layout with W = frames: element-major / frame-minor,
 * buf[frames * i + k] = element i of frame k (testbench.cc:134-139, 143-146, 179-182).
 * Symbols: +1 = bit 0, -1 = bit 1, 0 = erasure; LLR sign convention log p(+1)/p(-1)
 * (testbench.cc:160). `frozen` holds one byte (0/1) per code position; `program` is the
 * byte-code of PolarCompiler (polar_compiler.hh:11-13, 51-58).
 *
 * The reference has no error channel (asserts only, Makefile:2); here every call returns
 * POLAR_OK (0) or a negative POLAR_E_* code, and inputs on which the reference would trip
 * an assert are rejected with POLAR_E_PROGRAM / POLAR_E_ARG instead of aborting.
 *
 * Threading: a handle owns device scratch, like the reference functor objects own theirs
 * (polar_decoder.hh:128-129); one call at a time per handle, any number of handles.
 *
 * Stream ordering of POLAR_DEVICE buffers. The synchronous entry points enqueue on the handle's
 * stream and return after that stream has drained, so results are complete on return. Inputs are
 * the caller's business: by default the handle's stream is a private non-blocking stream that is
 * ordered against NOTHING else, so device inputs must be complete (producer stream synchronised)
 * before the call. A caller whose buffers are produced on a stream of its own makes that stream
 * the handle's with polar_hip_set_stream(): the library's work is then ordered behind the
 * producer like any other work on that stream, with no host synchronisation in between.
 * polar_hip_decode_async() always runs on the stream passed to it.
 */
#ifndef POLAR_HIP_H
#define POLAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct polar_hip_handle polar_hip_handle;

enum {
	POLAR_OK = 0,
	POLAR_E_ARG = -1,     /* bad argument (null pointer, level out of range, unknown id) */
	POLAR_E_PROGRAM = -2, /* malformed program / frozen set: the reference would assert */
	POLAR_E_HIP = -3,     /* a HIP runtime call failed, see polar_hip_last_error() */
	POLAR_E_NOMEM = -4,   /* device or host allocation failed */
	POLAR_E_NODEVICE = -5 /* no usable gfx950 device */
};

/* where the data pointers of a call live */
enum { POLAR_HOST = 0, POLAR_DEVICE = 1 };

/* Human-readable text for the last failure on this thread. */
const char *polar_hip_last_error(void);

/* ---- host-side code construction (stays on the CPU, as in the reference) ---- */

/* PolarCodeConst0<MAX_M>::operator() polar_freezer.hh:49-61. Returns POLAR_OK. */
int polar_code_const0(uint8_t *frozen, int level, int K, long double erasure_probability);
/* The same, and *ambiguous (may be NULL) = 1 when the K-th and (K+1)-th most reliable channels
 * have exactly the same erasure probability: the reference leaves their order to
 * std::nth_element (polar_freezer.hh:56), so its frozen set is implementation-defined there;
 * this library gives the lower index the information bit. 0 otherwise. */
int polar_code_const0_ex(uint8_t *frozen, int level, int K, long double erasure_probability, int *ambiguous);
/* `count` constructions of one (level, K), one per erasure probability -- a list of design SNRs
 * (testbench.cc:74-88 per SNR point) -- on the host cores in parallel.
 * frozen: count x N bytes; ambiguous: count ints or NULL. */
int polar_code_const0_batch(uint8_t *frozen, int *ambiguous, int level, int K, const long double *erasure_probability, int count);
/* PolarFreezer::operator() polar_freezer.hh:23-31. Returns K (>= 0). */
int polar_freezer(uint8_t *frozen, int level, long double erasure_probability, long double freezing_threshold);
/* PolarCompiler::operator() polar_compiler.hh:51-58. `program` needs room for 2N+1 bytes in
 * the worst case (N suffices for polar-ordered frozen sets, testbench.cc:95-97). Returns the
 * program length in bytes, or POLAR_E_PROGRAM where the reference asserts (polar_compiler.hh:23). */
int polar_compile(uint8_t *program, const uint8_t *frozen, int level);
/* Number of message symbols per frame a program emits (K), or POLAR_E_PROGRAM. */
int polar_program_rows(const uint8_t *program, int length);

/* ---- device objects ---- */

/* Creates a decoder/encoder context on HIP device `device` (replaces constructing the
 * reference functor objects, testbench.cc:83, 94, 100). */
int polar_hip_create(polar_hip_handle **out, int device);
void polar_hip_destroy(polar_hip_handle *h);

/* Tuning: frames per wavefront (4 ... 256) and highest LLR level staged in LDS -- one of the
 * compiled kernel variants, e.g. (128, 6), (64, 7), (256, 5), (16, 9); POLAR_E_ARG otherwise --
 * and the number of resident wavefronts (0 = fill the device). frames_per_wave = 0 restores the
 * default: chosen per call from the batch size, the code size and the free device memory -- for the
 * whole device, or for max_waves resident wavefronts when that is set: handles that decode side by
 * side on streams of their own (polar_hip_decode_async; mixed batches of several codes) each take a
 * share of the device this way. */
int polar_hip_configure(polar_hip_handle *h, int frames_per_wave, int lds_level, int max_waves);
/* The same with the team size: wavefronts per workgroup that decode neighbouring groups side by
 * side and pace their passes over the input rows (1 = one wavefront per workgroup, what
 * polar_hip_configure selects; 4 is compiled for (32, 8) and (64, 7)). */
int polar_hip_configure_team(polar_hip_handle *h, int frames_per_wave, int lds_level, int team, int max_waves);

/* Stream the synchronous entry points enqueue on: use_callers != 0 selects `stream` (a
 * hipStream_t, NULL = the null stream), 0 restores the handle's private stream. Work already
 * enqueued through the handle is waited for first (it shares the handle's scratch). */
int polar_hip_set_stream(polar_hip_handle *h, void *stream, int use_callers);

/* Uploads a PolarCompiler program as the constant-memory instruction stream; returns a
 * program id >= 0. (polar_decoder.hh:131 takes the program per call; here it is resident.) Loading the
 * same bytes again returns the id of the resident copy. */
int polar_hip_load_program(polar_hip_handle *h, const uint8_t *program, int length);

/* Code cache of the handle: PolarCodeConst0 + PolarCompiler + polar_hip_load_program
 * for (level, K, erasure_probability), done on the first request of that key only (testbench.cc:74-98
 * per design point). Returns the resident program id; the out-pointers (each may be NULL) receive
 * the frozen set (N bytes) and the program, owned by the handle until polar_hip_destroy, the
 * program length, and the ambiguity flag of polar_code_const0_ex. */
int polar_hip_code(polar_hip_handle *h, int level, int K, long double erasure_probability, const uint8_t **frozen, const uint8_t **program,
	int *program_length, int *ambiguous);

/* PolarDecoder<TYPE,MAX_M>::operator()(message, codeword, program) polar_decoder.hh:131-464
 * for `frames` frames at once. codeword: N x frames int8 LLRs (not modified);
 * message: K x frames int8 decisions. `where` says whether both pointers are host or device.
 * Host buffers are staged over PCIe: whole and one after the other when they are pageable; in
 * chunks of about 1 GiB of LLRs (at least 16384 frames), copy-in / decode / copy-out overlapped,
 * when both are page-locked (polar_hip_host_alloc, hipHostMalloc, hipHostRegister) and hold at
 * least two chunks. */
int polar_hip_decode(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, int where);

/* POLAR_DEVICE when `ptr` is device memory (hipMalloc), POLAR_HOST otherwise (pageable or
 * page-locked host memory, NULL): what to pass as `where`. polar.hh uses it so that the functors
 * take buffers on either side. */
int polar_hip_where(const void *ptr);

/* Page-locked host memory for frame buffers, in place of the aligned_alloc of
 * testbench.cc:72,92-93,100-103, so that polar_hip_decode takes the overlapped path above
 * (hipHostMalloc / hipHostFree for callers that do not include HIP headers; 4 KiB aligned).
 * NULL when the memory cannot be had. No counterpart in the reference. */
void *polar_hip_host_alloc(size_t bytes);
void polar_hip_host_free(void *ptr);
/* Same, enqueued on `stream` (a hipStream_t) without synchronising; device pointers only. A
 * handle owns one scratch arena: keep its launches on one stream (any mix of programs).
 * The first launch of a program with a new wavefront shape uploads its instruction stream, and
 * a launch that needs more scratch than the handle holds re-allocates it: both allocate (and
 * the latter waits for the device). Launches after that only enqueue. */
int polar_hip_decode_async(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, void *stream);

/* A set of resident batches of different codes and sizes decoded side by side (Eb/N0 sweeps with
 * per-SNR frozen sets, mixed N = 2^10 .. 2^16 batches; each job = one PolarDecoder::operator() call of
 * testbench.cc:171 on its own program). A batch of a large code holds few frames and is bound by the serial
 * depth of its program: jobs that fill the device on their own are launched one after the other, all the
 * others share one kernel launch whose workgroups are dealt to them so that they finish together; returns
 * when all are done. Device pointers only; program ids as returned by
 * polar_hip_load_program / polar_hip_code on `h`. No counterpart in the reference (one decoder, one call
 * at a time). */
typedef struct polar_hip_job {
	int program_id;
	int8_t *message;
	const int8_t *codeword;
	size_t frames;
	size_t message_pitch, codeword_pitch; /* bytes from row to row, 0 = contiguous (polar_hip_decode_pitched) */
} polar_hip_job;
int polar_hip_decode_set(polar_hip_handle *h, const polar_hip_job *jobs, int count);

/* The float path of the reference: PolarDecoder<SIMD<float,W>,MAX_M>::operator() with
 * PolarHelper<SIMD<VALUE,WIDTH>> (polar_helper.hh:62-111), selected in the bench by
 * `typedef float code_type` (testbench.cc:49-53). codeword: N x frames float LLRs, message:
 * K x frames float decisions (+1 / -1 / 0), same layout as the int8 call. Results equal the
 * reference's AVX2 build bit for bit, signed zeros included; meant for BER comparisons against the
 * int8 decoder (one frame per lane, all state in HBM), not for throughput. */
int polar_hip_decode_f32(polar_hip_handle *h, int program_id, float *message, const float *codeword, size_t frames, int where);

/* PolarEncoder / PolarSysEnc with float lanes (polar_encoder.hh:9-59 is generic in TYPE; the bench
 * selects it with `typedef float code_type`, testbench.cc:49-53). message: K x frames, codeword:
 * N x frames floats. Symbols must be +1, -1 or 0 (what the bench feeds and every decoder returns);
 * they are exact in either type, so the transform runs on the int8 kernels between two conversions
 * and the result equals the reference's products of floats (no -0 arises: PolarHelper<float>::one()
 * is +1 and (+-1) * (+-1) has no zero operand; a 0 symbol gives +0 or -0 in the reference depending on
 * the sign of its partner and +0 here). Anything else is rejected with POLAR_E_ARG. */
int polar_hip_encode_f32(polar_hip_handle *h, int systematic, float *codeword, const float *message, const uint8_t *frozen, int level, size_t frames, int where);

/* No counterpart in the reference: polar_hip_decode_async for callers that cannot allocate --
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc. The
 * wavefront scratch arenas come from a file-scope device array of POLAR_STATIC_SCRATCH_BYTES
 * (64 MiB, one copy per GPU, held until the process ends) instead of the handle's allocated arena,
 * and the call makes no hipMalloc / hipFree / allocating library call: it only enqueues one kernel
 * on `stream`. Limits, each rejected with POLAR_E_ARG and a message:
 *   - frames must be a multiple of 32 and both pointers 16-byte aligned (no staging copies);
 *     a multiple of 128 runs the faster team kernel (four wavefronts per workgroup) when four
 *     arenas fit the array together, anything else one wavefront per workgroup;
 *   - one wavefront's arena, 40 * 2^level bytes (32 frames: LLR levels + two partial-sum planes),
 *     must fit the array: codes up to N = 2^20; the array bounds the resident wavefronts (25 at
 *     N = 2^16), so big codes decode slowly here;
 *   - the program must have been loaded with polar_hip_load_program (its instruction stream for
 *     this entry point's kernel shape is uploaded then).
 * Every call on one GPU shares that GPU's array: calls that overlap on one GPU race (also across
 * handles and streams); serialise them. Nothing in the array has to start at zero: each
 * wavefront writes what it reads, and stale contents of earlier calls are never read. */
int polar_hip_decode_static(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, void *stream);

/* PolarEncoder<TYPE,M>::operator() polar_encoder.hh:15-27 (systematic = 0) and
 * PolarSysEnc<TYPE,M>::operator() polar_encoder.hh:36-58 (systematic = 1).
 * message: K x frames, codeword: N x frames. */
int polar_hip_encode(polar_hip_handle *h, int systematic, int8_t *codeword, const int8_t *message, const uint8_t *frozen, int level, size_t frames, int where);

/* Post-decode step of the systematic bench, testbench.cc:177-183: re-encode the decoded u
 * with the non-systematic encoder and gather the information positions, in place on
 * `decoded` (K x frames). */
int polar_hip_systematic_extract(polar_hip_handle *h, int8_t *decoded, const uint8_t *frozen, int level, size_t frames, int where);

/* The same step with a bit-packed result: instead of K x frames int8 symbols, one
 * bit per symbol in two planes of K rows x 4 * ceil(frames / 32) bytes each (rows padded to whole
 * dwords): sign_plane bit (f % 8) of byte (f / 8) of row k = 1 where the symbol of frame f is -1,
 * erasure_plane (may be NULL) = 1 where it is 0. `decoded` (K x frames int8, the decoder's
 * output) is not modified. 4x fewer bytes than int8 symbols (8x without the erasure plane). */
int polar_hip_systematic_extract_packed(polar_hip_handle *h, uint8_t *sign_plane, uint8_t *erasure_plane, const int8_t *decoded,
	const uint8_t *frozen, int level, size_t frames, int where);
/* Inverse of the packing: symbols[rows][frames] (+1 / -1 / 0) from the two planes. */
int polar_hip_unpack_symbols(polar_hip_handle *h, int8_t *symbols, const uint8_t *sign_plane, const uint8_t *erasure_plane, size_t rows, size_t frames, int where);

/* Bench counters of testbench.cc:185-192 over flat symbol arrays:
 *   counters[0] += #(noisy*orig < 0)  (awgn_errors)      counters[1] += #(noisy == 0)  (quantization_erasures)
 *   counters[2] += #(decoded*message <= 0)  (uncorrected_errors, 0 counts as an error)
 *   counters[3] += #(decoded == 0)  (ambiguity_erasures)
 * With where = POLAR_DEVICE `counters` is a device pointer too (int64[4], added to atomically),
 * ready for the RCCL all-reduce across frame shards. */
int polar_hip_count(polar_hip_handle *h, int64_t *counters, const int8_t *noisy, const int8_t *orig, size_t n_code,
	const int8_t *decoded, const int8_t *message, size_t n_mesg, int where);

/* ---- the bench's message source and channel, moved onto the device ----
 * The reference draws both from std::mt19937 streams (testbench.cc:63-67, 116-117), which cannot
 * be split over GPUs. Here every value is a pure function of (seed, row, global frame index):
 * Philox-4x32-10, key = seed, counter = (frame / 4 low, frame / 4 high, row, stream), stream
 * 0 = noise, 1 = message; lane frame % 4 of the block. Any sharding of the frames
 * [first_frame, first_frame + frames) over calls or GPUs therefore yields the same batch. */

/* testbench.cc:126-127: message[row][f] = 1 - 2 * bit, bit = bit 0 of the frame's Philox word. */
int polar_hip_random_message(polar_hip_handle *h, int8_t *message, size_t rows, size_t frames, uint64_t seed, uint64_t first_frame, int where);

/* testbench.cc:112-114, 151-165: sigma = sqrt(1 / (2 * 10^(snr_db/10))), y = codeword + sigma * n
 * with n ~ N(0,1) by Box-Muller in double from the Philox words (0,1) -> lanes 0,1 and (2,3) ->
 * lanes 2,3; llr = PolarHelper<int8_t>::quant(2 / sigma^2 * y) (polar_helper.hh:194-198).
 * codeword and llr: rows x frames; they may not alias. */
int polar_hip_channel(polar_hip_handle *h, int8_t *llr, const int8_t *codeword, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t first_frame, int where);

/* The channel of the float bench (code_type = float, testbench.cc:49-53, 151-165): llr = (float)(2 / sigma^2 *
 * (codeword + sigma * n)), unquantised (PolarHelper<float>::quant is the identity, polar_helper.hh:28-32), from
 * the same noise draws as polar_hip_channel. codeword: +1 / -1 / 0 as floats (else POLAR_E_ARG); llr
 * may be the same buffer, as in the reference (testbench.cc:164-165). */
int polar_hip_channel_f32(polar_hip_handle *h, float *llr, const float *codeword, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t first_frame, int where);

/* One SNR point of the bench loop, testbench.cc:125-193, for frames [first_frame, first_frame +
 * frames) entirely on the device: random message -> PolarEncoder / PolarSysEnc -> channel ->
 * PolarDecoder -> (systematic: re-encode + gather) -> counters. Adds to counters[0..3] as
 * polar_hip_count does, counters[4] += frames, counters[5] += frames * K (the BER denominator,
 * testbench.cc:197). `where` places `counters` (int64[6]) on the host or the device (ready for
 * the RCCL all-reduce across frame shards); the frame buffers are internal and chunked. */
int polar_hip_simulate(polar_hip_handle *h, int program_id, const uint8_t *frozen, int level, int systematic, double snr_db,
	uint64_t seed, uint64_t first_frame, size_t frames, int64_t *counters, int where);

/* The one exchange of the sharded bench (SURVEY 8 e): frames are independent, every GPU simulates its own
 * shard [first_frame, first_frame + frames) into device counters (polar_hip_simulate / polar_hip_count with
 * POLAR_DEVICE), and the int64 counters of testbench.cc:119-125, 185-192 are summed over the ranks of the
 * caller's RCCL communicator (`nccl_comm` = its ncclComm_t), in place, on the handle's stream; returns
 * when the sums are there. One process per GPU, every rank calls it with the same `count`. The library does
 * not link RCCL: it uses the ncclAllReduce of the RCCL already in the process (the one the communicator came
 * from), else loads librccl.so. No counterpart in the reference (one process, one counter set). */
int polar_hip_allreduce_counters(polar_hip_handle *h, int64_t *counters, size_t count, void *nccl_comm);

/* The same loop with code_type = float (testbench.cc:49-53): same message and noise draws (so both
 * decoders meet the very same frames), LLR = (float)(2 / sigma^2 * y) unquantised (PolarHelper<float>::quant
 * is the identity, polar_helper.hh:28-32), polar_hip_decode_f32, float comparisons in the counters. */
int polar_hip_simulate_f32(polar_hip_handle *h, int program_id, const uint8_t *frozen, int level, int systematic, double snr_db,
	uint64_t seed, uint64_t first_frame, size_t frames, int64_t *counters, int where);

/* Blocks until all work enqueued through this handle has finished. */
int polar_hip_synchronize(polar_hip_handle *h);

/* The kernel shape the last polar_hip_decode / _async / polar_hip_simulate launch on this handle
 * used: frames per wavefront, highest LLR level in LDS, team size (each pointer may be NULL). */
int polar_hip_last_decode_shape(polar_hip_handle *h, int *frames_per_wave, int *lds_level, int *team);

/* polar_hip_decode for device buffers whose rows are `*_pitch` bytes apart instead of `frames` (0 = contiguous):
 * column ranges of wider tensors (a sub-batch without a copy), or rows padded on purpose. Row lengths that are
 * multiples of 4 KiB put every row at the same place of the device memory's address interleaving:
 * polar_hip_row_pitch(frames) is the pitch this library suggests for an input buffer it will decode
 * (frames + 4352 for such lengths: -1.5 % decode time at N = 2^16 with 2^20 frames, else frames rounded up to 16).
 * Pitches that are multiples of 16 are used in place, others staged. The reference has no such notion (one SIMD
 * vector per row, polar_decoder.hh:415). */
int polar_hip_decode_pitched(polar_hip_handle *h, int program_id, int8_t *message, size_t message_pitch,
                             const int8_t *codeword, size_t codeword_pitch, size_t frames);
size_t polar_hip_row_pitch(size_t frames);

/* Kernel launches of the last decode on this handle: 1, or the number of relay launches the batch went out
 * as: two cohorts of half a grid each - one wavefront per SIMD - on the call's stream and on a second stream of
 * the handle that is forked from it and joined to it again, each with its half of the arenas; per segment of
 * about four sweeps one cohort is sent as two launches, the other as one, so that the cohort the SIMDs favour
 * (the older one: it runs 11 % faster) changes in the middle and both finish together (212 instead of 219 ms at
 * N = 2^16 with 2^20 frames, no more memory). Not a default but a measurement: a handle that decodes the same
 * resident batch again and again with polar_hip_decode (the one whose arena placement is chosen by the clock)
 * runs one call that way once the placement stands and keeps it if that call was a percent faster than the
 * best single launch; the pair of streams is tried first (two short kernels must run side by side). Same
 * results either way; POLAR_HIP_RELAY=1 in the environment turns it off, =2 forces it wherever a batch has two
 * sweeps and more. No counterpart in the reference (testbench.cc:170-175 is one loop). */
int polar_hip_last_decode_launches(polar_hip_handle *h);

/* Milliseconds the GPU spent in the decode kernel of the last polar_hip_decode call (first launch's start to
 * last launch's end when there were several), or of the
 * last chunk of the last polar_hip_simulate call (HIP events on the launch stream; the region
 * testbench.cc:170-175 times), or a negative error. */
float polar_hip_last_decode_ms(polar_hip_handle *h);

#ifdef __cplusplus
}
#endif

#endif
