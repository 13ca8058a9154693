// polar_hip.hip -- kernels and C ABI of libpolar_hip.so (gfx950 only).
//
//   sc_decode_kernel        : batched SC decoder, one wavefront per group of 4*Q frames (decode_kernel.hpp,
//                             sc_core.hpp, variants.hip; replaces PolarDecoder::operator(), polar_decoder.hh:131)
//   sc_decode_team_kernel   : the same with four wavefronts per workgroup that share the input lines (32- and 64-frame groups)
//   sc_decode_f32_kernel    : the reference's float helper per lane (polar_helper.hh:62-111)
//   packed_tile_kernel      : polar transform on bit-packed symbols through LDS tiles: the encoders (polar_encoder.hh:15-58)
//                             and the systematic extraction = re-encode + gather (testbench.cc:177-183)
//   packed_transform_kernel : the same in registers only (codes below N = 2^6)
//   butterfly_kernel        : transform stages on byte symbols (encoders on ragged batches)
//   channel_kernel, count_kernel : the bench's message source, AWGN channel and error counters (testbench.cc:112-192)
//
// Everything here fails loudly: there is no CPU fallback for any device entry point.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/polar_hip.h"
#include "construct.hpp"
#include "program.hpp"
#include "channel.hpp"
#include "decode_kernel.hpp"

using namespace polar;

static thread_local std::string g_error;

static int fail(int code, const std::string &msg)
{
	g_error = msg;
	return code;
}

#define HIP_TRY(expr) \
	do { \
		hipError_t e_ = (expr); \
		if (e_ != hipSuccess) \
			return fail(e_ == hipErrorOutOfMemory ? POLAR_E_NOMEM : POLAR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

// ---------------------------------------------------------------------------------------
// decoder kernel
// ---------------------------------------------------------------------------------------

// all compiled decode-kernel variants (variants.hip, one group per translation unit)
struct VariantTable {
	std::vector<KernelVariant> v;
	VariantTable()
	{
		typedef const KernelVariant *(*group_fn)(int *);
		static const group_fn groups[VARIANT_GROUPS] = {polar_variant_group_0, polar_variant_group_1, polar_variant_group_2, polar_variant_group_3,
			polar_variant_group_4, polar_variant_group_5, polar_variant_group_6, polar_variant_group_7, polar_variant_group_8};
		for (group_fn g : groups) {
			int n = 0;
			const KernelVariant *t = g(&n);
			v.insert(v.end(), t, t + n);
		}
	}
};
static const std::vector<KernelVariant> &variants()
{
	static const VariantTable t;
	return t.v;
}

static const KernelVariant *find_variant(int q, int rl, int team = 1)
{
	for (const KernelVariant &v : variants())
		if (v.q == q && v.rl == rl && v.team == team)
			return &v;
	return nullptr;
}

// Scratch of polar_hip_decode_static(): the wavefront arenas of callers that may not allocate.
// One copy per GPU, resident from code-object load until the process ends. Host code never uses
// the symbol's name as an address: g_static_scratch[device] holds what hipGetSymbolAddress
// returned with that device current (polar_hip_create).
enum : size_t { STATIC_SCRATCH_BYTES = (size_t)POLAR_STATIC_SCRATCH_BYTES };
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
enum { MAX_DEVICES = 64, STATIC_Q = 8, STATIC_RL = 8, STATIC_TEAM = 4 }; // the kernel shape of the static entry point: 32 frames per wavefront, in teams of four when the batch is whole teams
static void *g_static_scratch[MAX_DEVICES];

// ---------------------------------------------------------------------------------------
// float decoder: PolarDecoder<SIMD<float,W>,M> with PolarHelper<SIMD<VALUE,WIDTH>>
// (polar_helper.hh:62-111), the path the reference bench takes with `typedef float code_type`
// (testbench.cc:49-53) -- for BER comparisons against the int8 decoder, not for speed.
// One lane = one frame, exactly the reference's per-lane arithmetic in its AVX2 forms (vsignum:
// +0 for +-0 else copysign(1, a), avx2.hh:514-521; vcopysign :602-610; vmin(a, b) = a < b ? a : b),
// signed zeros included. soft[2N] / hard[N] (polar_decoder.hh:128-129) live in a per-wave arena,
// element-major / lane-minor so that every access of the wave is one 256-byte row.
// ---------------------------------------------------------------------------------------
struct FloatArgs {
	const uint8_t *prog; // the reference byte-code itself
	const float *cw;     // [N][frames]
	float *msg;          // [rows][frames]
	float *arena;        // per workgroup: 3N x 64 floats
	size_t frames;
	u32 groups;
	int top;
};

__device__ __forceinline__ float f_signum(float a) { return a == 0.0f ? 0.0f : copysignf(1.0f, a); }
__device__ __forceinline__ float f_min(float a, float b) { return a < b ? a : b; }
__device__ __forceinline__ float f_prod(float a, float b) { return (f_signum(a) * f_signum(b)) * f_min(fabsf(a), fabsf(b)); }

// Lane columns: element i of a vector sits at p + i * stride (stride 64 in the arena and in LDS, `frames`
// in the caller's message rows). The loops below move FU elements at a time, all loads of a block
// before its first store (source and destination of one node function never overlap), so that a lane
// has FU or 2 FU rows in flight instead of one: the kernel is bound by memory latency, not by arithmetic.
enum { FU = 8, F32_LDS_LEVEL = 5, F32_LDS_FLOATS = 64 << (F32_LDS_LEVEL + 1) }; // soft[1 .. 2^(L+1)) of the wave in LDS: 16 KiB

template <class F>
__device__ __forceinline__ void f_map2(float *d, const float *x, const float *y, size_t cnt, F fn)
{
	if (cnt >= FU) {
		for (size_t i = 0; i < cnt; i += FU) {
			float a[FU], b[FU];
#pragma unroll
			for (int u = 0; u < FU; ++u) {
				a[u] = x[(i + u) * 64];
				b[u] = y[(i + u) * 64];
			}
#pragma unroll
			for (int u = 0; u < FU; ++u)
				d[(i + u) * 64] = fn(a[u], b[u]);
		}
	} else {
		for (size_t i = 0; i < cnt; ++i)
			d[i * 64] = fn(x[i * 64], y[i * 64]);
	}
}

// trans (polar_decoder.hh:15-27) of `length` symbols: out / inp are lane columns with the given strides
__device__ __forceinline__ void f_trans(float *out, size_t os, const float *inp, size_t is, int length, bool active)
{
	if (!active)
		return;
	if (length >= 2 * FU) {
		for (int i = 0; i < length; i += 2 * FU) {
			float v[2 * FU];
#pragma unroll
			for (int u = 0; u < 2 * FU; ++u)
				v[u] = inp[(size_t)(i + u) * is];
#pragma unroll
			for (int u = 0; u < 2 * FU; u += 2) {
				out[(size_t)(i + u) * os] = v[u] * v[u + 1];
				out[(size_t)(i + u + 1) * os] = v[u + 1];
			}
		}
	} else {
		for (int i = 0; i < length; i += 2) {
			float b = inp[(size_t)(i + 1) * is];
			out[(size_t)i * os] = inp[(size_t)i * is] * b;
			out[(size_t)(i + 1) * os] = b;
		}
	}
	for (int h = 2; h < length; h *= 2) {
		if (h >= FU) {
			for (int i = 0; i < length; i += 2 * h)
				for (int j = i; j < i + h; j += FU) {
					float a[FU], b[FU];
#pragma unroll
					for (int u = 0; u < FU; ++u) {
						a[u] = out[(size_t)(j + u) * os];
						b[u] = out[(size_t)(j + u + h) * os];
					}
#pragma unroll
					for (int u = 0; u < FU; ++u)
						out[(size_t)(j + u) * os] = a[u] * b[u];
				}
		} else {
			for (int i = 0; i < length; i += 2 * h)
				for (int j = i; j < i + h; ++j)
					out[(size_t)j * os] = out[(size_t)j * os] * out[(size_t)(j + h) * os];
		}
	}
}

__global__ void __launch_bounds__(64) sc_decode_f32_kernel(FloatArgs a)
{
	__shared__ float low[F32_LDS_FLOATS];
	const size_t n = (size_t)1 << a.top;
	float *soft_g = a.arena + (size_t)blockIdx.x * 3 * n * 64 + threadIdx.x; // soft[i] at soft_g + 64 * i ...
	float *soft_l = low + threadIdx.x;                                        // ... except the low levels, in LDS
	// level l occupies soft[2^l .. 2^(l+1)): wholly on one side; blocks of FU indices never straddle
	auto S = [&](size_t idx) -> float * { return idx < ((size_t)2 << F32_LDS_LEVEL) ? soft_l + idx * 64 : soft_g + idx * 64; };
	float *hard0 = soft_g + 2 * n * 64;
	for (u32 grp = blockIdx.x; grp < a.groups; grp += gridDim.x) {
		const size_t f = (size_t)grp * 64 + threadIdx.x;
		const bool active = f < a.frames;
		const float *cw = a.cw + (active ? f : 0);
		float *msg = a.msg + (active ? f : 0);
		if (n >= 2 * FU) { // polar_decoder.hh:136-137
			for (size_t i = 0; i < n; i += 2 * FU) {
				float v[2 * FU];
#pragma unroll
				for (int u = 0; u < 2 * FU; ++u)
					v[u] = cw[(i + u) * a.frames];
#pragma unroll
				for (int u = 0; u < 2 * FU; ++u)
					S(n + i)[u * 64] = v[u];
			}
		} else {
			for (size_t i = 0; i < n; ++i)
				S(n + i)[0] = cw[i * a.frames];
		}
		float *hrd = hard0;
		int lvl = a.top;
		size_t row = 0;
		for (const uint8_t *pc = a.prog + 1; *pc != 255; ++pc) {
			const u32 op = *pc;
			if (op == OP_LEFT || op == OP_RATE0_RIGHT) {
				size_t len = (size_t)1 << lvl, h = len / 2;
				if (op == OP_LEFT)
					f_map2(S(h), S(len), S(len + h), h, [](float x, float y) { return f_prod(x, y); });
				else
					f_map2(S(h), S(len), S(len + h), h, [](float x, float y) { return x + y; });
				--lvl;
				if (op == OP_RATE0_RIGHT)
					hrd += ((size_t)64 << lvl);
			} else if (op == OP_RIGHT) {
				size_t len = (size_t)2 << lvl, h = len / 2;
				float *d = S(h);
				const float *x = S(len), *y = S(len + h);
				if (h >= FU) {
					for (size_t i = 0; i < h; i += FU) {
						float u0[FU], u1[FU], u2[FU];
#pragma unroll
						for (int u = 0; u < FU; ++u) {
							u0[u] = hrd[(i + u) * 64];
							u1[u] = x[(i + u) * 64];
							u2[u] = y[(i + u) * 64];
						}
#pragma unroll
						for (int u = 0; u < FU; ++u)
							d[(i + u) * 64] = u0[u] * u1[u] + u2[u];
					}
				} else {
					for (size_t i = 0; i < h; ++i)
						d[i * 64] = hrd[i * 64] * x[i * 64] + y[i * 64];
				}
				hrd += ((size_t)64 << lvl);
			} else if (op == OP_COMB || op == OP_RATE0_COMB) {
				hrd -= ((size_t)64 << lvl);
				++lvl;
				size_t h = (size_t)1 << (lvl - 1);
				if (op == OP_COMB)
					f_map2(hrd, hrd, hrd + h * 64, h, [](float l, float r) { return l * r; }); // in place: element i reads only i and i + h
				else
					f_map2(hrd, hrd, hrd + h * 64, h, [](float, float r) { return r; });
			} else if (op == OP_RATE0) {
				for (size_t i = 0; i < ((size_t)1 << lvl); ++i)
					hrd[i * 64] = 1.0f;
			} else if (op == OP_RATE1) {
				size_t len = (size_t)1 << lvl;
				f_map2(hrd, S(len), S(len), len, [](float x, float) { return f_signum(x); });
				f_trans(msg + row * a.frames, a.frames, hrd, 64, (int)len, active);
				row += len;
			} else if (op == OP_REP) {
				size_t len = (size_t)1 << lvl;
				for (size_t h = len; h > 1; h /= 2)
					f_map2(S(h / 2), S(h), S(h + h / 2), h / 2, [](float x, float y) { return x + y; });
				float d = f_signum(S(1)[0]);
				if (active)
					msg[row * a.frames] = d;
				for (size_t i = 0; i < len; ++i)
					hrd[i * 64] = d;
				++row;
			} else if (op == OP_SPC) {
				// magnitudes go to the dead arena rows [0, len) (always the arena, also for short nodes)
				size_t len = (size_t)1 << lvl;
				float parity = 1.0f, weak = 0.0f;
				const float *x = S(len);
				for (size_t i = 0; i < len; i += FU) {
					float v[FU];
#pragma unroll
					for (int u = 0; u < FU; ++u)
						v[u] = i + u < len ? x[(i + u) * 64] : 0.0f;
#pragma unroll
					for (int u = 0; u < FU; ++u)
						if (i + u < len) {
							float d = copysignf(1.0f, v[u]), m = fabsf(v[u]);
							hrd[(i + u) * 64] = d;
							soft_g[(i + u) * 64] = m;
							parity = i + u ? parity * d : d;
							weak = i + u ? f_min(weak, m) : m;
						}
				}
				for (size_t i = 0; i < len; i += FU) {
					float m[FU], d[FU];
#pragma unroll
					for (int u = 0; u < FU; ++u)
						if (i + u < len) {
							m[u] = soft_g[(i + u) * 64];
							d[u] = hrd[(i + u) * 64];
						}
#pragma unroll
					for (int u = 0; u < FU; ++u)
						if (i + u < len && weak == m[u])
							hrd[(i + u) * 64] = d[u] * parity;
				}
				f_trans(soft_g, 64, hrd, 64, (int)len, true);
				if (active)
					for (size_t i = 0; i + 1 < len; ++i)
						msg[(row + i) * a.frames] = soft_g[(i + 1) * 64];
				row += len - 1;
			} else if (op == OP_RATE1_COMB) {
				++lvl;
				size_t h = (size_t)1 << (lvl - 1);
				const float *x = S(2 * h), *y = S(3 * h);
				for (size_t i = 0; i < h; i += FU) {
					float l[FU], u1[FU], u2[FU];
#pragma unroll
					for (int u = 0; u < FU; ++u)
						if (i + u < h) {
							l[u] = hrd[(i + u) * 64];
							u1[u] = x[(i + u) * 64];
							u2[u] = y[(i + u) * 64];
						}
#pragma unroll
					for (int u = 0; u < FU; ++u)
						if (i + u < h) {
							float r = f_signum(l[u] * u1[u] + u2[u]);
							hrd[(i + u + h) * 64] = r;
							hrd[(i + u) * 64] = l[u] * r;
						}
				}
				f_trans(msg + row * a.frames, a.frames, hrd + h * 64, 64, (int)h, active);
				row += h;
			}
		}
	}
}

// ---------------------------------------------------------------------------------------
// encoder kernels
// ---------------------------------------------------------------------------------------

// Symbols in {-1,0,+1} as two masks per byte: neg = bit 7, nz = bit 0. The reference multiplies
// symbols (qmul, polar_helper.hh:145-152): the product negates by xor and is nonzero by and.
template <typename T> struct Word;
template <> struct Word<u32> { static const u32 NEG = 0x80808080u, NZ = 0x01010101u; };
template <> struct Word<uint8_t> { static const u32 NEG = 0x80u, NZ = 0x01u; };

enum { LOAD_INPLACE = 0, LOAD_MESSAGE = 1, LOAD_REFREEZE = 2 };

// Stages h = 2^s0 .. 2^(s0+C-1) of the transform (polar_encoder.hh:17-26), 2^C rows per thread
// in registers. cols = row length in units of T. src_row[i] = message row feeding code
// position i, or -1 where the position is frozen (+1), used by LOAD_MESSAGE / LOAD_REFREEZE.
template <typename T, int C>
__global__ void __launch_bounds__(256) butterfly_kernel(T *cw, const T *msg, const int *src_row, size_t cols, int level, int s0, int mode)
{
	const size_t colblocks = (cols + 255) / 256;
	const size_t col = (blockIdx.x % colblocks) * 256 + threadIdx.x;
	if (col >= cols)
		return;
	const int c = level - s0 < C ? level - s0 : C; // stages in this pass
	const size_t j = blockIdx.x / colblocks;       // enumerates rows with bits [s0, s0+c) clear
	const size_t low = j & (((size_t)1 << s0) - 1);
	const size_t row0 = ((j >> s0) << (s0 + c)) | low;
	u32 neg[1 << C], nz[1 << C];
#pragma unroll
	for (int k = 0; k < (1 << C); ++k) {
		if (k < (1 << c)) {
			size_t row = row0 + ((size_t)k << s0);
			u32 x;
			if (mode == LOAD_INPLACE)
				x = cw[row * cols + col];
			else {
				int src = src_row[row];
				x = src < 0 ? Word<T>::NZ : mode == LOAD_MESSAGE ? (u32)msg[(size_t)src * cols + col] : (u32)cw[row * cols + col];
			}
			neg[k] = x & Word<T>::NEG;
			nz[k] = x & Word<T>::NZ;
		}
	}
#pragma unroll
	for (int t = 0; t < C; ++t) {
		if (t < c) {
#pragma unroll
			for (int k = 0; k < (1 << C); ++k)
				if (!(k & (1 << t)) && k + (1 << t) < (1 << c)) {
					neg[k] ^= neg[k + (1 << t)];
					nz[k] &= nz[k + (1 << t)];
				}
		}
	}
#pragma unroll
	for (int k = 0; k < (1 << C); ++k) {
		if (k < (1 << c)) {
			size_t row = row0 + ((size_t)k << s0);
			u32 m = nz[k] * 0xFFu; // 0xFF where nonzero
			u32 v = m & (((neg[k] >> 7) * 0xFFu) | Word<T>::NZ);
			cw[row * cols + col] = (T)v;
		}
	}
}

// ---------------------------------------------------------------------------------------
// systematic post-processing (testbench.cc:177-183) on packed symbols
// ---------------------------------------------------------------------------------------
// Re-encoding the decoded u (PolarEncoder, polar_encoder.hh:15-27) and gathering the information
// rows needs only the sign and the "is zero" bit of every symbol, so between the passes of the
// transform a row of 32 frames is two dwords (neg, nz: bit f = frame 32*col + f) instead of 32
// bytes. The first pass places the message rows (frozen positions are +1), the last pass writes
// only the information rows -- as int8 symbols or as the two bit planes of the wire format.
// Thread = one column of 32 frames x 2^C rows {row0 + (k << s0)} in registers, C stages per pass.
struct PackedRow { u32 neg, nz; };

// bit 7 of the four bytes of x -> bits 0..3
// (shifts and ors: a 32-bit multiply runs at a quarter of their rate)
__device__ __forceinline__ u32 gather_msb(u32 x)
{
	u32 t = (x >> 7) & 0x01010101u;
	return (t | (t >> 7) | (t >> 14) | (t >> 21)) & 15u;
}
// bits 0..3 -> bit 0 of the four bytes
__device__ __forceinline__ u32 spread_lsb(u32 nib) { return (nib | (nib << 7) | (nib << 14) | (nib << 21)) & 0x01010101u; }
// 0xFF in every byte whose bit 0 is set (bytes are 0 or 1)
__device__ __forceinline__ u32 byte_fill(u32 lsb) { return (lsb << 8) - lsb; }

// ragged batches / unaligned buffers: byte by byte, kept out of line so that the unrolled passes stay small
__device__ __noinline__ PackedRow pack_symbols_slow(const uint8_t *row, size_t f0, size_t frames)
{
	PackedRow r = {0u, 0u};
	for (int f = 0; f < 32; ++f)
		if (f0 + f < frames) {
			int v = (int8_t)row[f0 + f];
			r.neg |= (u32)(v < 0) << f;
			r.nz |= (u32)(v != 0) << f;
		}
	return r;
}
__device__ __noinline__ void unpack_symbols_slow(uint8_t *row, size_t f0, size_t frames, PackedRow r)
{
	for (int f = 0; f < 32; ++f)
		if (f0 + f < frames)
			row[f0 + f] = (r.nz >> f) & 1u ? ((r.neg >> f) & 1u ? 0xFFu : 0x01u) : 0x00u;
}

__device__ __forceinline__ PackedRow pack_symbols(const uint8_t *row, size_t f0, size_t frames, bool wide)
{
	PackedRow r = {0u, 0u};
	if (wide) { // 32 frames, 16-byte aligned
		const u32x4 *p = reinterpret_cast<const u32x4 *>(row + f0);
		u32x4 a = p[0], b = p[1];
		u32 w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			r.neg |= gather_msb(w[k]) << (4 * k);
			r.nz |= (gather_msb(zero_bytes(w[k])) ^ 15u) << (4 * k);
		}
	} else {
		r = pack_symbols_slow(row, f0, frames);
	}
	return r;
}

__device__ __forceinline__ void unpack_symbols(uint8_t *row, size_t f0, size_t frames, bool wide, PackedRow r)
{
	if (wide) {
		u32 w[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			u32 ng = byte_fill(spread_lsb((r.neg >> (4 * k)) & 15u)), nz = byte_fill(spread_lsb((r.nz >> (4 * k)) & 15u));
			w[k] = nz & (ng | 0x01010101u);
		}
		u32x4 a, b;
		a.x = w[0]; a.y = w[1]; a.z = w[2]; a.w = w[3];
		b.x = w[4]; b.y = w[5]; b.z = w[6]; b.w = w[7];
		u32x4 *p = reinterpret_cast<u32x4 *>(row + f0);
		p[0] = a;
		p[1] = b;
	} else {
		unpack_symbols_slow(row, f0, frames, r);
	}
}

struct PackedPass {
	PackedRow *tmp;        // [N][cols]
	const uint8_t *msg;    // first pass: [K][frames] symbols
	uint8_t *out;          // last pass: [K][frames] symbols, or null
	u32 *out_sign, *out_eras; // last pass: [K][cols] bit planes (sign, erasure), or null
	const int *src_row;    // [N]: message row of a code position, -1 = frozen
	size_t frames, cols;
	int level, s0, first, last, wide; // first: PACK_LOAD_*, last: PACK_STORE_*
	int S, CW;                        // packed_tile_kernel: stages of the pass, columns per tile
};
enum { PACK_LOAD_TMP = 0, PACK_LOAD_MESSAGE = 1, PACK_LOAD_REFREEZE = 2 }; // refreeze: tmp, with +1 at the frozen positions (polar_encoder.hh:46-48)
enum { PACK_STORE_TMP = 0, PACK_STORE_INFO = 1, PACK_STORE_ALL = 2 };      // info rows gathered to [K][...]; all rows as symbols to out[N][frames]

template <int C>
__global__ void __launch_bounds__(256) packed_transform_kernel(PackedPass a)
{
	const int c = C; // stages in this pass (the host picks the instantiation)
	const size_t item = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t col = item % a.cols, j = item / a.cols;   // j enumerates rows with bits [s0, s0+c) clear
	if (j >= ((size_t)1 << a.level) >> c)
		return;
	const size_t low = j & (((size_t)1 << a.s0) - 1);
	const size_t row0 = ((j >> a.s0) << (a.s0 + c)) | low;
	const size_t f0 = 32 * col;
	const u32 valid = a.frames - f0 >= 32 ? ~0u : (1u << (a.frames - f0)) - 1u;
	u32 neg[1 << C], nz[1 << C];
#pragma unroll
	for (int k = 0; k < (1 << C); ++k) {
		{
			const size_t row = row0 + ((size_t)k << a.s0);
			PackedRow r;
			if (a.first != PACK_LOAD_TMP) {
				int src = a.src_row[row];
				if (src < 0) {
					r.neg = 0u;
					r.nz = valid;
				} else if (a.first == PACK_LOAD_MESSAGE) {
					r = pack_symbols(a.msg + (size_t)src * a.frames, f0, a.frames, a.wide);
				} else {
					r = a.tmp[row * a.cols + col];
				}
			} else {
				r = a.tmp[row * a.cols + col];
			}
			neg[k] = r.neg;
			nz[k] = r.nz;
		}
	}
#pragma unroll
	for (int t = 0; t < C; ++t) {
		{
#pragma unroll
			for (int k = 0; k < (1 << C); ++k)
				if (!(k & (1 << t))) { // qmul(a, b): sign xor, zero if either is zero
					neg[k] ^= neg[k + (1 << t)];
					nz[k] &= nz[k + (1 << t)];
				}
		}
	}
#pragma unroll
	for (int k = 0; k < (1 << C); ++k) {
		{
			const size_t row = row0 + ((size_t)k << a.s0);
			PackedRow r = {neg[k] & nz[k], nz[k]};
			if (a.last == PACK_STORE_TMP) {
				a.tmp[row * a.cols + col] = r;
				continue;
			}
			if (a.last == PACK_STORE_ALL) {
				unpack_symbols(a.out + row * a.frames, f0, a.frames, a.wide, r);
				continue;
			}
			int dst = a.src_row[row];
			if (dst < 0)
				continue;
			if (a.out)
				unpack_symbols(a.out + (size_t)dst * a.frames, f0, a.frames, a.wide, r);
			if (a.out_sign)
				a.out_sign[(size_t)dst * a.cols + col] = r.neg;
			if (a.out_eras)
				a.out_eras[(size_t)dst * a.cols + col] = ~r.nz & valid;
		}
	}
}

// One row of one 32-frame column for the first pass of a transform (a.first) ...
__device__ __forceinline__ PackedRow packed_load(const PackedPass &a, size_t row, size_t col, size_t f0, u32 valid)
{
	if (a.first != PACK_LOAD_TMP) {
		int src = a.src_row[row];
		if (src < 0) {
			PackedRow r = {0u, valid};
			return r;
		}
		if (a.first == PACK_LOAD_MESSAGE)
			return pack_symbols(a.msg + (size_t)src * a.frames, f0, a.frames, a.wide);
	}
	return a.tmp[row * a.cols + col];
}
// ... and for the last one (a.last)
__device__ __forceinline__ void packed_store(const PackedPass &a, size_t row, size_t col, size_t f0, u32 valid, PackedRow r)
{
	if (a.last == PACK_STORE_TMP) {
		a.tmp[row * a.cols + col] = r;
		return;
	}
	if (a.last == PACK_STORE_ALL) {
		unpack_symbols(a.out + row * a.frames, f0, a.frames, a.wide, r);
		return;
	}
	int dst = a.src_row[row];
	if (dst < 0)
		return;
	if (a.out)
		unpack_symbols(a.out + (size_t)dst * a.frames, f0, a.frames, a.wide, r);
	if (a.out_sign)
		a.out_sign[(size_t)dst * a.cols + col] = r.neg;
	if (a.out_eras)
		a.out_eras[(size_t)dst * a.cols + col] = ~r.nz & valid;
}

// S = 4..10 stages per pass: a workgroup owns a tile of 2^S rows {row_base + (r << s0)} x CW
// columns (2^S * CW <= 4096 packed rows, 32 KiB of LDS), a thread 16 rows of one column in
// registers. The stages are done four at a time on the rows that differ in those four bits of r;
// between such phases the tile goes through LDS and the threads regroup. The last phase takes the
// top four bits and skips the stages already done. Global accesses are CW * 8 bytes (CW * 32 for
// symbols) contiguous per row.
#ifndef POLAR_TILE_ITEMS
#define POLAR_TILE_ITEMS 4096
#endif
enum { TILE_ITEMS = POLAR_TILE_ITEMS, TILE_ROWS = 16 };
__global__ void __launch_bounds__(TILE_ITEMS / TILE_ROWS) packed_tile_kernel(PackedPass a)
{
	__shared__ u32 tile_neg[TILE_ITEMS], tile_nz[TILE_ITEMS];
	const int S = a.S, CW = a.CW;
	const int c = (int)threadIdx.x % CW, g = (int)threadIdx.x / CW; // g: the S - 4 bits of r a phase does not touch
	const size_t col_tiles = (a.cols + CW - 1) / CW;
	const size_t col = (blockIdx.x % col_tiles) * CW + c, j = blockIdx.x / col_tiles; // j enumerates rows with bits [s0, s0+S) clear
	const bool live = col < a.cols;
	const size_t row_base = ((j >> a.s0) << (a.s0 + S)) | (j & (((size_t)1 << a.s0) - 1));
	const size_t f0 = 32 * col;
	const u32 valid = !live ? 0u : a.frames - f0 >= 32 ? ~0u : (1u << (a.frames - f0)) - 1u;
	const int phases = (S + 3) / 4;
	u32 neg[TILE_ROWS], nz[TILE_ROWS];
	for (int p = 0; p < phases; ++p) {
		const int done = 4 * p;                            // stages below this bit are finished
		const int lo = p == phases - 1 ? S - 4 : done;     // this phase's rows differ in bits [lo, lo+4)
		const int r0 = ((g >> lo) << (lo + 4)) | (g & ((1 << lo) - 1));
#pragma unroll
		for (int k = 0; k < TILE_ROWS; ++k) {
			const int r = r0 | (k << lo);
			if (p == 0) {
				PackedRow x = {0u, 0u};
				if (live)
					x = packed_load(a, row_base + ((size_t)r << a.s0), col, f0, valid);
				neg[k] = x.neg;
				nz[k] = x.nz;
			} else {
				neg[k] = tile_neg[r * CW + c];
				nz[k] = tile_nz[r * CW + c];
			}
		}
#pragma unroll
		for (int t = 0; t < 4; ++t)
			if (lo + t >= done) {
#pragma unroll
				for (int k = 0; k < TILE_ROWS; ++k)
					if (!(k & (1 << t))) { // qmul(a, b): sign xor, zero if either is zero
						neg[k] ^= neg[k + (1 << t)];
						nz[k] &= nz[k + (1 << t)];
					}
			}
		if (p == phases - 1) {
#pragma unroll
			for (int k = 0; k < TILE_ROWS; ++k) {
				PackedRow x = {neg[k] & nz[k], nz[k]};
				if (live)
					packed_store(a, row_base + ((size_t)(r0 | (k << lo)) << a.s0), col, f0, valid, x);
			}
		} else {
			if (p)
				__syncthreads(); // every thread has read its rows of the previous phase
#pragma unroll
			for (int k = 0; k < TILE_ROWS; ++k) {
				const int r = r0 | (k << lo);
				tile_neg[r * CW + c] = neg[k];
				tile_nz[r * CW + c] = nz[k];
			}
			__syncthreads();
		}
	}
}

// symbols[rows][frames] <- bit planes [rows][cols] (sign, erasure; erasure may be null)
__global__ void __launch_bounds__(256) unpack_planes_kernel(uint8_t *symbols, const u32 *sign, const u32 *eras, size_t rows, size_t frames, size_t cols, int wide)
{
	const size_t item = (size_t)blockIdx.x * 256 + threadIdx.x;
	const size_t col = item % cols, row = item / cols;
	if (row >= rows)
		return;
	PackedRow r;
	r.neg = sign[row * cols + col];
	r.nz = eras ? ~eras[row * cols + col] : ~0u;
	r.neg &= r.nz;
	unpack_symbols(symbols + row * frames, 32 * col, frames, wide, r);
}


// Bench counters (testbench.cc:185-192) over flat symbol arrays, 16 symbols per load:
//   c[0] += #(noisy * orig < 0)   c[1] += #(noisy == 0)   over n_code symbols
//   c[2] += #(decoded * message <= 0)   c[3] += #(decoded == 0)   over n_mesg symbols
__device__ __forceinline__ void count_pair(u32 a, u32 b, unsigned &lt, unsigned &za_n, unsigned &le)
{
	u32 za = zero_bytes(a), zb = zero_bytes(b), d = (a ^ b) & SIGN_BITS;
	lt += __builtin_popcount(d & ~za & ~zb);
	za_n += __builtin_popcount(za);
	le += __builtin_popcount(d | za | zb);
}

__global__ void __launch_bounds__(256) count_kernel(unsigned long long *c, const uint8_t *noisy, const uint8_t *orig, size_t n_code, const uint8_t *decoded, const uint8_t *message, size_t n_mesg)
{
	unsigned v[4] = {0, 0, 0, 0}, unused = 0;
	const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
	for (int part = 0; part < 2; ++part) {
		const uint8_t *a = part ? decoded : noisy, *b = part ? message : orig;
		size_t n = part ? n_mesg : n_code;
		if (!n)
			continue;
		bool wide = ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0);
		size_t nvec = wide ? n / 16 : 0;
		for (size_t i = tid; i < nvec; i += nt) {
			u32x4 x = reinterpret_cast<const u32x4 *>(a)[i], y = reinterpret_cast<const u32x4 *>(b)[i];
			unsigned &lt = part ? unused : v[0], &z = part ? v[3] : v[1], &le = part ? v[2] : unused;
			count_pair(x.x, y.x, lt, z, le);
			count_pair(x.y, y.y, lt, z, le);
			count_pair(x.z, y.z, lt, z, le);
			count_pair(x.w, y.w, lt, z, le);
		}
		for (size_t i = nvec * 16 + tid; i < n; i += nt) {
			int x = (int8_t)a[i], y = (int8_t)b[i];
			if (part) {
				v[2] += x * y <= 0;
				v[3] += x == 0;
			} else {
				v[0] += x * y < 0;
				v[1] += x == 0;
			}
		}
	}
	for (int k = 0; k < 4; ++k) {
		unsigned long long w = v[k];
		for (int off = 32; off; off >>= 1)
			w += __shfl_down(w, off, 64);
		if ((threadIdx.x & 63) == 0 && w)
			atomicAdd(&c[k], w);
	}
}

// ---------------------------------------------------------------------------------------
// message source and channel (channel.hpp): one thread per (row, global frame quad)
// ---------------------------------------------------------------------------------------

// MODE 0: llr = quant(fact * (cw + sigma * n)); MODE 1: msg = 1 - 2 * bit
// Two standard normals in single precision, for the screening pass of channel_kernel: same
// draws, same Box-Muller as normal_pair() (channel.hpp). `ok` is cleared where single precision
// is not good enough in absolute terms: u1 so close to 1 that r = sqrt(-2 log u1) is tiny.
__device__ __forceinline__ void normal_pair_f32(uint32_t a, uint32_t b, float &n0, float &n1, bool &ok)
{
	const float S = 1.0f / 4294967296.0f;
	ok &= a < 0xFFFE0000u; // u1 < 1 - 3e-5: r > 7e-3, its rounding error below 1e-5
	float u1 = ((float)a + 0.5f) * S, u2 = ((float)b + 0.5f) * S;
	float r = sqrtf(-2.0f * logf(u1)), sn, cs;
	sincospif(2.0f * u2, &sn, &cs);
	n0 = r * cs;
	n1 = r * sn;
}

// MODE 0: LLRs of the noisy codeword; MODE 1: the message source. One Philox block per frame quad.
// The LLR is quant(fact * (x + sigma * n)) in double (channel.hpp, testbench.cc:151-165), but the
// double-precision log / sin / cos are 90 % of the bench loop's channel time. So the quad is first
// computed in single precision (error below `thr` LLR units, see launch_channel); unless one of its
// four values lies within `thr` of a rounding boundary k + 1/2 that result IS the double-precision
// one after rounding, and is stored. Otherwise (a few per cent of the wavefronts) the quad is redone in
// double. thr <= 0: double only. `wide`: whole aligned quads move as one dword.
template <int MODE>
__global__ void __launch_bounds__(256) channel_kernel(int8_t *out, const int8_t *cw, size_t rows, size_t frames, double sigma, double fact, uint64_t seed, uint64_t frame0, float thr, int wide, unsigned long long *counts)
{
	unsigned n_flipped = 0, n_zero = 0; // counts[0] += #(llr * x < 0), counts[1] += #(llr == 0) (testbench.cc:185-188), when asked for
	const uint64_t q0 = frame0 >> 2, nq = ((frame0 + frames - 1) >> 2) - q0 + 1;
	const size_t items = rows * nq, nt = (size_t)gridDim.x * blockDim.x;
	const float sigmaf = (float)sigma, factf = (float)fact;
	for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += nt) {
		const size_t i = it / nq;
		const uint64_t quad = q0 + it % nq;
		PhiloxBlock b = draw(seed, MODE ? STREAM_MESSAGE : STREAM_NOISE, (uint32_t)i, quad);
		const size_t at0 = i * frames + (size_t)(4 * quad - frame0); // of the quad's first frame (wide: in range)
		int8_t res[4];
		if (MODE == 1) {
#pragma unroll
			for (int c = 0; c < 4; ++c)
				res[c] = (int8_t)(1 - 2 * (int)(b.v[c] & 1));
		} else {
			int8_t x[4] = {0, 0, 0, 0};
			if (wide) {
				uint32_t w = *reinterpret_cast<const uint32_t *>(cw + at0);
#pragma unroll
				for (int c = 0; c < 4; ++c)
					x[c] = (int8_t)(w >> (8 * c));
			} else {
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					uint64_t g = 4 * quad + c;
					if (g >= frame0 && g < frame0 + frames)
						x[c] = cw[i * frames + (size_t)(g - frame0)];
				}
			}
			bool ok = thr > 0.0f;
			if (ok) {
				float n[4];
				normal_pair_f32(b.v[0], b.v[1], n[0], n[1], ok);
				normal_pair_f32(b.v[2], b.v[3], n[2], n[3], ok);
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					float l = factf * ((float)x[c] + sigmaf * n[c]);
					float fl = floorf(l);
					ok &= fabsf(l - fl - 0.5f) > thr;
					float r = fl + (l - fl > 0.5f ? 1.0f : 0.0f);
					res[c] = (int8_t)(r < -128.0f ? -128.0f : r > 127.0f ? 127.0f : r);
				}
			}
			if (!ok) {
				double n[4];
				normal_pair(b.v[0], b.v[1], n[0], n[1]);
				normal_pair(b.v[2], b.v[3], n[2], n[3]);
#pragma unroll
				for (int c = 0; c < 4; ++c)
					res[c] = quant_llr(fact * ((double)x[c] + sigma * n[c]));
			}
		}
		if (wide) {
			uint32_t w = 0;
#pragma unroll
			for (int c = 0; c < 4; ++c)
				w |= (uint32_t)(uint8_t)res[c] << (8 * c);
			*reinterpret_cast<uint32_t *>(out + at0) = w;
		} else {
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				uint64_t g = 4 * quad + c;
				if (g >= frame0 && g < frame0 + frames)
					out[i * frames + (size_t)(g - frame0)] = res[c];
			}
		}
		if (MODE == 0 && counts) {
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				uint64_t g = 4 * quad + c;
				if (wide || (g >= frame0 && g < frame0 + frames)) {
					int xs = cw[i * frames + (size_t)(g - frame0)];
					n_flipped += (int)res[c] * xs < 0;
					n_zero += res[c] == 0;
				}
			}
		}
	}
	if (MODE == 0 && counts) {
		unsigned long long a = n_flipped, z = n_zero;
		for (int off = 32; off; off >>= 1) {
			a += __shfl_down(a, off, 64);
			z += __shfl_down(z, off, 64);
		}
		if ((threadIdx.x & 63) == 0) {
			if (a)
				atomicAdd(&counts[0], a);
			if (z)
				atomicAdd(&counts[1], z);
		}
	}
}

__global__ void add_counters_kernel(unsigned long long *c, unsigned long long a, unsigned long long b)
{
	c[4] += a;
	c[5] += b;
}

// ---------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------

struct DeviceBuffer {
	void *p = nullptr;
	size_t bytes = 0;
	int reserve(size_t want)
	{
		if (want <= bytes)
			return POLAR_OK;
		if (p)
			(void)hipFree(p);
		p = nullptr;
		bytes = 0;
		HIP_TRY(hipMalloc(&p, want));
		bytes = want;
		return POLAR_OK;
	}
	void release()
	{
		if (p)
			(void)hipFree(p);
		p = nullptr;
		bytes = 0;
	}
};

struct LoadedProgram {
	HostProgram host;                // unfused translation: level, rows, validation
	std::vector<uint8_t> bytes;      // the reference byte-code as loaded
	std::map<int, DevOp *> dev;      // device instruction stream per frame-quads-per-wave (fusion depends on it)
	uint8_t *dev_bytes = nullptr;    // the byte-code itself on the device (float decoder)
};

struct polar_hip_handle {
	int device = 0;
	int cus = 0;
	hipStream_t own_stream = nullptr; // created with the handle
	hipStream_t stream = nullptr;     // where the synchronous entry points enqueue: own_stream or the caller's (polar_hip_set_stream)
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	// host-buffer pipeline (polar_hip_decode with pinned POLAR_HOST buffers): copy-in and copy-out
	// streams beside `stream`, and per staging slot the events "copied in", "decoded", "copied out"
	hipStream_t pipe_in = nullptr, pipe_out = nullptr;
	hipEvent_t pipe_ev[2][3] = {};
	void *bounce_in = nullptr, *bounce_out = nullptr; // page-locked staging for pageable caller buffers, two slots each
	size_t bounce_in_bytes = 0, bounce_out_bytes = 0;
	std::map<const void *, int> occupancy; // resident workgroups per CU, per kernel variant
	bool timed = false;
	std::vector<LoadedProgram> programs;
	struct CachedCode {
		std::vector<uint8_t> frozen, program;
		int program_id, ambiguous;
	};
	struct CodeKey {
		int level, K;
		long double p;
		bool operator<(const CodeKey &o) const { return level != o.level ? level < o.level : K != o.K ? K < o.K : p < o.p; }
	};
	std::map<CodeKey, CachedCode> codes; // polar_hip_code: (level, K, p) -> frozen set, program, resident id
	DeviceBuffer scratch, hot, stage_in, stage_out, enc_tmp, tables, counters, pad_in, pad_out;
	DeviceBuffer sim_cw, sim_llr, sim_msg, sim_dec; // polar_hip_simulate's chunk
	DeviceBuffer f32_a, f32_b;                      // float path: LLRs / decisions of a chunk, symbol conversions
	std::vector<uint8_t> frozen_cache; // frozen set the device tables were built from
	int cfg_frames_per_wave = 0, cfg_lds_level = 0, cfg_team = 1, cfg_max_waves = 0;
	int last_shape[3] = {0, 0, 0}; // frames per wavefront, LDS level, team of the last decode launch
};

static int use_device(polar_hip_handle *h)
{
	HIP_TRY(hipSetDevice(h->device));
	return POLAR_OK;
}

// Device instruction stream of a loaded program for one wavefront shape (fusion depends on the
// frame quads per wavefront and the variant's chain depth): fused and uploaded on first use.
static int program_shape(const KernelVariant *v) { return v->q * 8 + v->chain_depth; }
static int upload_shape(LoadedProgram &lp, const KernelVariant *v)
{
	const int shape = program_shape(v);
	if (lp.dev.count(shape))
		return POLAR_OK;
	HostProgram hp;
	if (translate_program(lp.bytes.data(), (int)lp.bytes.size(), hp, v->q, v->chain_depth))
		return fail(POLAR_E_PROGRAM, "program translation failed");
	DevOp *d = nullptr;
	size_t bytes = hp.ops.size() * sizeof(DevOp);
	HIP_TRY(hipMalloc((void **)&d, bytes));
	hipError_t e = hipMemcpy(d, hp.ops.data(), bytes, hipMemcpyHostToDevice);
	if (e != hipSuccess) {
		(void)hipFree(d);
		HIP_TRY(e);
	}
	lp.dev[shape] = d;
	return POLAR_OK;
}

extern "C" const char *polar_hip_last_error(void)
{
	return g_error.c_str();
}

// ---- host-side construction ----

extern "C" int polar_code_const0_ex(uint8_t *frozen, int level, int K, long double erasure_probability, int *ambiguous)
{
	if (!frozen || level < 0 || level > MAX_LEVEL || K < 0 || K > (1 << level))
		return fail(POLAR_E_ARG, "polar_code_const0: bad argument");
	int amb = code_const0(frozen, level, K, erasure_probability);
	if (ambiguous)
		*ambiguous = amb;
	return POLAR_OK;
}

extern "C" int polar_code_const0(uint8_t *frozen, int level, int K, long double erasure_probability)
{
	return polar_code_const0_ex(frozen, level, K, erasure_probability, nullptr);
}

extern "C" int polar_code_const0_batch(uint8_t *frozen, int *ambiguous, int level, int K, const long double *erasure_probability, int count)
{
	if (!frozen || !erasure_probability || count < 0 || level < 0 || level > MAX_LEVEL || K < 0 || K > (1 << level))
		return fail(POLAR_E_ARG, "polar_code_const0_batch: bad argument");
	const size_t n = (size_t)1 << level;
	unsigned hw = std::thread::hardware_concurrency();
	int workers = (int)(hw ? (hw > 16 ? 16 : hw) : 4);
	if (workers > count)
		workers = count;
	auto work = [&](int w) {
		for (int i = w; i < count; i += workers) {
			int amb = code_const0(frozen + (size_t)i * n, level, K, erasure_probability[i]);
			if (ambiguous)
				ambiguous[i] = amb;
		}
	};
	std::vector<std::thread> pool;
	for (int w = 1; w < workers; ++w)
		pool.emplace_back(work, w);
	if (workers > 0)
		work(0);
	for (auto &t : pool)
		t.join();
	return POLAR_OK;
}

extern "C" int polar_freezer(uint8_t *frozen, int level, long double erasure_probability, long double freezing_threshold)
{
	if (!frozen || level < 0 || level > MAX_LEVEL)
		return fail(POLAR_E_ARG, "polar_freezer: bad argument");
	return freezer(frozen, level, erasure_probability, freezing_threshold);
}

extern "C" int polar_compile(uint8_t *program, const uint8_t *frozen, int level)
{
	if (!program || !frozen || level < 1 || level > MAX_LEVEL)
		return fail(POLAR_E_ARG, "polar_compile: bad argument");
	ProgramBuilder b;
	int n = b.run(program, frozen, level);
	if (n < 0)
		return fail(POLAR_E_PROGRAM, "polar_compile: an information bit directly left of a frozen bit cannot be compiled (reference asserts level > 0)");
	return n;
}

extern "C" int polar_program_rows(const uint8_t *program, int length)
{
	if (!program)
		return fail(POLAR_E_ARG, "polar_program_rows: null program");
	HostProgram hp;
	int err = translate_program(program, length, hp);
	if (err)
		return fail(POLAR_E_PROGRAM, "malformed program (rule " + std::to_string(err) + ")");
	return (int)hp.rows;
}

// ---- device objects ----

extern "C" int polar_hip_create(polar_hip_handle **out, int device)
{
	if (!out)
		return fail(POLAR_E_ARG, "polar_hip_create: null out pointer");
	*out = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
		return fail(POLAR_E_NODEVICE, "polar_hip_create: no HIP device");
	if (device < 0 || device >= count)
		return fail(POLAR_E_ARG, "polar_hip_create: device index out of range");
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if (!strstr(prop.gcnArchName, "gfx950"))
		return fail(POLAR_E_NODEVICE, std::string("polar_hip_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
	polar_hip_handle *h = new polar_hip_handle;
	h->device = device;
	h->cus = prop.multiProcessorCount;
	auto init = [&]() -> int {
		HIP_TRY(hipSetDevice(device));
		HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
		h->stream = h->own_stream;
		HIP_TRY(hipEventCreate(&h->ev0));
		HIP_TRY(hipEventCreate(&h->ev1));
		if (device < MAX_DEVICES && !g_static_scratch[device]) { // once per device, with that device current
			void *ptr = nullptr;
			HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
			g_static_scratch[device] = ptr;
		}
		for (int team : {1, (int)STATIC_TEAM}) { // the static entry point must not meet a first-use query: resident workgroups of its kernel shapes, now
			const KernelVariant *sv = find_variant(STATIC_Q, STATIC_RL, team);
			if (!sv)
				return fail(POLAR_E_HIP, "polar_hip_create: the kernel shape of polar_hip_decode_static is not built into this library");
			int per_cu = 0;
			HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sv->fn, 64 * team, 0));
			h->occupancy[(const void *)sv->fn] = per_cu;
		}
		return POLAR_OK;
	};
	int rc = init();
	if (rc) { // nothing of a half-made handle is left behind (the error text is kept)
		std::string keep = g_error;
		polar_hip_destroy(h);
		g_error = keep;
		return rc;
	}
	*out = h;
	return POLAR_OK;
}

extern "C" void polar_hip_destroy(polar_hip_handle *h)
{
	if (!h)
		return;
	(void)hipSetDevice(h->device);
	(void)hipStreamSynchronize(h->stream);
	if (h->stream != h->own_stream)
		(void)hipStreamSynchronize(h->own_stream);
	for (auto &p : h->programs) {
		for (auto &d : p.dev)
			(void)hipFree(d.second);
		if (p.dev_bytes)
			(void)hipFree(p.dev_bytes);
	}
	h->scratch.release();
	h->stage_in.release();
	h->stage_out.release();
	h->enc_tmp.release();
	h->tables.release();
	h->counters.release();
	h->pad_in.release();
	h->hot.release();
	h->pad_out.release();
	h->f32_a.release();
	h->f32_b.release();
	h->sim_cw.release();
	h->sim_llr.release();
	h->sim_msg.release();
	h->sim_dec.release();
	(void)hipEventDestroy(h->ev0);
	(void)hipEventDestroy(h->ev1);
	if (h->bounce_in)
		(void)hipHostFree(h->bounce_in);
	if (h->bounce_out)
		(void)hipHostFree(h->bounce_out);
	if (h->pipe_in) {
		(void)hipStreamDestroy(h->pipe_in);
		(void)hipStreamDestroy(h->pipe_out);
		for (auto &slot : h->pipe_ev)
			for (hipEvent_t e : slot)
				(void)hipEventDestroy(e);
	}
	(void)hipStreamDestroy(h->own_stream);
	delete h;
}

extern "C" int polar_hip_configure_team(polar_hip_handle *h, int frames_per_wave, int lds_level, int team, int max_waves)
{
	if (!h)
		return fail(POLAR_E_ARG, "polar_hip_configure: null handle");
	if (frames_per_wave || lds_level) {
		if (frames_per_wave % 4 || !find_variant(frames_per_wave / 4, lds_level, team))
			return fail(POLAR_E_ARG, "polar_hip_configure: no kernel built for this (frames_per_wave, lds_level, team)");
	}
	if (max_waves < 0)
		return fail(POLAR_E_ARG, "polar_hip_configure: negative max_waves");
	h->cfg_frames_per_wave = frames_per_wave;
	h->cfg_lds_level = lds_level;
	h->cfg_team = team;
	h->cfg_max_waves = max_waves;
	return POLAR_OK;
}

extern "C" int polar_hip_configure(polar_hip_handle *h, int frames_per_wave, int lds_level, int max_waves)
{
	return polar_hip_configure_team(h, frames_per_wave, lds_level, 1, max_waves);
}

extern "C" int polar_hip_set_stream(polar_hip_handle *h, void *stream, int use_callers)
{
	if (!h)
		return fail(POLAR_E_ARG, "polar_hip_set_stream: null handle");
	int rc = use_device(h);
	if (rc)
		return rc;
	hipStream_t next = use_callers ? (hipStream_t)stream : h->own_stream;
	if (next != h->stream) // work already enqueued through the handle shares its scratch: let it drain
		HIP_TRY(hipStreamSynchronize(h->stream));
	h->stream = next;
	return POLAR_OK;
}

extern "C" int polar_hip_load_program(polar_hip_handle *h, const uint8_t *program, int length)
{
	if (!h || !program)
		return fail(POLAR_E_ARG, "polar_hip_load_program: null argument");
	for (size_t i = 0; i < h->programs.size(); ++i) // the same byte-code again: the resident copy
		if ((int)h->programs[i].bytes.size() == length && !memcmp(h->programs[i].bytes.data(), program, (size_t)length))
			return (int)i;
	LoadedProgram lp;
	int err = translate_program(program, length, lp.host);
	if (err)
		return fail(POLAR_E_PROGRAM, "polar_hip_load_program: malformed program (rule " + std::to_string(err) + ")");
	lp.bytes.assign(program, program + length);
	// the shape polar_hip_decode_static() runs is made resident now: that entry point may not allocate
	int rc = use_device(h);
	if (rc || (rc = upload_shape(lp, find_variant(STATIC_Q, STATIC_RL))))
		return rc;
	h->programs.push_back(lp);
	return (int)h->programs.size() - 1;
}

// Bytes of scratch one wavefront needs: LLR levels above the LDS levels plus two partial-sum planes.
static size_t scratch_bytes_per_wave(int q, int top)
{
	size_t soft = (size_t)q << top;
	return (soft + 2 * (soft / 8)) * sizeof(u32);
}

// Default shape: the 20 KiB-of-LDS family (8 wavefronts per CU), widest first. 128 frames per
// wavefront measured best at N = 2^16; narrower groups when the batch is too small
// to give every resident wavefront a group, or when the scratch arenas would not fit.
// Scratch budget: what the handle holds already plus 5/8 of the free device memory (so that
// N = 2^20 with 2^16 frames gets its 80 GiB of (32, 8) arenas beside 128 GiB of caller data). The query
// is slow (milliseconds), so it is made only when a request exceeds what is held.
struct ScratchBudget {
	const polar_hip_handle *h;
	size_t cap = 0;
	bool queried = false;
	bool fits(size_t bytes)
	{
		if (bytes <= h->scratch.bytes)
			return true;
		return bytes <= limit();
	}
	size_t limit()
	{
		if (!queried) {
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
				free_b = 0;
			cap = h->scratch.bytes + free_b / 8 * 5;
			queried = true;
		}
		return cap;
	}
};

extern "C" int polar_hip_load_program(polar_hip_handle *h, const uint8_t *program, int length);

extern "C" int polar_hip_code(polar_hip_handle *h, int level, int K, long double erasure_probability, const uint8_t **frozen, const uint8_t **program, int *program_length, int *ambiguous)
{
	if (!h || level < 1 || level > MAX_LEVEL || K < 0 || K > (1 << level) || !(erasure_probability >= 0) || !(erasure_probability <= 1))
		return fail(POLAR_E_ARG, "polar_hip_code: bad argument");
	polar_hip_handle::CodeKey key = {level, K, erasure_probability};
	auto it = h->codes.find(key);
	if (it == h->codes.end()) {
		polar_hip_handle::CachedCode c;
		const size_t n = (size_t)1 << level;
		c.frozen.resize(n);
		c.ambiguous = code_const0(c.frozen.data(), level, K, erasure_probability);
		c.program.resize(2 * n + 1);
		ProgramBuilder b;
		int len = b.run(c.program.data(), c.frozen.data(), level);
		if (len < 0)
			return fail(POLAR_E_PROGRAM, "polar_hip_code: the frozen set of this (level, K, p) cannot be compiled (reference asserts level > 0)");
		c.program.resize((size_t)len);
		c.program_id = polar_hip_load_program(h, c.program.data(), len);
		if (c.program_id < 0)
			return c.program_id;
		it = h->codes.emplace(key, std::move(c)).first;
	}
	if (frozen)
		*frozen = it->second.frozen.data();
	if (program)
		*program = it->second.program.data();
	if (program_length)
		*program_length = (int)it->second.program.size();
	if (ambiguous)
		*ambiguous = it->second.ambiguous;
	return it->second.program_id;
}

// Highest scratch level that keeps the default cache policy (sc_core.hpp, GVec::hot): the levels just
// above the LDS levels, as many as the Infinity Cache holds for all resident wavefronts together
// (wavefronts * frames per wavefront * 2^level bytes per level).
// Their vectors are read back and rewritten in place before the streamed (non-temporal) levels above
// can displace them, so they never travel to HBM (tools/ubench/hbm_ceiling.hip).
static int hot_top_for(const KernelVariant *v, int top, size_t waves)
{
	if (v->team > 1) // the team kernels are built without hot levels (sc_core.hpp, Group::HOT)
		return v->rl;
	static const char *e = getenv("POLAR_HIP_HOT_LEVELS"); // experiments: number of hot levels
	if (e)
		return v->rl + atoi(e);
	static const char *b = getenv("POLAR_HIP_HOT_BYTES"); // experiments: cache budget in MiB
	const size_t budget = (b ? (size_t)atoi(b) : (size_t)480) << 20;
	size_t used = 0;
	int lvl = v->rl;
	while (lvl + 1 < top) {
		size_t bytes = waves * 4 * (size_t)v->q << (lvl + 1);
		if (used + bytes > budget)
			break;
		used += bytes;
		++lvl;
	}
	return lvl;
}

static const KernelVariant *pick_variant(const polar_hip_handle *h, int top, size_t frames, ScratchBudget &budget)
{
	if (h->cfg_frames_per_wave)
		return find_variant(h->cfg_frames_per_wave / 4, h->cfg_lds_level, h->cfg_team);
	// Measured (tools/gpu_sweep_pick.sh, tools/gpu_team*.sh): a wavefront walks
	// the program in about the same time whatever its width, so the widest group wins as soon as
	// every resident wavefront has one -- two for 128 frames, whose one-sweep batches the 64-frame
	// teams decode faster; groups of 32 frames already pay off when three eighths of the wavefronts
	// have one, narrower ones when half of them do. Groups of 32 and 64 frames go in teams of four (whole 128-byte input lines).
	const size_t resident = (size_t)8 * h->cus;
	static const int family[][3] = {{32, 6, 1}, {16, 7, 4}, {8, 8, 4}, {4, 9, 1}, {2, 10, 1}, {1, 11, 1}};
	for (const auto &f : family) {
		size_t groups = (frames + 4 * f[0] - 1) / (4 * f[0]);
		if (f[0] > 1 && groups < (f[0] == 32 ? 2 * resident : f[0] == 8 ? resident * 3 / 8 : f[0] < 8 ? resident / 2 : resident))
			continue;
		if (f[0] > 1 && top > f[1] && !budget.fits(resident * scratch_bytes_per_wave(f[0], top)))
			continue;
		return find_variant(f[0], f[1], f[2]);
	}
	return nullptr;
}

// Enqueues the decode of `frames` frames on `stream`. The kernel wants whole groups of 4Q frames
// and dword-aligned rows; anything else is staged through padded device copies (zero-filled
// tail frames decode to garbage that is never copied back).
static int launch_decode(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, hipStream_t stream, bool timed)
{
	LoadedProgram &lp = h->programs[program_id];
	ScratchBudget budget{h};
	const KernelVariant *v = pick_variant(h, lp.host.top, frames, budget);
	if (!v)
		return fail(POLAR_E_ARG, "no kernel variant for the configured (frames_per_wave, lds_level)");
	const int shape = program_shape(v);
	int rc;
	if ((rc = upload_shape(lp, v))) // first use with this wavefront shape: fuse and upload
		return rc;
	// a workgroup takes `team` neighbouring groups at a time (1, or TEAM_WAVES in the team form)
	const size_t team = (size_t)v->team, fpw = 4 * (size_t)v->q * team, n = (size_t)1 << lp.host.top, rows = lp.host.rows;
	const size_t units = (frames + fpw - 1) / fpw, groups = units * team;
	if (groups > 0x7FFFFFFFu)
		return fail(POLAR_E_ARG, "too many frames for one call");
	h->timed = false;
	if (!groups)
		return POLAR_OK;
	const size_t padded = units * fpw;
	const bool direct = padded == frames && (uintptr_t)codeword % 16 == 0 && (uintptr_t)message % 16 == 0;
	const uint8_t *cw = (const uint8_t *)codeword;
	uint8_t *msg = (uint8_t *)message;
	if (!direct) {
		if ((rc = h->pad_in.reserve(n * padded)) || (rc = h->pad_out.reserve((rows ? rows : 1) * padded)))
			return rc;
		HIP_TRY(hipMemsetAsync(h->pad_in.p, 0, n * padded, stream));
		HIP_TRY(hipMemcpy2DAsync(h->pad_in.p, padded, codeword, frames, frames, n, hipMemcpyDeviceToDevice, stream));
		cw = (const uint8_t *)h->pad_in.p;
		msg = (uint8_t *)h->pad_out.p;
	}
	int per_cu = 0;
	auto occ = h->occupancy.find((const void *)v->fn);
	if (occ == h->occupancy.end()) {
		HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v->fn, 64 * v->team, 0));
		h->occupancy[(const void *)v->fn] = per_cu;
	} else {
		per_cu = occ->second;
	}
	if (per_cu < 1)
		return fail(POLAR_E_HIP, "decode kernel does not fit on a CU");
	size_t blocks = (size_t)per_cu * h->cus;
	if (h->cfg_max_waves && (size_t)h->cfg_max_waves / team < blocks)
		blocks = (size_t)h->cfg_max_waves / team ? (size_t)h->cfg_max_waves / team : 1;
	size_t block_dwords = 0, max_blocks = ~(size_t)0 >> 8;
	if (lp.host.top > v->rl) { // LLR levels above the LDS levels and the partial-sum planes live in scratch
		size_t soft = (size_t)v->q << lp.host.top;
		block_dwords = soft + 2 * (soft / 8);
		if (const char *e = getenv("POLAR_HIP_ARENA_PAD")) // experiments: dwords between the arenas (a multiple of 4)
			block_dwords += (size_t)atoi(e) / 4 * 4;
		if (!budget.fits(blocks * team * block_dwords * sizeof(u32))) { // fewer resident wavefronts, more sweeps
			max_blocks = budget.limit() / (team * block_dwords * sizeof(u32));
			if (max_blocks < 1)
				return fail(POLAR_E_NOMEM, "not enough device memory for one wavefront's scratch arena");
			if (blocks > max_blocks)
				blocks = max_blocks;
		}
	}
	// every wavefront walks the same number of groups: no idle tail behind a last partial sweep
	size_t sweeps = (units + blocks - 1) / blocks;
	blocks = (units + sweeps - 1) / sweeps;
	if (blocks >= 8 && blocks % 8) // multiple of 8 for the XCD-aware group order
		blocks = blocks + 8 - blocks % 8 <= max_blocks ? blocks + 8 - blocks % 8 : blocks - blocks % 8;
	if (getenv("POLAR_HIP_DEBUG"))
		fprintf(stderr, "polar_hip: decode Q=%d RL=%d team=%d groups=%zu blocks=%zu (%d per CU x %d CUs) lds=%zu B direct=%d hot levels <= %d\n", v->q, v->rl, v->team, groups, blocks, per_cu, h->cus, v->lds_bytes, (int)direct, hot_top_for(v, lp.host.top, blocks * team));
	h->last_shape[0] = 4 * v->q;
	h->last_shape[1] = v->rl;
	h->last_shape[2] = v->team;
	DecodeArgs a;
	a.prog = lp.dev[shape];
	a.cw = cw;
	a.msg = msg;
	a.stride = padded;
	if (const char *e = getenv("POLAR_HIP_ROW_STRIDE")) // experiments: the caller's rows are this many bytes apart (a multiple of 16)
		if (direct && (size_t)atol(e) >= padded && atol(e) % 16 == 0)
			a.stride = (size_t)atol(e);
	a.groups = (u32)groups;
	a.top = lp.host.top;
	a.hot_top = v->rl;
	a.hot = nullptr;
	a.hot_dwords = 0;
	a.scratch = nullptr;
	a.block_dwords = 0;
	if (block_dwords) {
		a.block_dwords = block_dwords;
		if ((rc = h->scratch.reserve(blocks * team * block_dwords * sizeof(u32))))
			return rc;
		a.scratch = (u32 *)h->scratch.p;
		a.hot_top = hot_top_for(v, lp.host.top, blocks * team);
		if (a.hot_top > v->rl) { // the wavefronts' hot levels side by side in one compact region
			a.hot_dwords = ((size_t)v->q << (a.hot_top + 1)) - ((size_t)v->q << (v->rl + 1));
			if ((rc = h->hot.reserve(blocks * team * a.hot_dwords * sizeof(u32))))
				return rc;
			a.hot = (u32 *)h->hot.p;
		}
	}
	a.prof = nullptr;
#ifdef POLAR_PROFILE
	static unsigned long long *prof_dev = nullptr;
	if (!prof_dev)
		HIP_TRY(hipMalloc((void **)&prof_dev, 16 * 32 * 2 * 8));
	HIP_TRY(hipMemsetAsync(prof_dev, 0, 16 * 32 * 2 * 8, stream));
	a.prof = prof_dev;
#endif
	if (timed)
		HIP_TRY(hipEventRecord(h->ev0, stream));
	hipLaunchKernelGGL(v->fn, dim3((unsigned)blocks), dim3(64 * (unsigned)team), 0, stream, a);
	HIP_TRY(hipGetLastError());
#ifdef POLAR_PROFILE
	{
		static unsigned long long host[16 * 32 * 2];
		HIP_TRY(hipMemcpyAsync(host, prof_dev, sizeof(host), hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipStreamSynchronize(stream));
		static const char *names[16] = {"left", "right", "comb", "rate0", "rate1", "rep", "spc", "r0right", "r0comb", "r1comb", "?", "chain1", "chain2", "chain3", "chain4", "?"};
		unsigned long long total = 0;
		for (int i = 0; i < 16 * 32; ++i)
			total += host[2 * i];
		fprintf(stderr, "polar_hip profile: wave-cycles per node kind and level (blocks=%zu, total %.3e cycles = %.1f Mcycles per wave)\n", blocks, (double)total, total / 1e6 / blocks);
		for (int k = 0; k < 16; ++k)
			for (int l = 0; l < 32; ++l)
				if (host[2 * (k * 32 + l) + 1])
					fprintf(stderr, "  %-8s L%-2d calls/wave %8.1f  cycles/call %9.0f  share %5.2f%%\n", names[k], l, (double)host[2 * (k * 32 + l) + 1] / blocks,
						(double)host[2 * (k * 32 + l)] / host[2 * (k * 32 + l) + 1], 100.0 * host[2 * (k * 32 + l)] / total);
	}
#endif
	if (timed) {
		HIP_TRY(hipEventRecord(h->ev1, stream));
		h->timed = true;
	}
	if (!direct && rows)
		HIP_TRY(hipMemcpy2DAsync(message, frames, msg, padded, frames, rows, hipMemcpyDeviceToDevice, stream));
	return POLAR_OK;
}

static int check_decode_args(polar_hip_handle *h, int program_id, const void *message, const void *codeword, size_t frames)
{
	if (!h)
		return fail(POLAR_E_ARG, "polar_hip_decode: null handle");
	if (program_id < 0 || (size_t)program_id >= h->programs.size())
		return fail(POLAR_E_ARG, "polar_hip_decode: unknown program id");
	if (frames && (!codeword || (!message && h->programs[program_id].host.rows)))
		return fail(POLAR_E_ARG, "polar_hip_decode: null buffer");
	return POLAR_OK;
}

// True when the runtime knows `p` as page-locked host memory (polar_hip_host_alloc, hipHostMalloc,
// hipHostRegister): only then do asynchronous copies run beside the host thread.
static bool host_pinned(const void *p)
{
	hipPointerAttribute_t attr;
	if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
		(void)hipGetLastError(); // pageable memory is "invalid value" to this query
		return false;
	}
	return attr.type == hipMemoryTypeHost;
}

// frames per chunk of the host-buffer pipeline: about 1 GiB of LLRs, never so few frames that the
// decode of a chunk falls into the small-batch regime, whole groups of every kernel shape
static size_t pipeline_chunk(size_t n)
{
	if (const char *e = getenv("POLAR_HIP_HOST_CHUNK")) { // frames; tests and experiments only
		size_t c = (size_t)atoll(e);
		return c < 512 ? 512 : c - c % 512;
	}
	size_t c = ((size_t)1 << 30) / n;
	if (c < 16384)
		c = 16384;
	return c - c % 512;
}

extern "C" int polar_hip_where(const void *ptr)
{
	hipPointerAttribute_t attr;
	if (!ptr || hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
		(void)hipGetLastError(); // pageable host memory is "invalid value" to this query
		return POLAR_HOST;
	}
	return attr.type == hipMemoryTypeDevice ? POLAR_DEVICE : POLAR_HOST;
}

extern "C" void *polar_hip_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		fail(POLAR_E_NOMEM, "polar_hip_host_alloc: no page-locked memory of that size");
		return nullptr;
	}
	return p;
}

extern "C" void polar_hip_host_free(void *ptr)
{
	if (ptr)
		(void)hipHostFree(ptr);
}

enum { COPIED_IN = 0, DECODED = 1, COPIED_OUT = 2 }; // pipe_ev[slot][...]

static int pipeline_setup(polar_hip_handle *h, size_t n, size_t rows, size_t chunk)
{
	if (!h->pipe_in) {
		HIP_TRY(hipStreamCreateWithFlags(&h->pipe_in, hipStreamNonBlocking));
		HIP_TRY(hipStreamCreateWithFlags(&h->pipe_out, hipStreamNonBlocking));
		for (auto &slot : h->pipe_ev)
			for (hipEvent_t &e : slot)
				HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	}
	int rc;
	if ((rc = h->stage_in.reserve(2 * n * chunk)) || (rc = h->stage_out.reserve(2 * (rows ? rows : 1) * chunk)))
		return rc;
	return POLAR_OK;
}

// rows [0, count) split over a few host threads
template <class F>
static void parallel_rows(size_t count, F fn)
{
	static const unsigned want = [] {
		const char *e = getenv("POLAR_HIP_HOST_THREADS");
		unsigned t = e ? (unsigned)atoi(e) : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
		return t < 1 ? 1u : t;
	}();
	unsigned threads = (unsigned)std::min<size_t>(want, count ? count : 1);
	std::vector<std::thread> pool;
	for (unsigned t = 1; t < threads; ++t)
		pool.emplace_back([=] { fn(count * t / threads, count * (t + 1) / threads); });
	fn(0, count / threads);
	for (auto &th : pool)
		th.join();
}

// Pageable host buffers, several chunks' worth of frames: the column block of a chunk is gathered
// by host threads into a page-locked slot, copied in, decoded and copied out asynchronously, and
// scattered back into the caller's rows while the next chunk is on its way.
static int decode_host_staged_body(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, size_t chunk)
{
	const HostProgram &hp = h->programs[program_id].host;
	const size_t n = (size_t)1 << hp.top, rows = hp.rows;
	int rc;
	if ((rc = pipeline_setup(h, n, rows, chunk)))
		return rc;
	if (h->bounce_in_bytes < 2 * n * chunk) {
		if (h->bounce_in)
			(void)hipHostFree(h->bounce_in);
		h->bounce_in = nullptr;
		h->bounce_in_bytes = 0;
		HIP_TRY(hipHostMalloc(&h->bounce_in, 2 * n * chunk, hipHostMallocDefault));
		h->bounce_in_bytes = 2 * n * chunk;
	}
	if (rows && h->bounce_out_bytes < 2 * rows * chunk) {
		if (h->bounce_out)
			(void)hipHostFree(h->bounce_out);
		h->bounce_out = nullptr;
		h->bounce_out_bytes = 0;
		HIP_TRY(hipHostMalloc(&h->bounce_out, 2 * rows * chunk, hipHostMallocDefault));
		h->bounce_out_bytes = 2 * rows * chunk;
	}
	const size_t chunks = (frames + chunk - 1) / chunk;
	for (size_t i = 0; i <= chunks; ++i) {
		const int b = (int)(i & 1);
		if (i < chunks) {
			const size_t f0 = i * chunk, c = frames - f0 < chunk ? frames - f0 : chunk;
			int8_t *hin = (int8_t *)h->bounce_in + b * n * chunk, *hout = (int8_t *)h->bounce_out + b * rows * chunk;
			int8_t *din = (int8_t *)h->stage_in.p + b * n * chunk, *dout = (int8_t *)h->stage_out.p + b * rows * chunk;
			if (i >= 2) // the slot's previous chunk has left the page-locked buffer
				HIP_TRY(hipEventSynchronize(h->pipe_ev[b][COPIED_IN]));
			parallel_rows(n, [=](size_t r0, size_t r1) {
				for (size_t r = r0; r < r1; ++r)
					memcpy(hin + r * c, codeword + r * frames + f0, c);
			});
			if (i >= 2)
				HIP_TRY(hipStreamWaitEvent(h->pipe_in, h->pipe_ev[b][DECODED], 0));
			HIP_TRY(hipMemcpyAsync(din, hin, n * c, hipMemcpyHostToDevice, h->pipe_in));
			HIP_TRY(hipEventRecord(h->pipe_ev[b][COPIED_IN], h->pipe_in));
			HIP_TRY(hipStreamWaitEvent(h->stream, h->pipe_ev[b][COPIED_IN], 0));
			if (i >= 2 && rows)
				HIP_TRY(hipStreamWaitEvent(h->stream, h->pipe_ev[b][COPIED_OUT], 0));
			if ((rc = launch_decode(h, program_id, dout, din, c, h->stream, true)))
				return rc;
			HIP_TRY(hipEventRecord(h->pipe_ev[b][DECODED], h->stream));
			if (rows) { // the slot's previous messages were scattered in the iteration before this one
				HIP_TRY(hipStreamWaitEvent(h->pipe_out, h->pipe_ev[b][DECODED], 0));
				HIP_TRY(hipMemcpyAsync(hout, dout, rows * c, hipMemcpyDeviceToHost, h->pipe_out));
				HIP_TRY(hipEventRecord(h->pipe_ev[b][COPIED_OUT], h->pipe_out));
			}
		}
		if (i >= 1 && rows) {
			const int pb = (int)((i - 1) & 1);
			const size_t f0 = (i - 1) * chunk, c = frames - f0 < chunk ? frames - f0 : chunk;
			const int8_t *hout = (const int8_t *)h->bounce_out + pb * rows * chunk;
			HIP_TRY(hipEventSynchronize(h->pipe_ev[pb][COPIED_OUT]));
			parallel_rows(rows, [=](size_t r0, size_t r1) {
				for (size_t r = r0; r < r1; ++r)
					memcpy(message + r * frames + f0, hout + r * c, c);
			});
		}
	}
	HIP_TRY(hipStreamSynchronize(h->pipe_in));
	HIP_TRY(hipStreamSynchronize(h->stream));
	HIP_TRY(hipStreamSynchronize(h->pipe_out));
	return POLAR_OK;
}

// Page-locked host buffers, several chunks' worth of frames: chunk i+1 is copied in and chunk i-1
// copied out while chunk i decodes (three streams, two staging slots, column blocks of the
// [rows][frames] matrices as 2D copies).
static int decode_host_pipelined_body(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, size_t chunk)
{
	const HostProgram &hp = h->programs[program_id].host;
	const size_t n = (size_t)1 << hp.top, rows = hp.rows;
	int rc;
	if ((rc = pipeline_setup(h, n, rows, chunk)))
		return rc;
	size_t i = 0;
	for (size_t f0 = 0; f0 < frames; f0 += chunk, ++i) {
		const size_t c = frames - f0 < chunk ? frames - f0 : chunk;
		const int b = (int)(i & 1);
		int8_t *din = (int8_t *)h->stage_in.p + b * n * chunk, *dout = (int8_t *)h->stage_out.p + b * rows * chunk;
		if (i >= 2) // the slot's previous chunk has been decoded
			HIP_TRY(hipStreamWaitEvent(h->pipe_in, h->pipe_ev[b][DECODED], 0));
		HIP_TRY(hipMemcpy2DAsync(din, c, codeword + f0, frames, c, n, hipMemcpyHostToDevice, h->pipe_in));
		HIP_TRY(hipEventRecord(h->pipe_ev[b][COPIED_IN], h->pipe_in));
		HIP_TRY(hipStreamWaitEvent(h->stream, h->pipe_ev[b][COPIED_IN], 0));
		if (i >= 2 && rows) // ... and its messages have left the slot
			HIP_TRY(hipStreamWaitEvent(h->stream, h->pipe_ev[b][COPIED_OUT], 0));
		if ((rc = launch_decode(h, program_id, dout, din, c, h->stream, true))) // timed: polar_hip_last_decode_ms reports the last chunk
			return rc;
		HIP_TRY(hipEventRecord(h->pipe_ev[b][DECODED], h->stream));
		if (rows) {
			HIP_TRY(hipStreamWaitEvent(h->pipe_out, h->pipe_ev[b][DECODED], 0));
			HIP_TRY(hipMemcpy2DAsync(message + f0, frames, dout, c, c, rows, hipMemcpyDeviceToHost, h->pipe_out));
			HIP_TRY(hipEventRecord(h->pipe_ev[b][COPIED_OUT], h->pipe_out));
		}
	}
	HIP_TRY(hipStreamSynchronize(h->pipe_in));
	HIP_TRY(hipStreamSynchronize(h->stream));
	HIP_TRY(hipStreamSynchronize(h->pipe_out));
	return POLAR_OK;
}

// a failure in the middle of a pipeline leaves copies in flight that still name the caller's
// buffers and the page-locked slots: let them finish before the error goes back
static int drained(polar_hip_handle *h, int rc)
{
	if (rc && h->pipe_in) {
		const std::string why = g_error;
		(void)hipStreamSynchronize(h->pipe_in);
		(void)hipStreamSynchronize(h->stream);
		(void)hipStreamSynchronize(h->pipe_out);
		(void)hipGetLastError();
		g_error = why;
	}
	return rc;
}
static int decode_host_staged(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, size_t chunk)
{
	return drained(h, decode_host_staged_body(h, program_id, message, codeword, frames, chunk));
}
static int decode_host_pipelined(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, size_t chunk)
{
	return drained(h, decode_host_pipelined_body(h, program_id, message, codeword, frames, chunk));
}

extern "C" int polar_hip_decode(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, int where)
{
	int rc = check_decode_args(h, program_id, message, codeword, frames);
	if (rc)
		return rc;
	if (where != POLAR_HOST && where != POLAR_DEVICE)
		return fail(POLAR_E_ARG, "polar_hip_decode: bad `where`");
	if ((rc = use_device(h)))
		return rc;
	if (!frames)
		return POLAR_OK;
	const HostProgram &hp = h->programs[program_id].host;
	if (where == POLAR_DEVICE) {
		if ((rc = launch_decode(h, program_id, message, codeword, frames, h->stream, true)))
			return rc;
		HIP_TRY(hipStreamSynchronize(h->stream));
		return POLAR_OK;
	}
	const size_t chunk = pipeline_chunk((size_t)1 << hp.top);
	if (frames >= 2 * chunk) {
		if (host_pinned(codeword) && (!hp.rows || host_pinned(message)))
			return decode_host_pipelined(h, program_id, message, codeword, frames, chunk);
		if ((chunk << hp.top) <= ((size_t)1 << 30)) // two page-locked slots of a chunk each way: 3 GiB at most
			return decode_host_staged(h, program_id, message, codeword, frames, chunk);
	}
	size_t in_bytes = ((size_t)1 << hp.top) * frames, out_bytes = (size_t)hp.rows * frames;
	if ((rc = h->stage_in.reserve(in_bytes)) || (rc = h->stage_out.reserve(out_bytes ? out_bytes : 4)))
		return rc;
	HIP_TRY(hipMemcpyAsync(h->stage_in.p, codeword, in_bytes, hipMemcpyHostToDevice, h->stream));
	if ((rc = launch_decode(h, program_id, (int8_t *)h->stage_out.p, (const int8_t *)h->stage_in.p, frames, h->stream, true)))
		return rc;
	if (out_bytes)
		HIP_TRY(hipMemcpyAsync(message, h->stage_out.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

extern "C" int polar_hip_decode_async(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, void *stream)
{
	int rc = check_decode_args(h, program_id, message, codeword, frames);
	if (rc)
		return rc;
	if ((rc = use_device(h)))
		return rc;
	return launch_decode(h, program_id, message, codeword, frames, (hipStream_t)stream, false);
}

// No allocation on this path: fixed kernel shape (resident since polar_hip_load_program), arenas in
// the file-scope `scratch` array through the per-device pointer taken at polar_hip_create, no
// staging copies (so only whole groups and aligned rows), occupancy known since create.
extern "C" int polar_hip_decode_static(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, void *stream)
{
	int rc = check_decode_args(h, program_id, message, codeword, frames);
	if (rc)
		return rc;
	if ((rc = use_device(h)))
		return rc;
	if (!frames)
		return POLAR_OK;
	const size_t fpw = 4 * (size_t)STATIC_Q;
	LoadedProgram &lp = h->programs[program_id];
	// teams of four for whole teams whose arenas fit the array together, else single wavefronts
	const bool teams = frames % (fpw * STATIC_TEAM) == 0 &&
		(lp.host.top <= STATIC_RL || STATIC_SCRATCH_BYTES / (scratch_bytes_per_wave(STATIC_Q, lp.host.top) * STATIC_TEAM) >= 1);
	const KernelVariant *v = find_variant(STATIC_Q, STATIC_RL, teams ? STATIC_TEAM : 1);
	if (!v)
		return fail(POLAR_E_HIP, "polar_hip_decode_static: kernel shape not built into this library");
	const size_t team = (size_t)v->team;
	if (frames % fpw || (uintptr_t)codeword % 16 || (uintptr_t)message % 16)
		return fail(POLAR_E_ARG, "polar_hip_decode_static: frames must be a multiple of 32 and both buffers 16-byte aligned (this entry point cannot stage ragged batches: it does not allocate)");
	void *arena = h->device < MAX_DEVICES ? g_static_scratch[h->device] : nullptr;
	if (!arena)
		return fail(POLAR_E_HIP, "polar_hip_decode_static: no static scratch on this device");
	auto dev = lp.dev.find(program_shape(v));
	auto occ = h->occupancy.find((const void *)v->fn);
	if (dev == lp.dev.end() || occ == h->occupancy.end() || occ->second < 1)
		return fail(POLAR_E_ARG, "polar_hip_decode_static: program not resident for this entry point");
	const size_t groups = frames / fpw;
	if (groups > 0x7FFFFFFFu)
		return fail(POLAR_E_ARG, "too many frames for one call");
	size_t blocks = (size_t)occ->second * h->cus; // workgroups of `team` wavefronts
	DecodeArgs a;
	a.scratch = nullptr;
	a.block_dwords = 0;
	if (lp.host.top > v->rl) {
		const size_t per_wave = scratch_bytes_per_wave(v->q, lp.host.top);
		const size_t fit = STATIC_SCRATCH_BYTES / (per_wave * team);
		if (fit < 1)
			return fail(POLAR_E_ARG, "polar_hip_decode_static: one wavefront of this code needs " + std::to_string(per_wave) + " bytes of scratch, the static array holds " + std::to_string((size_t)STATIC_SCRATCH_BYTES) + " (codes up to N = 2^20)");
		if (blocks > fit)
			blocks = fit;
		a.scratch = (u32 *)arena; // the pointer, never the symbol
		a.block_dwords = per_wave / sizeof(u32);
	}
	if (blocks > groups / team)
		blocks = groups / team;
	a.prog = dev->second;
	a.cw = (const uint8_t *)codeword;
	a.msg = (uint8_t *)message;
	a.stride = frames;
	a.groups = (u32)groups;
	a.top = lp.host.top;
	a.hot_top = v->rl; // no hot region here: it would have to be carved out of the static array
	a.hot = nullptr;
	a.hot_dwords = 0;
	a.prof = nullptr;
	h->timed = false;
	hipLaunchKernelGGL(v->fn, dim3((unsigned)blocks), dim3(64 * (unsigned)team), 0, (hipStream_t)stream, a);
	HIP_TRY(hipGetLastError());
	return POLAR_OK;
}

extern "C" int polar_hip_last_decode_shape(polar_hip_handle *h, int *frames_per_wave, int *lds_level, int *team)
{
	if (!h || !h->last_shape[0])
		return fail(POLAR_E_ARG, "polar_hip_last_decode_shape: no decode on this handle");
	if (frames_per_wave)
		*frames_per_wave = h->last_shape[0];
	if (lds_level)
		*lds_level = h->last_shape[1];
	if (team)
		*team = h->last_shape[2];
	return POLAR_OK;
}

extern "C" float polar_hip_last_decode_ms(polar_hip_handle *h)
{
	if (!h || !h->timed)
		return (float)fail(POLAR_E_ARG, "polar_hip_last_decode_ms: no timed decode on this handle");
	float ms = 0;
	if (hipEventSynchronize(h->ev1) != hipSuccess || hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess)
		return (float)fail(POLAR_E_HIP, "polar_hip_last_decode_ms: event query failed");
	return ms;
}

extern "C" int polar_hip_synchronize(polar_hip_handle *h)
{
	if (!h)
		return fail(POLAR_E_ARG, "polar_hip_synchronize: null handle");
	int rc = use_device(h);
	if (rc)
		return rc;
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

// ---- float decoder ----

// enqueues the float decode of device buffers on the handle's stream (timed: polar_hip_last_decode_ms)
static int launch_decode_f32(polar_hip_handle *h, int program_id, float *msg, const float *cw, size_t frames)
{
	LoadedProgram &lp = h->programs[program_id];
	const size_t n = (size_t)1 << lp.host.top;
	const size_t groups = (frames + 63) / 64, per_wave = 3 * n * 64 * sizeof(float);
	if (groups > 0x7FFFFFFFu)
		return fail(POLAR_E_ARG, "too many frames for one call");
	if (!lp.dev_bytes) { // the float kernel interprets the reference byte-code as it is
		HIP_TRY(hipMalloc((void **)&lp.dev_bytes, lp.bytes.size()));
		HIP_TRY(hipMemcpy(lp.dev_bytes, lp.bytes.data(), lp.bytes.size(), hipMemcpyHostToDevice));
	}
	ScratchBudget budget{h};
	size_t blocks = (size_t)8 * h->cus < groups ? (size_t)8 * h->cus : groups;
	if (!budget.fits(blocks * per_wave)) {
		blocks = budget.limit() / per_wave;
		if (blocks < 1)
			return fail(POLAR_E_NOMEM, "not enough device memory for one wavefront's float arena");
	}
	int rc;
	if ((rc = h->scratch.reserve(blocks * per_wave)))
		return rc;
	FloatArgs a;
	a.prog = lp.dev_bytes;
	a.cw = cw;
	a.msg = msg;
	a.arena = (float *)h->scratch.p;
	a.frames = frames;
	a.groups = (u32)groups;
	a.top = lp.host.top;
	h->timed = false;
	HIP_TRY(hipEventRecord(h->ev0, h->stream));
	hipLaunchKernelGGL(sc_decode_f32_kernel, dim3((unsigned)blocks), dim3(64), 0, h->stream, a);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(h->ev1, h->stream));
	h->timed = true;
	return POLAR_OK;
}

extern "C" int polar_hip_decode_f32(polar_hip_handle *h, int program_id, float *message, const float *codeword, size_t frames, int where)
{
	int rc = check_decode_args(h, program_id, message, codeword, frames);
	if (rc)
		return rc;
	if (where != POLAR_HOST && where != POLAR_DEVICE)
		return fail(POLAR_E_ARG, "polar_hip_decode_f32: bad `where`");
	if ((rc = use_device(h)))
		return rc;
	if (!frames)
		return POLAR_OK;
	LoadedProgram &lp = h->programs[program_id];
	const size_t n = (size_t)1 << lp.host.top, rows = lp.host.rows;
	const float *cw = codeword;
	float *msg = message;
	if (where == POLAR_HOST) {
		if ((rc = h->stage_in.reserve(n * frames * sizeof(float))) || (rc = h->stage_out.reserve((rows ? rows : 1) * frames * sizeof(float))))
			return rc;
		HIP_TRY(hipMemcpyAsync(h->stage_in.p, codeword, n * frames * sizeof(float), hipMemcpyHostToDevice, h->stream));
		cw = (const float *)h->stage_in.p;
		msg = (float *)h->stage_out.p;
	}
	if ((rc = launch_decode_f32(h, program_id, msg, cw, frames)))
		return rc;
	if (where == POLAR_HOST && rows)
		HIP_TRY(hipMemcpyAsync(message, msg, rows * frames * sizeof(float), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

// ---- encoders ----

// device tables for a frozen set: src_row[N] (message row per position or -1) then info_pos[K]
static int upload_tables(polar_hip_handle *h, const uint8_t *frozen, int level, int *K_out)
{
	size_t n = (size_t)1 << level;
	std::vector<int> t(2 * n);
	int k = 0;
	for (size_t i = 0; i < n; ++i) {
		if (frozen[i])
			t[i] = -1;
		else {
			t[n + k] = (int)i;
			t[i] = k++;
		}
	}
	*K_out = k;
	if (h->frozen_cache.size() == n && !memcmp(h->frozen_cache.data(), frozen, n) && h->tables.p)
		return POLAR_OK;
	int rc = h->tables.reserve(2 * n * sizeof(int));
	if (rc)
		return rc;
	h->frozen_cache.clear();
	HIP_TRY(hipMemcpyAsync(h->tables.p, t.data(), 2 * n * sizeof(int), hipMemcpyHostToDevice, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream)); // `t` goes out of scope
	h->frozen_cache.assign(frozen, frozen + n);
	return POLAR_OK;
}

static const int BFLY_C = 4;

template <typename T>
static int transform_passes(polar_hip_handle *h, T *cw, const T *msg, const int *src_row, size_t cols, int level, int first_mode)
{
	int mode = first_mode;
	for (int s0 = 0; s0 < level; s0 += BFLY_C) {
		int c = level - s0 < BFLY_C ? level - s0 : BFLY_C;
		size_t blocks = (((size_t)1 << level) >> c) * ((cols + 255) / 256);
		if (blocks > 0x7FFFFFFFu)
			return fail(POLAR_E_ARG, "polar_hip_encode: too many frames for one call");
		hipLaunchKernelGGL((butterfly_kernel<T, BFLY_C>), dim3((unsigned)blocks), dim3(256), 0, h->stream, cw, msg, src_row, cols, level, s0, mode);
		HIP_TRY(hipGetLastError());
		mode = LOAD_INPLACE;
	}
	return POLAR_OK;
}

// codeword (device, N x frames) <- encode(message (device, K x frames))
static int encode_packed(polar_hip_handle *h, int systematic, int8_t *cw, const int8_t *msg, int level, size_t frames);

static int encode_device(polar_hip_handle *h, int systematic, int8_t *cw, const int8_t *msg, int level, size_t frames)
{
	static const bool bytes_only = getenv("POLAR_HIP_ENCODE_BYTES") != nullptr; // experiments only
	if (frames % 32 == 0 && (uintptr_t)cw % 16 == 0 && (uintptr_t)msg % 16 == 0 && !bytes_only)
		return encode_packed(h, systematic, cw, msg, level, frames);
	// ragged batches and unaligned buffers: byte symbols in place
	const int *src_row = (const int *)h->tables.p;
	bool wide = frames % 4 == 0 && (uintptr_t)cw % 4 == 0 && (uintptr_t)msg % 4 == 0;
	int rc;
	for (int round = 0; round <= (systematic ? 1 : 0); ++round) {
		int mode = round ? LOAD_REFREEZE : LOAD_MESSAGE;
		if (wide)
			rc = transform_passes<u32>(h, (u32 *)cw, (const u32 *)msg, src_row, frames / 4, level, mode);
		else
			rc = transform_passes<uint8_t>(h, (uint8_t *)cw, (const uint8_t *)msg, src_row, frames, level, mode);
		if (rc)
			return rc;
	}
	return POLAR_OK;
}

extern "C" int polar_hip_encode(polar_hip_handle *h, int systematic, int8_t *codeword, const int8_t *message, const uint8_t *frozen, int level, size_t frames, int where)
{
	if (!h || !frozen || level < 1 || level > MAX_LEVEL || (where != POLAR_HOST && where != POLAR_DEVICE))
		return fail(POLAR_E_ARG, "polar_hip_encode: bad argument");
	if (frames && !codeword)
		return fail(POLAR_E_ARG, "polar_hip_encode: null codeword");
	int rc = use_device(h), K = 0;
	if (rc)
		return rc;
	if (!frames)
		return POLAR_OK;
	if ((rc = upload_tables(h, frozen, level, &K)))
		return rc;
	if (K && !message)
		return fail(POLAR_E_ARG, "polar_hip_encode: null message");
	size_t n = (size_t)1 << level;
	if (where == POLAR_DEVICE) {
		if ((rc = encode_device(h, systematic, codeword, message, level, frames)))
			return rc;
		HIP_TRY(hipStreamSynchronize(h->stream));
		return POLAR_OK;
	}
	if ((rc = h->stage_in.reserve(n * frames)) || (rc = h->stage_out.reserve((size_t)(K ? K : 1) * frames)))
		return rc;
	if (K)
		HIP_TRY(hipMemcpyAsync(h->stage_out.p, message, (size_t)K * frames, hipMemcpyHostToDevice, h->stream));
	if ((rc = encode_device(h, systematic, (int8_t *)h->stage_in.p, (const int8_t *)h->stage_out.p, level, frames)))
		return rc;
	HIP_TRY(hipMemcpyAsync(codeword, h->stage_in.p, n * frames, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

// decoded (device, K x frames, u domain) -> message estimate (testbench.cc:177-183): re-encode and
// gather the information rows through packed passes; `tmp` holds N x ceil(frames / 32) PackedRows.
// Output: int8 symbols in `out` (may alias `decoded`) and / or the two bit planes. Tables must be current.
static const int PACK_C_MAX = 6;
// stages per pass: spread evenly over the fewest passes of at most `cmax` stages
static int pack_cmax()
{
	static int c = 0;
	if (!c) {
		const char *e = getenv("POLAR_HIP_PACK_C"); // experiments only
		c = e ? atoi(e) : 5;
		if (c < 1 || c > PACK_C_MAX)
			c = 5;
	}
	return c;
}
static size_t packed_cols(size_t frames) { return (frames + 31) / 32; }
static size_t packed_tmp_bytes(int level, size_t frames) { return level > pack_cmax() ? ((size_t)1 << level) * packed_cols(frames) * sizeof(PackedRow) : 0; } // also covers the tile passes (one pass up to level 10)

// one transform over packed rows: `load` for the first pass, `store` for the last one
static int packed_transform(polar_hip_handle *h, PackedPass a, int level, int load, int store, uint8_t *out, u32 *out_sign, u32 *out_eras, const char *who)
{
	static const bool tiles = !getenv("POLAR_HIP_PACK_C"); // experiments: the register-only passes for every size
	if (tiles && level >= 6) { // tile passes through LDS: one pass up to 2^10 rows (4 columns = 128 bytes of symbols per row), then at most 8 stages per pass
		static const int single = getenv("POLAR_HIP_TILE_SINGLE") ? atoi(getenv("POLAR_HIP_TILE_SINGLE")) : 10; // experiments only
		static const int smax = getenv("POLAR_HIP_TILE_S") ? atoi(getenv("POLAR_HIP_TILE_S")) : 8;
		const int passes = level <= single ? 1 : (level + smax - 1) / smax;
		for (int s0 = 0, pass = 0; s0 < level; ++pass) {
			int S = (level - s0 + (passes - pass) - 1) / (passes - pass);
			if (const char *e = getenv("POLAR_HIP_TILE_FIRST")) // experiments only: stages of the first pass
				if (pass == 0 && passes == 2 && atoi(e) >= 4 && atoi(e) <= 10 && level - atoi(e) >= 4 && level - atoi(e) <= 10)
					S = atoi(e);
			const bool last = s0 + S >= level;
			int CW = TILE_ITEMS >> S > 64 ? 64 : TILE_ITEMS >> S;
			while (CW > 1 && (size_t)CW / 2 >= a.cols)
				CW /= 2;
			a.s0 = s0;
			a.S = S;
			a.CW = CW;
			a.first = s0 == 0 ? load : PACK_LOAD_TMP;
			a.last = last ? store : PACK_STORE_TMP;
			a.out = last ? out : nullptr;
			a.out_sign = last ? out_sign : nullptr;
			a.out_eras = last ? out_eras : nullptr;
			size_t blocks = (((size_t)1 << level) >> S) * ((a.cols + CW - 1) / CW);
			if (blocks > 0x7FFFFFFFu)
				return fail(POLAR_E_ARG, std::string(who) + ": too many frames for one call");
			hipLaunchKernelGGL(packed_tile_kernel, dim3((unsigned)blocks), dim3((unsigned)((CW << S) / TILE_ROWS)), 0, h->stream, a);
			HIP_TRY(hipGetLastError());
			s0 += S;
		}
		return POLAR_OK;
	}
	const int passes = (level + pack_cmax() - 1) / pack_cmax();
	for (int s0 = 0, pass = 0; s0 < level; ++pass) {
		int c = (level - s0 + (passes - pass) - 1) / (passes - pass); // what is left, spread over the passes left
		bool last = s0 + c >= level;
		a.s0 = s0;
		a.first = s0 == 0 ? load : PACK_LOAD_TMP;
		a.last = last ? store : PACK_STORE_TMP;
		a.out = last ? out : nullptr;
		a.out_sign = last ? out_sign : nullptr;
		a.out_eras = last ? out_eras : nullptr;
		size_t items = (((size_t)1 << level) >> c) * a.cols, blocks = (items + 255) / 256;
		if (blocks > 0x7FFFFFFFu)
			return fail(POLAR_E_ARG, std::string(who) + ": too many frames for one call");
		switch (c) {
		case 1: hipLaunchKernelGGL(packed_transform_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, h->stream, a); break;
		case 2: hipLaunchKernelGGL(packed_transform_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, h->stream, a); break;
		case 3: hipLaunchKernelGGL(packed_transform_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, h->stream, a); break;
		case 4: hipLaunchKernelGGL(packed_transform_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, h->stream, a); break;
		case 5: hipLaunchKernelGGL(packed_transform_kernel<5>, dim3((unsigned)blocks), dim3(256), 0, h->stream, a); break;
		default: hipLaunchKernelGGL(packed_transform_kernel<6>, dim3((unsigned)blocks), dim3(256), 0, h->stream, a); break;
		}
		HIP_TRY(hipGetLastError());
		s0 += c;
	}
	return POLAR_OK;
}

static int extract_device(polar_hip_handle *h, int8_t *out, u32 *out_sign, u32 *out_eras, const int8_t *decoded, void *tmp, int level, size_t frames)
{
	PackedPass a = {};
	a.tmp = (PackedRow *)tmp;
	a.msg = (const uint8_t *)decoded;
	a.src_row = (const int *)h->tables.p;
	a.frames = frames;
	a.cols = packed_cols(frames);
	a.level = level;
	a.wide = frames % 32 == 0 && (uintptr_t)decoded % 16 == 0 && (uintptr_t)out % 16 == 0;
	return packed_transform(h, a, level, PACK_LOAD_MESSAGE, PACK_STORE_INFO, (uint8_t *)out, out_sign, out_eras, "polar_hip_systematic_extract");
}

// Encoders on packed rows (whole 32-frame columns, 16-byte aligned buffers): K bytes in, N bytes
// out and N/4 bytes per intermediate pass instead of N in and out per pass of butterfly_kernel.
static int encode_packed(polar_hip_handle *h, int systematic, int8_t *cw, const int8_t *msg, int level, size_t frames)
{
	PackedPass a = {};
	a.msg = (const uint8_t *)msg;
	a.src_row = (const int *)h->tables.p;
	a.frames = frames;
	a.cols = packed_cols(frames);
	a.level = level;
	a.wide = 1;
	int rc = h->enc_tmp.reserve(((size_t)1 << level) * a.cols * sizeof(PackedRow) + 16);
	if (rc)
		return rc;
	a.tmp = (PackedRow *)h->enc_tmp.p;
	if (!systematic)
		return packed_transform(h, a, level, PACK_LOAD_MESSAGE, PACK_STORE_ALL, (uint8_t *)cw, nullptr, nullptr, "polar_hip_encode");
	if ((rc = packed_transform(h, a, level, PACK_LOAD_MESSAGE, PACK_STORE_TMP, nullptr, nullptr, nullptr, "polar_hip_encode")))
		return rc;
	return packed_transform(h, a, level, PACK_LOAD_REFREEZE, PACK_STORE_ALL, (uint8_t *)cw, nullptr, nullptr, "polar_hip_encode");
}

// shared body of polar_hip_systematic_extract and polar_hip_systematic_extract_packed
static int extract_entry(polar_hip_handle *h, int8_t *out, uint8_t *sign, uint8_t *eras, const int8_t *decoded, const uint8_t *frozen, int level, size_t frames, int where, const char *who)
{
	if (!h || !frozen || level < 1 || level > MAX_LEVEL || (where != POLAR_HOST && where != POLAR_DEVICE))
		return fail(POLAR_E_ARG, std::string(who) + ": bad argument");
	int rc = use_device(h), K = 0;
	if (rc)
		return rc;
	if (!frames)
		return POLAR_OK;
	if ((rc = upload_tables(h, frozen, level, &K)))
		return rc;
	if (!K)
		return POLAR_OK;
	if (!decoded || (!out && !sign))
		return fail(POLAR_E_ARG, std::string(who) + ": null buffer");
	if ((rc = h->enc_tmp.reserve(packed_tmp_bytes(level, frames) + 16)))
		return rc;
	const size_t sym_bytes = (size_t)K * frames, plane_bytes = (size_t)K * packed_cols(frames) * sizeof(u32);
	if (where == POLAR_DEVICE) {
		if (((uintptr_t)sign | (uintptr_t)eras) % 4)
			return fail(POLAR_E_ARG, std::string(who) + ": bit planes must be 4-byte aligned");
		if ((rc = extract_device(h, out, (u32 *)sign, (u32 *)eras, decoded, h->enc_tmp.p, level, frames)))
			return rc;
		HIP_TRY(hipStreamSynchronize(h->stream));
		return POLAR_OK;
	}
	if ((rc = h->stage_out.reserve(sym_bytes)) || (rc = h->stage_in.reserve(2 * plane_bytes + 16)))
		return rc;
	int8_t *dev = (int8_t *)h->stage_out.p;
	u32 *dsign = sign ? (u32 *)h->stage_in.p : nullptr, *deras = eras ? (u32 *)((uint8_t *)h->stage_in.p + plane_bytes) : nullptr;
	HIP_TRY(hipMemcpyAsync(dev, decoded, sym_bytes, hipMemcpyHostToDevice, h->stream));
	if ((rc = extract_device(h, out ? dev : nullptr, dsign, deras, dev, h->enc_tmp.p, level, frames)))
		return rc;
	if (out)
		HIP_TRY(hipMemcpyAsync(out, dev, sym_bytes, hipMemcpyDeviceToHost, h->stream));
	if (sign)
		HIP_TRY(hipMemcpyAsync(sign, dsign, plane_bytes, hipMemcpyDeviceToHost, h->stream));
	if (eras)
		HIP_TRY(hipMemcpyAsync(eras, deras, plane_bytes, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

extern "C" int polar_hip_systematic_extract(polar_hip_handle *h, int8_t *decoded, const uint8_t *frozen, int level, size_t frames, int where)
{
	return extract_entry(h, decoded, nullptr, nullptr, decoded, frozen, level, frames, where, "polar_hip_systematic_extract");
}

extern "C" int polar_hip_systematic_extract_packed(polar_hip_handle *h, uint8_t *sign_plane, uint8_t *erasure_plane, const int8_t *decoded, const uint8_t *frozen, int level, size_t frames, int where)
{
	if (!sign_plane)
		return fail(POLAR_E_ARG, "polar_hip_systematic_extract_packed: null sign plane");
	return extract_entry(h, nullptr, sign_plane, erasure_plane, decoded, frozen, level, frames, where, "polar_hip_systematic_extract_packed");
}

extern "C" int polar_hip_unpack_symbols(polar_hip_handle *h, int8_t *symbols, const uint8_t *sign_plane, const uint8_t *erasure_plane, size_t rows, size_t frames, int where)
{
	if (!h || (where != POLAR_HOST && where != POLAR_DEVICE) || (rows && frames && (!symbols || !sign_plane)))
		return fail(POLAR_E_ARG, "polar_hip_unpack_symbols: bad argument");
	int rc = use_device(h);
	if (rc)
		return rc;
	if (!rows || !frames)
		return POLAR_OK;
	const size_t cols = packed_cols(frames), plane_bytes = rows * cols * sizeof(u32), sym_bytes = rows * frames;
	uint8_t *dsym = (uint8_t *)symbols;
	const u32 *dsign = (const u32 *)sign_plane, *deras = (const u32 *)erasure_plane;
	if (where == POLAR_HOST) {
		if ((rc = h->stage_out.reserve(sym_bytes)) || (rc = h->stage_in.reserve(2 * plane_bytes + 16)))
			return rc;
		dsym = (uint8_t *)h->stage_out.p;
		HIP_TRY(hipMemcpyAsync(h->stage_in.p, sign_plane, plane_bytes, hipMemcpyHostToDevice, h->stream));
		dsign = (const u32 *)h->stage_in.p;
		if (erasure_plane) {
			HIP_TRY(hipMemcpyAsync((uint8_t *)h->stage_in.p + plane_bytes, erasure_plane, plane_bytes, hipMemcpyHostToDevice, h->stream));
			deras = (const u32 *)((uint8_t *)h->stage_in.p + plane_bytes);
		}
	} else if (((uintptr_t)sign_plane | (uintptr_t)erasure_plane) % 4) {
		return fail(POLAR_E_ARG, "polar_hip_unpack_symbols: bit planes must be 4-byte aligned");
	}
	size_t blocks = (rows * cols + 255) / 256;
	if (blocks > 0x7FFFFFFFu)
		return fail(POLAR_E_ARG, "polar_hip_unpack_symbols: too many symbols for one call");
	int wide = frames % 32 == 0 && (uintptr_t)dsym % 16 == 0;
	hipLaunchKernelGGL(unpack_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, dsym, dsign, deras, rows, frames, cols, wide);
	HIP_TRY(hipGetLastError());
	if (where == POLAR_HOST)
		HIP_TRY(hipMemcpyAsync(symbols, dsym, sym_bytes, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

// ---- bench counters ----

static int count_device(polar_hip_handle *h, unsigned long long *c, const uint8_t *noisy, const uint8_t *orig, size_t n_code, const uint8_t *decoded, const uint8_t *message, size_t n_mesg)
{
	size_t work = (n_code > n_mesg ? n_code : n_mesg) / 16 / 256 + 1;
	unsigned blocks = (unsigned)(work < (size_t)h->cus * 8 ? work : (size_t)h->cus * 8);
	hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(256), 0, h->stream, c, noisy, orig, n_code, decoded, message, n_mesg);
	HIP_TRY(hipGetLastError());
	return POLAR_OK;
}

extern "C" int polar_hip_count(polar_hip_handle *h, int64_t *counters, const int8_t *noisy, const int8_t *orig, size_t n_code, const int8_t *decoded, const int8_t *message, size_t n_mesg, int where)
{
	if (!h || !counters || (where != POLAR_HOST && where != POLAR_DEVICE) || (n_code && (!noisy || !orig)) || (n_mesg && (!decoded || !message)))
		return fail(POLAR_E_ARG, "polar_hip_count: bad argument");
	int rc = use_device(h);
	if (rc)
		return rc;
	const uint8_t *p[4] = {(const uint8_t *)noisy, (const uint8_t *)orig, (const uint8_t *)decoded, (const uint8_t *)message};
	unsigned long long *c = (unsigned long long *)counters;
	if (where == POLAR_HOST) {
		// four staged arrays share stage_in; counters are accumulated on the device then added
		size_t off[5] = {0, n_code, 2 * n_code, 2 * n_code + n_mesg, 2 * (n_code + n_mesg)};
		for (int k = 0; k < 5; ++k)
			off[k] = (off[k] + 15) / 16 * 16 + 16 * k;
		if ((rc = h->stage_in.reserve(off[4] + 16)) || (rc = h->counters.reserve(32)))
			return rc;
		for (int k = 0; k < 4; ++k) {
			size_t n = k < 2 ? n_code : n_mesg;
			if (n)
				HIP_TRY(hipMemcpyAsync((uint8_t *)h->stage_in.p + off[k], p[k], n, hipMemcpyHostToDevice, h->stream));
			p[k] = (const uint8_t *)h->stage_in.p + off[k];
		}
		c = (unsigned long long *)h->counters.p;
		HIP_TRY(hipMemsetAsync(c, 0, 32, h->stream));
	}
	if ((rc = count_device(h, c, p[0], p[1], n_code, p[2], p[3], n_mesg)))
		return rc;
	if (where == POLAR_HOST) {
		int64_t tmp[4];
		HIP_TRY(hipMemcpyAsync(tmp, c, 32, hipMemcpyDeviceToHost, h->stream));
		HIP_TRY(hipStreamSynchronize(h->stream));
		for (int k = 0; k < 4; ++k)
			counters[k] += tmp[k];
		return POLAR_OK;
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

// ---- message source, channel, and the whole bench loop on the device ----

static int launch_channel(polar_hip_handle *h, int mode, int8_t *out, const int8_t *cw, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t frame0, unsigned long long *counts = nullptr)
{
	double sigma = channel_sigma(snr_db), fact = 2.0 / (sigma * sigma);
	size_t items = rows * (((frame0 + frames - 1) >> 2) - (frame0 >> 2) + 1);
	size_t want = (items + 255) / 256, cap = (size_t)h->cus * 32;
	unsigned blocks = (unsigned)(want < cap ? want : cap);
	// Screening threshold in LLR units. The single-precision normal is good to 1e-5 absolute
	// (|n| < 6.7; logf, sqrtf, sincospif within 2 ulp; u1 near 1 excluded in normal_pair_f32), which
	// the LLR scale multiplies by fact * sigma; the LLR arithmetic itself adds 128 * 2^-22. Four
	// times that bound is asked for. No screening when it would reject most values anyway.
	const bool f64_only = getenv("POLAR_HIP_CHANNEL_F64") != nullptr; // tests and experiments
	double bound = 4.0 * (1e-5 * fact * sigma + 128.0 / 4194304.0);
	float thr = f64_only || bound > 0.05 ? 0.0f : (float)bound;
	int wide = frames % 4 == 0 && frame0 % 4 == 0 && (uintptr_t)out % 4 == 0 && (uintptr_t)cw % 4 == 0;
	if (mode)
		hipLaunchKernelGGL(channel_kernel<1>, dim3(blocks), dim3(256), 0, h->stream, out, cw, rows, frames, sigma, fact, seed, frame0, thr, wide, counts);
	else
		hipLaunchKernelGGL(channel_kernel<0>, dim3(blocks), dim3(256), 0, h->stream, out, cw, rows, frames, sigma, fact, seed, frame0, thr, wide, counts);
	HIP_TRY(hipGetLastError());
	return POLAR_OK;
}

static int channel_entry(polar_hip_handle *h, int mode, int8_t *out, const int8_t *cw, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t frame0, int where, const char *who)
{
	if (!h || (where != POLAR_HOST && where != POLAR_DEVICE) || rows > 0xFFFFFFFFu || (rows && frames && (!out || (!mode && !cw))))
		return fail(POLAR_E_ARG, who);
	int rc = use_device(h);
	if (rc)
		return rc;
	if (!rows || !frames)
		return POLAR_OK;
	size_t bytes = rows * frames;
	if (where == POLAR_DEVICE) {
		if ((rc = launch_channel(h, mode, out, cw, rows, frames, snr_db, seed, frame0)))
			return rc;
		HIP_TRY(hipStreamSynchronize(h->stream));
		return POLAR_OK;
	}
	if ((rc = h->stage_in.reserve(bytes)) || (rc = h->stage_out.reserve(bytes)))
		return rc;
	if (!mode)
		HIP_TRY(hipMemcpyAsync(h->stage_in.p, cw, bytes, hipMemcpyHostToDevice, h->stream));
	if ((rc = launch_channel(h, mode, (int8_t *)h->stage_out.p, (const int8_t *)h->stage_in.p, rows, frames, snr_db, seed, frame0)))
		return rc;
	HIP_TRY(hipMemcpyAsync(out, h->stage_out.p, bytes, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

extern "C" int polar_hip_random_message(polar_hip_handle *h, int8_t *message, size_t rows, size_t frames, uint64_t seed, uint64_t first_frame, int where)
{
	return channel_entry(h, 1, message, nullptr, rows, frames, 0.0, seed, first_frame, where, "polar_hip_random_message: bad argument");
}

extern "C" int polar_hip_channel(polar_hip_handle *h, int8_t *llr, const int8_t *codeword, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t first_frame, int where)
{
	return channel_entry(h, 0, llr, codeword, rows, frames, snr_db, seed, first_frame, where, "polar_hip_channel: bad argument");
}

extern "C" int polar_hip_simulate(polar_hip_handle *h, int program_id, const uint8_t *frozen, int level, int systematic, double snr_db, uint64_t seed, uint64_t first_frame, size_t frames, int64_t *counters, int where)
{
	if (!h || !frozen || !counters || level < 1 || level > MAX_LEVEL || (where != POLAR_HOST && where != POLAR_DEVICE))
		return fail(POLAR_E_ARG, "polar_hip_simulate: bad argument");
	if (program_id < 0 || (size_t)program_id >= h->programs.size() || h->programs[program_id].host.top != level)
		return fail(POLAR_E_ARG, "polar_hip_simulate: program id unknown or not of this level");
	int rc = use_device(h), K = 0;
	if (rc)
		return rc;
	if ((rc = upload_tables(h, frozen, level, &K)))
		return rc;
	if ((uint32_t)K != h->programs[program_id].host.rows)
		return fail(POLAR_E_PROGRAM, "polar_hip_simulate: program and frozen set disagree on K");
	const size_t n = (size_t)1 << level;
	if ((rc = h->counters.reserve(64)))
		return rc;
	unsigned long long *c = where == POLAR_DEVICE ? (unsigned long long *)counters : (unsigned long long *)h->counters.p;
	if (where == POLAR_HOST)
		HIP_TRY(hipMemsetAsync(c, 0, 48, h->stream));
	// chunk: whole 1024-frame blocks within a quarter of the free memory (and at most 64 GiB)
	size_t free_b = 0, total_b = 0;
	HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	size_t held = h->sim_cw.bytes + h->sim_llr.bytes + h->sim_msg.bytes + h->sim_dec.bytes;
	size_t budget = held + free_b / 4;
	if (budget > ((size_t)64 << 30))
		budget = (size_t)64 << 30;
	size_t per_frame = 2 * n + 2 * (size_t)(K ? K : 1);
	size_t chunk = budget / per_frame;
	if (chunk >= 1024)
		chunk -= chunk % 1024;
	if (chunk < 1)
		return fail(POLAR_E_NOMEM, "polar_hip_simulate: not enough device memory for one frame");
	if (chunk > frames)
		chunk = frames;
	if (chunk >= 1024 && frames > chunk) { // equal chunks instead of full ones and a small rest (which would decode in the small-batch regime)
		size_t pieces = (frames + chunk - 1) / chunk, even = (frames + pieces - 1) / pieces;
		even = (even + 1023) / 1024 * 1024;
		if (even < chunk)
			chunk = even;
	}
	if (frames) {
		if ((rc = h->sim_cw.reserve(n * chunk > packed_tmp_bytes(level, chunk) ? n * chunk : packed_tmp_bytes(level, chunk))) || (rc = h->sim_llr.reserve(n * chunk)) || (rc = h->sim_msg.reserve((size_t)(K ? K : 1) * chunk)) || (rc = h->sim_dec.reserve((size_t)(K ? K : 1) * chunk)))
			return rc;
	}
	int8_t *cw = (int8_t *)h->sim_cw.p, *llr = (int8_t *)h->sim_llr.p, *msg = (int8_t *)h->sim_msg.p, *dec = (int8_t *)h->sim_dec.p;
	for (size_t a = 0; a < frames; a += chunk) {
		size_t f = frames - a < chunk ? frames - a : chunk;
		uint64_t f0 = first_frame + a;
		if (K && (rc = launch_channel(h, 1, msg, nullptr, (size_t)K, f, 0.0, seed, f0)))
			return rc;
		if ((rc = encode_device(h, systematic, cw, msg, level, f)))
			return rc;
		if ((rc = launch_channel(h, 0, llr, cw, n, f, snr_db, seed, f0, c))) // counts the channel errors and erasures as it goes
			return rc;
		if ((rc = launch_decode(h, program_id, dec, llr, f, h->stream, true))) // timed: polar_hip_last_decode_ms reports the last chunk
			return rc;
		if (systematic && K && (rc = extract_device(h, dec, nullptr, nullptr, dec, cw, level, f)))
			return rc;
		if (K && (rc = count_device(h, c, nullptr, nullptr, 0, (const uint8_t *)dec, (const uint8_t *)msg, (size_t)K * f)))
			return rc;
	}
	hipLaunchKernelGGL(add_counters_kernel, dim3(1), dim3(1), 0, h->stream, c, (unsigned long long)frames, (unsigned long long)frames * (unsigned long long)K);
	HIP_TRY(hipGetLastError());
	if (where == POLAR_HOST) {
		int64_t tmp[6];
		HIP_TRY(hipMemcpyAsync(tmp, c, 48, hipMemcpyDeviceToHost, h->stream));
		HIP_TRY(hipStreamSynchronize(h->stream));
		for (int k = 0; k < 6; ++k)
			counters[k] += tmp[k];
		return POLAR_OK;
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

// ---------------------------------------------------------------------------------------
// float path around the float decoder: the reference selects it with
// `typedef float code_type` (testbench.cc:49-53) and every functor follows. Symbols (+1 / -1 / 0)
// are exact in either type, so the encoders run on the int8 kernels between two conversions; the
// channel writes the unquantised LLR (PolarHelper<float>::quant is the identity, polar_helper.hh:28-32).
// ---------------------------------------------------------------------------------------

// out = in as int8 symbols; *bad is raised when a value is none of +1 / -1 / 0 (the int8 kernels
// multiply symbols, the reference's float encoder would multiply whatever it is given)
__global__ void __launch_bounds__(256) f32_to_symbols_kernel(int8_t *out, const float *in, size_t n, unsigned *bad)
{
	unsigned b = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		float v = in[i];
		b |= !(v == 1.0f || v == -1.0f || v == 0.0f);
		out[i] = (int8_t)((v > 0.0f) - (v < 0.0f));
	}
	if (b && bad)
		atomicOr(bad, 1u);
}
__global__ void __launch_bounds__(256) symbols_to_f32_kernel(float *out, const int8_t *in, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
		out[i] = (float)in[i];
}
// llr = (float)(fact * (x + sigma * n)) with the int8 channel's draws (channel.hpp), in double as the
// bench does (testbench.cc:151-165); counts[0] += #(llr * x < 0), counts[1] += #(llr == 0)
__global__ void __launch_bounds__(256) channel_f32_kernel(float *out, const int8_t *cw, size_t rows, size_t frames, double sigma, double fact, uint64_t seed, uint64_t frame0, unsigned long long *counts)
{
	unsigned n_flipped = 0, n_zero = 0;
	const uint64_t q0 = frame0 >> 2, nq = ((frame0 + frames - 1) >> 2) - q0 + 1;
	const size_t items = rows * nq, nt = (size_t)gridDim.x * blockDim.x;
	for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += nt) {
		const size_t i = it / nq;
		const uint64_t quad = q0 + it % nq;
		PhiloxBlock b = draw(seed, STREAM_NOISE, (uint32_t)i, quad);
		double n[4];
		normal_pair(b.v[0], b.v[1], n[0], n[1]);
		normal_pair(b.v[2], b.v[3], n[2], n[3]);
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			uint64_t g = 4 * quad + c;
			if (g >= frame0 && g < frame0 + frames) {
				size_t at = i * frames + (size_t)(g - frame0);
				int8_t x = cw[at];
				float l = (float)(fact * ((double)x + sigma * n[c]));
				out[at] = l;
				n_flipped += l * (float)x < 0.0f;
				n_zero += l == 0.0f;
			}
		}
	}
	if (counts) {
		if (n_flipped)
			atomicAdd(&counts[0], (unsigned long long)n_flipped);
		if (n_zero)
			atomicAdd(&counts[1], (unsigned long long)n_zero);
	}
}

static unsigned grid_for(const polar_hip_handle *h, size_t n)
{
	size_t want = (n + 255) / 256, cap = (size_t)h->cus * 32;
	return (unsigned)(want < cap ? (want ? want : 1) : cap);
}

/* PolarEncoder / PolarSysEnc with TYPE = float lanes (polar_encoder.hh:9-59 is generic in TYPE). */
extern "C" int polar_hip_encode_f32(polar_hip_handle *h, int systematic, float *codeword, const float *message, const uint8_t *frozen, int level, size_t frames, int where)
{
	if (!h || !frozen || level < 1 || level > MAX_LEVEL || (where != POLAR_HOST && where != POLAR_DEVICE))
		return fail(POLAR_E_ARG, "polar_hip_encode_f32: bad argument");
	if (frames && !codeword)
		return fail(POLAR_E_ARG, "polar_hip_encode_f32: null codeword");
	int rc = use_device(h), K = 0;
	if (rc)
		return rc;
	if (!frames)
		return POLAR_OK;
	if ((rc = upload_tables(h, frozen, level, &K)))
		return rc;
	if (K && !message)
		return fail(POLAR_E_ARG, "polar_hip_encode_f32: null message");
	const size_t n = (size_t)1 << level, km = (size_t)(K ? K : 1) * frames;
	// f32_a: int8 codeword then int8 message then the flag word; f32_b: float staging for host callers
	if ((rc = h->f32_a.reserve(n * frames + km + 16)))
		return rc;
	int8_t *cw8 = (int8_t *)h->f32_a.p, *msg8 = cw8 + n * frames;
	unsigned *bad = (unsigned *)(((uintptr_t)(msg8 + km) + 3) / 4 * 4);
	HIP_TRY(hipMemsetAsync(bad, 0, 4, h->stream));
	const float *msg = message;
	float *cw = codeword;
	if (where == POLAR_HOST) {
		if ((rc = h->f32_b.reserve((n > (size_t)K ? n : (size_t)K) * frames * sizeof(float))))
			return rc;
		if (K)
			HIP_TRY(hipMemcpyAsync(h->f32_b.p, message, (size_t)K * frames * sizeof(float), hipMemcpyHostToDevice, h->stream));
		msg = (const float *)h->f32_b.p;
		cw = (float *)h->f32_b.p;
	}
	if (K) {
		hipLaunchKernelGGL(f32_to_symbols_kernel, dim3(grid_for(h, (size_t)K * frames)), dim3(256), 0, h->stream, msg8, msg, (size_t)K * frames, bad);
		HIP_TRY(hipGetLastError());
	}
	if ((rc = encode_device(h, systematic, cw8, msg8, level, frames)))
		return rc;
	hipLaunchKernelGGL(symbols_to_f32_kernel, dim3(grid_for(h, n * frames)), dim3(256), 0, h->stream, cw, cw8, n * frames);
	HIP_TRY(hipGetLastError());
	unsigned flag = 0;
	HIP_TRY(hipMemcpyAsync(&flag, bad, 4, hipMemcpyDeviceToHost, h->stream));
	if (where == POLAR_HOST)
		HIP_TRY(hipMemcpyAsync(codeword, cw, n * frames * sizeof(float), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	if (flag)
		return fail(POLAR_E_ARG, "polar_hip_encode_f32: message symbols must be +1, -1 or 0");
	return POLAR_OK;
}

/* The bench loop of one SNR point with code_type = float (testbench.cc:49-53, 125-193): message
 * source -> encoder -> BPSK/AWGN with unquantised float LLRs -> float SC decoder -> (systematic:
 * re-encode + gather) -> the four counters. Same counter-based message and noise streams as
 * polar_hip_simulate, so the two decoders meet the very same frames. counters as polar_hip_simulate. */
extern "C" int polar_hip_simulate_f32(polar_hip_handle *h, int program_id, const uint8_t *frozen, int level, int systematic, double snr_db, uint64_t seed, uint64_t first_frame, size_t frames, int64_t *counters, int where)
{
	if (!h || !frozen || !counters || level < 1 || level > MAX_LEVEL || (where != POLAR_HOST && where != POLAR_DEVICE))
		return fail(POLAR_E_ARG, "polar_hip_simulate_f32: bad argument");
	if (program_id < 0 || (size_t)program_id >= h->programs.size() || h->programs[program_id].host.top != level)
		return fail(POLAR_E_ARG, "polar_hip_simulate_f32: program id unknown or not of this level");
	int rc = use_device(h), K = 0;
	if (rc)
		return rc;
	if ((rc = upload_tables(h, frozen, level, &K)))
		return rc;
	if ((uint32_t)K != h->programs[program_id].host.rows)
		return fail(POLAR_E_PROGRAM, "polar_hip_simulate_f32: program and frozen set disagree on K");
	const size_t n = (size_t)1 << level, kk = (size_t)(K ? K : 1);
	if ((rc = h->counters.reserve(64)))
		return rc;
	unsigned long long *c = where == POLAR_DEVICE ? (unsigned long long *)counters : (unsigned long long *)h->counters.p;
	if (where == POLAR_HOST)
		HIP_TRY(hipMemsetAsync(c, 0, 48, h->stream));
	// chunks of whole 64-frame wavefront groups: the float decoder keeps 3 N floats per frame in its arena
	size_t chunk = (size_t)8 * h->cus * 64 * 4;
	while (chunk > 64 && chunk * n * 6 > ((size_t)8 << 30))
		chunk /= 2;
	if (chunk > frames)
		chunk = frames;
	if (frames) {
		size_t tmpb = packed_tmp_bytes(level, chunk);
		if ((rc = h->sim_cw.reserve(n * chunk > tmpb ? n * chunk : tmpb)) || (rc = h->sim_msg.reserve(kk * chunk)) || (rc = h->sim_dec.reserve(kk * chunk)) ||
		    (rc = h->f32_a.reserve(n * chunk * sizeof(float))) || (rc = h->f32_b.reserve(kk * chunk * sizeof(float))))
			return rc;
	}
	int8_t *cw = (int8_t *)h->sim_cw.p, *msg = (int8_t *)h->sim_msg.p, *dec = (int8_t *)h->sim_dec.p;
	float *llr = (float *)h->f32_a.p, *decf = (float *)h->f32_b.p;
	const double sigma = channel_sigma(snr_db), fact = 2.0 / (sigma * sigma);
	for (size_t a = 0; a < frames; a += chunk) {
		size_t f = frames - a < chunk ? frames - a : chunk;
		uint64_t f0 = first_frame + a;
		if (K && (rc = launch_channel(h, 1, msg, nullptr, (size_t)K, f, 0.0, seed, f0)))
			return rc;
		if ((rc = encode_device(h, systematic, cw, msg, level, f)))
			return rc;
		size_t items = n * (((f0 + f - 1) >> 2) - (f0 >> 2) + 1);
		hipLaunchKernelGGL(channel_f32_kernel, dim3(grid_for(h, items)), dim3(256), 0, h->stream, llr, cw, n, f, sigma, fact, seed, f0, c);
		HIP_TRY(hipGetLastError());
		if ((rc = launch_decode_f32(h, program_id, decf, llr, f)))
			return rc;
		if (K) {
			hipLaunchKernelGGL(f32_to_symbols_kernel, dim3(grid_for(h, (size_t)K * f)), dim3(256), 0, h->stream, dec, decf, (size_t)K * f, (unsigned *)nullptr);
			HIP_TRY(hipGetLastError());
			if (systematic && (rc = extract_device(h, dec, nullptr, nullptr, dec, cw, level, f)))
				return rc;
			if ((rc = count_device(h, c, nullptr, nullptr, 0, (const uint8_t *)dec, (const uint8_t *)msg, (size_t)K * f)))
				return rc;
		}
	}
	hipLaunchKernelGGL(add_counters_kernel, dim3(1), dim3(1), 0, h->stream, c, (unsigned long long)frames, (unsigned long long)frames * (unsigned long long)K);
	HIP_TRY(hipGetLastError());
	if (where == POLAR_HOST) {
		int64_t tmp[6];
		HIP_TRY(hipMemcpyAsync(tmp, c, 48, hipMemcpyDeviceToHost, h->stream));
		HIP_TRY(hipStreamSynchronize(h->stream));
		for (int k = 0; k < 6; ++k)
			counters[k] += tmp[k];
		return POLAR_OK;
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

/* The float bench's channel alone (testbench.cc:151-165 with code_type = float): llr = (float)(2 / sigma^2 *
 * (codeword + sigma * n)), the same noise draws as polar_hip_channel. codeword: +1 / -1 / 0 floats;
 * llr may be the same buffer (the reference overwrites its codeword, testbench.cc:164-165). */
extern "C" int polar_hip_channel_f32(polar_hip_handle *h, float *llr, const float *codeword, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t first_frame, int where)
{
	if (!h || (where != POLAR_HOST && where != POLAR_DEVICE))
		return fail(POLAR_E_ARG, "polar_hip_channel_f32: bad argument");
	const size_t n = rows * frames;
	if (n && (!llr || !codeword))
		return fail(POLAR_E_ARG, "polar_hip_channel_f32: null buffer");
	if (rows > 0xFFFFFFFFull)
		return fail(POLAR_E_ARG, "polar_hip_channel_f32: too many rows");
	int rc = use_device(h);
	if (rc)
		return rc;
	if (!n)
		return POLAR_OK;
	if ((rc = h->f32_a.reserve(n + 16)))
		return rc;
	int8_t *cw8 = (int8_t *)h->f32_a.p;
	unsigned *bad = (unsigned *)(((uintptr_t)(cw8 + n) + 3) / 4 * 4);
	HIP_TRY(hipMemsetAsync(bad, 0, 4, h->stream));
	const float *src = codeword;
	float *dst = llr;
	if (where == POLAR_HOST) {
		if ((rc = h->f32_b.reserve(n * sizeof(float))))
			return rc;
		HIP_TRY(hipMemcpyAsync(h->f32_b.p, codeword, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
		src = dst = (float *)h->f32_b.p;
	}
	hipLaunchKernelGGL(f32_to_symbols_kernel, dim3(grid_for(h, n)), dim3(256), 0, h->stream, cw8, src, n, bad);
	HIP_TRY(hipGetLastError());
	const double sigma = channel_sigma(snr_db), fact = 2.0 / (sigma * sigma);
	size_t items = rows * (((first_frame + frames - 1) >> 2) - (first_frame >> 2) + 1);
	hipLaunchKernelGGL(channel_f32_kernel, dim3(grid_for(h, items)), dim3(256), 0, h->stream, dst, cw8, rows, frames, sigma, fact, seed, first_frame, (unsigned long long *)nullptr);
	HIP_TRY(hipGetLastError());
	unsigned flag = 0;
	HIP_TRY(hipMemcpyAsync(&flag, bad, 4, hipMemcpyDeviceToHost, h->stream));
	if (where == POLAR_HOST)
		HIP_TRY(hipMemcpyAsync(llr, dst, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	if (flag)
		return fail(POLAR_E_ARG, "polar_hip_channel_f32: codeword symbols must be +1, -1 or 0");
	return POLAR_OK;
}

// ---------------------------------------------------------------------------------------
// Counter exchange for callers that shard the frames over GPUs themselves (SURVEY 8 e): the one
// collective of the path, an all-reduce (sum) of the int64 counters of testbench.cc:119-125, 185-192.
// RCCL is not linked: the library that made the caller's communicator is already in the process (or
// librccl.so is loaded here), and its ncclAllReduce is looked up once.
// ---------------------------------------------------------------------------------------
#include <dlfcn.h>

typedef int (*nccl_allreduce_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*nccl_errstr_t)(int);

extern "C" int polar_hip_allreduce_counters(polar_hip_handle *h, int64_t *counters, size_t count, void *nccl_comm)
{
	if (!h || !counters || !count || !nccl_comm)
		return fail(POLAR_E_ARG, "polar_hip_allreduce_counters: bad argument");
	if (polar_hip_where(counters) != POLAR_DEVICE)
		return fail(POLAR_E_ARG, "polar_hip_allreduce_counters: counters must be device memory (polar_hip_simulate / polar_hip_count with POLAR_DEVICE)");
	static nccl_allreduce_t allreduce = nullptr;
	static nccl_errstr_t errstr = nullptr;
	if (!allreduce) {
		void *lib = nullptr;
		if (dlsym(RTLD_DEFAULT, "ncclAllReduce")) // the caller linked RCCL: its symbols are global
			lib = RTLD_DEFAULT;
		const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
		for (int k = 0; !lib && k < 3; ++k)
			lib = dlopen(names[k], RTLD_NOW | RTLD_NOLOAD); // loaded by the caller without global symbols (dlopen, Python)
		for (int k = 0; !lib && k < 3; ++k)
			lib = dlopen(names[k], RTLD_NOW);
		if (!lib)
			return fail(POLAR_E_ARG, "polar_hip_allreduce_counters: RCCL (librccl.so) is not in this process and cannot be loaded");
		allreduce = (nccl_allreduce_t)dlsym(lib, "ncclAllReduce");
		errstr = (nccl_errstr_t)dlsym(lib, "ncclGetErrorString");
		if (!allreduce)
			return fail(POLAR_E_ARG, "polar_hip_allreduce_counters: ncclAllReduce not found in RCCL");
	}
	int rc = use_device(h);
	if (rc)
		return rc;
	const int nccl_int64 = 4, nccl_sum = 0; // ncclInt64, ncclSum (rccl.h)
	int st = allreduce(counters, counters, count, nccl_int64, nccl_sum, nccl_comm, h->stream);
	if (st) {
		static thread_local char msg[160];
		snprintf(msg, sizeof msg, "polar_hip_allreduce_counters: ncclAllReduce failed (%d): %s", st, errstr ? errstr(st) : "?");
		return fail(POLAR_E_HIP, msg);
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}
