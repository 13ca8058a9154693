// kernels_float.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): the float decoder kernel.
#pragma once

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
