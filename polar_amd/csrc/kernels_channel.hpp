// kernels_channel.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): message source and channel kernels.
#pragma once

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
