// channel.hpp -- the bench's message source and channel on the device (testbench.cc:112-117,
// 126-127, 151-165): msg = 1 - 2*bit, y = x + N(0, sigma^2) in double, LLR = quant(2/sigma^2 * y),
// quant(x) = clamp(nearbyint(x), -128, 127) (polar_helper.hh:194-198).
//
// The reference draws from std::mt19937 / std::normal_distribution, whose stream cannot be
// split over GPUs. Here every value is a pure function of (seed, stream, element, frame):
// Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) keyed by
// the seed, counter = (frame quad low, frame quad high, element, stream), so any partition of
// the frames over wavefronts, launches or GPUs produces the same batch. One Philox block
// serves the four frames of a quad: two Box-Muller pairs for the noise, four bits for messages.
#pragma once

#include <math.h>
#include <stdint.h>

namespace polar {

#ifdef __HIPCC__
#define CH_FN __host__ __device__ inline
#else
#define CH_FN inline
#endif

enum : uint32_t { STREAM_NOISE = 0, STREAM_MESSAGE = 1 };

struct PhiloxBlock {
	uint32_t v[4];
};

CH_FN PhiloxBlock philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1)
{
	for (int r = 0; r < 10; ++r) {
		uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
		uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
		c1 = (uint32_t)p1;
		c3 = (uint32_t)p0;
		c0 = n0;
		c2 = n2;
		k0 += 0x9E3779B9u;
		k1 += 0xBB67AE85u;
	}
	return PhiloxBlock{{c0, c1, c2, c3}};
}

CH_FN PhiloxBlock draw(uint64_t seed, uint32_t stream, uint32_t element, uint64_t quad)
{
	return philox4x32_10((uint32_t)quad, (uint32_t)(quad >> 32), element, stream, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// two standard normals from two 32-bit draws (Box-Muller, double)
CH_FN void normal_pair(uint32_t a, uint32_t b, double &n0, double &n1)
{
	const double S = 1.0 / 4294967296.0;
	double u1 = ((double)a + 0.5) * S, u2 = ((double)b + 0.5) * S;
	double r = sqrt(-2.0 * log(u1)), t = 6.283185307179586476925286766559 * u2;
	n0 = r * cos(t);
	n1 = r * sin(t);
}

CH_FN int8_t quant_llr(double x)
{
	double r = nearbyint(x); // round-half-even, the default rounding mode (polar_helper.hh:196)
	return (int8_t)(r < -128.0 ? -128.0 : r > 127.0 ? 127.0 : r);
}

// noise standard deviation and LLR scale for Es/N0 in dB (testbench.cc:112-114, 162-163)
CH_FN double channel_sigma(double snr_db) { return sqrt(1.0 / (2.0 * pow(10.0, snr_db / 10.0))); }

} // namespace polar
