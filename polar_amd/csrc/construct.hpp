// construct.hpp -- host-side code construction and program compilation (product code).
//
// These stay on the CPU exactly as in the reference: they run once per code, not per frame.
//   bec_channels()   Bhattacharyya / BEC recursion of polar_freezer.hh:13-21 and :37-45
//   code_const0()    PolarCodeConst0<MAX_M>::operator()   polar_freezer.hh:49-61
//   freezer()        PolarFreezer::operator()             polar_freezer.hh:23-31
//   compile()        PolarCompiler::operator()            polar_compiler.hh:21-58
// `long double` is kept as the arithmetic type because the frozen set of the reference is
// defined by it (x87 80-bit on the x86 hosts this runs on).
#pragma once

#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace polar {

// erasure probability of every synthetic channel, natural index order: splitting a channel
// of erasure probability p gives the worse channel p(2-p) at the lower index and the
// better one p*p at the upper index.
inline void bec_channels(std::vector<long double> &pe, int level, long double p)
{
	size_t n = (size_t)1 << level;
	pe.resize(n);
	// the recursion applies one split per index bit, most significant bit first
	for (size_t i = 0; i < n; ++i) {
		long double v = p;
		for (int bit = level - 1; bit >= 0; --bit)
			v = (i >> bit) & 1 ? v * v : v * (2 - v);
		pe[i] = v;
	}
}

// Returns 1 when the choice was ambiguous: the K-th and (K+1)-th most reliable channels have
// exactly the same erasure probability, so which of them carries information is left to
// std::nth_element in the reference (polar_freezer.hh:56, no tie-break); here the lower index wins.
inline int code_const0(uint8_t *frozen, int level, int K, long double p)
{
	size_t n = (size_t)1 << level;
	std::vector<long double> pe;
	bec_channels(pe, level, p);
	std::vector<int> order(n);
	std::iota(order.begin(), order.end(), 0);
	auto better = [&pe](int a, int b) { return pe[a] < pe[b] || (pe[a] == pe[b] && a < b); };
	if (K > 0 && (size_t)K < n)
		std::nth_element(order.begin(), order.begin() + K, order.end(), better);
	for (size_t i = 0; i < n; ++i)
		frozen[order[i]] = i >= (size_t)K;
	if (K <= 0 || (size_t)K >= n)
		return 0;
	long double worst_info = pe[order[0]]; // nth_element leaves the first K unordered
	for (int i = 1; i < K; ++i)
		worst_info = std::max(worst_info, pe[order[i]]);
	return worst_info == pe[order[K]];
}

inline int freezer(uint8_t *frozen, int level, long double p, long double threshold)
{
	size_t n = (size_t)1 << level;
	std::vector<long double> pe;
	bec_channels(pe, level, p);
	int K = 0;
	for (size_t i = 0; i < n; ++i) {
		frozen[i] = pe[i] > threshold;
		K += !frozen[i];
	}
	return K;
}

// Node classification in the reference's rule order (polar_compiler.hh:26-48). Returns the
// number of bytes written or -1 (the reference asserts level > 0, polar_compiler.hh:23).
class ProgramBuilder {
	const uint8_t *frozen;
	std::vector<uint32_t> sum; // sum[i] = frozen positions below i
	uint8_t *out;
	bool bad = false;
	uint32_t count(size_t a, size_t b) const { return sum[b] - sum[a]; }
	void node(size_t base, int level)
	{
		if (level < 1) {
			bad = true;
			return;
		}
		size_t half = (size_t)1 << (level - 1), mid = base + half, end = mid + half;
		uint32_t l = count(base, mid), r = count(mid, end);
		if (l == half && r == half)
			*out++ = 3; // rate0
		else if (!l && !r)
			*out++ = 4; // rate1
		else if (l == half && r == half - 1 && !frozen[end - 1])
			*out++ = 5; // rep
		else if (l == 1 && !r && frozen[base])
			*out++ = 6; // spc
		else if (l == half) {
			*out++ = 7; // rate0_right
			node(mid, level - 1);
			*out++ = 8; // rate0_comb
		} else if (!r) {
			*out++ = 0; // left
			node(base, level - 1);
			*out++ = 9; // rate1_comb
		} else {
			*out++ = 0; // left
			node(base, level - 1);
			*out++ = 1; // right
			node(mid, level - 1);
			*out++ = 2; // comb
		}
	}
public:
	int run(uint8_t *program, const uint8_t *frozen_bits, int level)
	{
		size_t n = (size_t)1 << level;
		frozen = frozen_bits;
		sum.assign(n + 1, 0);
		for (size_t i = 0; i < n; ++i)
			sum[i + 1] = sum[i] + (frozen[i] != 0);
		out = program;
		*out++ = (uint8_t)level;
		node(0, level);
		*out++ = 255;
		return bad ? -1 : (int)(out - program);
	}
};

} // namespace polar
