// tests/emu/emu_decode.cc -- TEST INFRASTRUCTURE ONLY.
//
// Compiles polar_amd/csrc/sc_core.hpp for the host: the 64 lanes of a wavefront run one after
// another inside every PFOR_LANES region, and the handful of gfx950 instructions the core
// relies on (v_pk_add_i16 clamp, v_pk_sub_i16 clamp, v_pk_min/max_i16, v_perm_b32) are
// emulated from their ISA definitions. This lets the CPU test-suite check the decoder's bit
// tricks and addressing against the oracle without a GPU. It is never part of the product.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define POLAR_EMU 1
#define PDEV inline
#define PFOR_LANES(lane) for (int lane = 0; lane < 64; ++lane)

namespace polar {
typedef uint32_t u32;
struct alignas(16) u32x4 { u32 x, y, z, w; };
static inline u32x4 operator^(u32x4 a, u32x4 b) { return u32x4{a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w}; }
static inline u32x4 operator|(u32x4 a, u32x4 b) { return u32x4{a.x | b.x, a.y | b.y, a.z | b.z, a.w | b.w}; }
static inline u32x4 operator&(u32x4 a, u32x4 b) { return u32x4{a.x & b.x, a.y & b.y, a.z & b.z, a.w & b.w}; }
typedef u32 lds_u32;
typedef u32x4 lds_u32x4;

static inline int16_t sat16(int v) { return (int16_t)(v < -32768 ? -32768 : v > 32767 ? 32767 : v); }
#define EMU_PK(name, expr) \
	static inline u32 name(u32 a, u32 b) \
	{ \
		u32 r = 0; \
		for (int h = 0; h < 2; ++h) { \
			int x = (int16_t)(a >> (16 * h)), y = (int16_t)(b >> (16 * h)); \
			r |= (u32)(uint16_t)(expr) << (16 * h); \
		} \
		return r; \
	}
EMU_PK(pk_add_sat, sat16(x + y))
EMU_PK(pk_sub_sat, sat16(x - y))
EMU_PK(pk_min, (int16_t)(x < y ? x : y))
EMU_PK(pk_max, (int16_t)(x > y ? x : y))
// V_PERM_B32: byte select from {S0,S1} (S0 = bytes 4..7), 8..11 replicate a sign bit, 12 = 0x00, >=13 = 0xFF
static inline u32 perm(u32 s0, u32 s1, u32 sel)
{
	uint64_t in = ((uint64_t)s0 << 32) | s1;
	u32 r = 0;
	for (int k = 0; k < 4; ++k) {
		u32 s = (sel >> (8 * k)) & 0xFF, b;
		if (s >= 13) b = 0xFF;
		else if (s == 12) b = 0;
		else if (s >= 8) b = ((in >> (16 * (s - 8) + 15)) & 1) ? 0xFF : 0;
		else b = (u32)(in >> (8 * s)) & 0xFF;
		r |= b << (8 * k);
	}
	return r;
}
static inline void wave_sync() {}
static inline void team_sync() {}
static inline u32 wave_any(u32 v) { return v != 0; }
}

#include "../../polar_amd/csrc/sc_core.hpp"
#include "../../polar_amd/csrc/program.hpp"

using namespace polar;

template <int Q, int RL>
static int run(int8_t *msg, const int8_t *cw, const HostProgram &hp, size_t frames)
{
	typedef Group<Q, RL> G;
	const size_t FR = 4 * Q, n = (size_t)1 << hp.top;
	// ragged batches go through a padded copy, exactly as the host wrapper of the library does
	size_t fp = (frames + FR - 1) / FR * FR;
	std::vector<int8_t> pin(n * fp, 0), pout((size_t)(hp.rows ? hp.rows : 1) * fp, 0);
	for (size_t i = 0; i < n; ++i)
		memcpy(&pin[i * fp], cw + i * frames, frames);
	// same arena layout as the kernel (polar_hip.hip): soft, sign plane, zero plane; 16-byte aligned
	const size_t nsoft = (size_t)Q << hp.top, nplane = nsoft / 8 + 1;
	std::vector<u32x4> lds_((G::LDS_DWORDS + 3) / 4), soft_((nsoft + 3) / 4), planes_((2 * nplane + 3) / 4 + 1);
	struct Span { u32 *p; size_t n; u32 *data() { return p; } u32 *begin() { return p; } u32 *end() { return p + n; } size_t size() const { return n; } };
	Span lds{(u32 *)lds_.data(), (size_t)G::LDS_DWORDS}, soft{(u32 *)soft_.data(), nsoft}, planes{(u32 *)planes_.data(), 2 * nplane};
	for (size_t g0 = 0; g0 < fp; g0 += FR) {
		// poison everything a fresh wave would find uninitialised
		for (auto &v : lds) v = 0xDEADBEEF;
		for (auto &v : soft) v = 0xDEADBEEF;
		for (auto &v : planes) v = 0xDEADBEEF;
		G g;
		g.sl = lds.data();
		g.hls = g.sl + G::LDS_SOFT;
		g.hlz = g.hls + G::LDS_PLANE;
		g.sg = soft.data();
		g.hgs = planes.data();
		g.hgz = planes.data() + nsoft / 8;
		g.top = hp.top;
		// two hot levels in a block of their own, as the kernel places them (decode_kernel.hpp)
		g.hot_top = RL + 2 < hp.top ? RL + 2 : hp.top - 1;
		g.sh = nullptr;
		std::vector<u32x4> hot_;
		if (g.hot_top > RL) {
			hot_.assign((((size_t)Q << (g.hot_top + 1)) - ((size_t)Q << (RL + 1)) + 3) / 4, u32x4{0xDEADBEEF, 0xDEADBEEF, 0xDEADBEEF, 0xDEADBEEF});
			g.sh = (u32 *)hot_.data() - ((size_t)Q << (RL + 1));
		}
		g.in = InVec<Q>{(const uint8_t *)pin.data() + g0, fp};
		g.out = OutRows<Q>{(uint8_t *)pout.data() + g0, fp};
		g.run(hp.ops.data());
	}
	for (size_t r = 0; r < hp.rows; ++r)
		memcpy(msg + r * frames, &pout[r * fp], frames);
	return 0;
}

extern "C" int emu_decode(int8_t *msg, const int8_t *cw, const uint8_t *prog, int len, size_t frames, int Q, int RL)
{
	HostProgram hp;
	int err;
#define CASE(q, r) \
	if (Q == q && RL == r) { \
		if ((err = translate_program(prog, len, hp, q, Group<q, r>::CHAIN_DEPTH))) \
			return -err; \
		return run<q, r>(msg, cw, hp, frames) ? -100 : (int)hp.rows; \
	}
	CASE(1, 4) CASE(1, 6) CASE(2, 5) CASE(4, 4) CASE(4, 7) CASE(4, 9) CASE(8, 6) CASE(16, 7) CASE(16, 5) CASE(32, 5) CASE(32, 6)
#undef CASE
	return -101;
}

extern "C" unsigned emu_op(int which, unsigned a, unsigned b, unsigned hs, unsigned hz)
{
	switch (which) {
	case 0: return prod4(a, b);
	case 1: return qadd4(a, b);
	case 2: return madd4(hs, hz, a, b);
	case 3: return zero_bytes(a);
	case 4: return symbols(a, b);
	}
	return 0;
}

// number of device instruction words with opcode `opcode` (any if < 0) after translation for a
// wavefront of q frame quads: lets the tests see that the host-side fusions took place
extern "C" int emu_op_count(const uint8_t *prog, int len, int q, int opcode)
{
	HostProgram hp;
	if (translate_program(prog, len, hp, q, CHAIN_MAX))
		return -1;
	int n = 0;
	for (const DevOp &d : hp.ops)
		n += (d.op_level & 255) != OP_END && (opcode < 0 || (int)(d.op_level & 255) == opcode);
	return n;
}
