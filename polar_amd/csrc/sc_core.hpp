// sc_core.hpp -- wavefront-level successive-cancellation decoder core for gfx950 (MI355X).
//
// One 64-lane wavefront owns a GROUP of G = 4*Q codewords (frames) and interprets the
// reference's node program (polar_compiler.hh:11-13 opcodes) for all of them in lock-step.
//
// Data layout ("frame-packed"): one dword carries the same code element of 4 consecutive
// frames (one int8 LLR per byte), exactly the bytes of the reference's element-major /
// frame-minor buffers (testbench.cc:134-139), so input rows and message rows move as whole
// dwords with no transposition. A level-l LLR vector of the group is the linear dword array
//        v[i*Q + q]     i = element 0..2^l-1, q = frame quad 0..Q-1
// stored at dword offset Q*2^l of the group's soft arena -- the reference's soft[] indexing
// (polar_decoder.hh:128) scaled by Q. Levels <= RL live in LDS, levels RL+1..top-1 in a
// per-wave HBM scratch arena (coalesced 16-byte accesses), level `top` is the caller's input.
// Every node function of polar_decoder.hh:28-127 therefore becomes an element-wise pass
// over linear arrays with all 64 lanes busy whenever Q*2^(l-1) >= 64.
//
// Arithmetic: packed 16-bit VALU ops. A dword is split into two 2xi16 registers holding
// the int8 values in the HIGH byte of each half ("scaled by 256"): v_pk_add_i16/v_pk_sub_i16
// with clamp then saturate exactly like int8 (vqadd, simd.hh:1120-1127), -(-128) saturates
// to +127 (vqabs, simd.hh:292-299), and v_perm_b32 splits / re-interleaves the bytes.
//
// Partial sums ("hard", values in {-1,0,+1}, polar_decoder.hh:129) are kept as two bit
// planes (sign, zero). Eight consecutive elements share one dword per frame quad:
//        bit (8*k + 7 - j)  <->  element 8*o + j of frame 4*q + k     (plane dword o*Q + q)
// so `plane << j` puts element j's bits of the 4 frames on the byte sign bits, where
// v_perm_b32's sign-replicating selectors turn them into 16-bit lane masks.
//
// Address spaces are explicit on the hot paths (ds_* for LDS, global_* for scratch, s_load for
// the instruction stream); only the rare nodes above the LDS levels use flat pointers.
//
// This header is compiled for the device by polar_hip.hip. tests/emu/ compiles the same text
// for the host with POLAR_EMU defined (lanes run one after another inside PFOR_LANES) so the
// bit tricks can be unit-tested without a GPU; the product never uses that build.
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace polar {

typedef uint32_t u32;

enum : u32 {
	OP_LEFT = 0, OP_RIGHT = 1, OP_COMB = 2, OP_RATE0 = 3, OP_RATE1 = 4, OP_REP = 5, OP_SPC = 6,
	OP_RATE0_RIGHT = 7, OP_RATE0_COMB = 8, OP_RATE1_COMB = 9,
	// device-only: a run of left / right / rate0_right nodes descending level by level, fused by
	// the host into one pass (program.hpp); `pad` holds depth and stage kinds
	OP_CHAIN = 16,
	// device-only: a whole composite node of level SUBTREE_LEVEL (its left / right / comb ops and
	// every node below them), decoded per frame-quad lane in registers; `pad` = frozen mask of its
	// positions (bit j = position j frozen), from which the lanes re-derive the compiler's own
	// node decomposition (polar_compiler.hh:26-48); the host fuses only where that derivation
	// reproduces the program's bytes (program.hpp)
	OP_SUBTREE = 17,
	OP_END = 255
};
#ifndef POLAR_TEAM_HOT
#define POLAR_TEAM_HOT 0
#endif
#ifndef POLAR_HOT_ENABLE
#define POLAR_HOT_ENABLE 1
#endif
enum { CHAIN_MAX = 4 }; // per kernel variant: Group::CHAIN_DEPTH
#ifndef POLAR_SUBTREE_MAX_Q
#define POLAR_SUBTREE_MAX_Q 16
#endif
enum { SUBTREE_LEVEL = 3, SUBTREE_MAX_Q = POLAR_SUBTREE_MAX_Q }; // OP_SUBTREE roots: level 3 (one plane octet) and level 4 (two)
// DevOp.pad of OP_CHAIN: bits 0-2 depth D, bits 3-4 kind of stage 1 (0 = f, 1 = g, 2 = qadd),
// bit 4+k kind of stage k >= 2 (0 = f, 1 = qadd), bit 12+k: output of stage k is not stored (it
// only feeds a rate0_right stage of the same chain)


// Device instruction word: the reference byte-code with the interpreter's running state
// (lvl / hrd / msg of polar_decoder.hh:133-134) resolved at load time by the host.
struct alignas(16) DevOp {
	u32 op_level; // opcode | node level << 8
	u32 hrd;      // absolute position of the node in the partial-sum array
	u32 row;      // first message row the node emits
	u32 pad;
};

#ifndef POLAR_EMU
#define PDEV __device__ __forceinline__
// The lane id is re-materialised (opaque to the optimiser) in every parallel region: otherwise
// LICM hoists the per-lane address arithmetic of every node type out of the interpreter loop
// and the kernel needs > 300 VGPRs.
__device__ __forceinline__ int lane_id_()
{
	int l = (int)(threadIdx.x & 63);
	asm volatile("" : "+v"(l));
	return l;
}
#define PFOR_LANES(lane) for (int lane = lane_id_(), once_ = 1; once_; once_ = 0)
typedef short v2s_ __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32 lds_u32;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef const __attribute__((address_space(4))) DevOp *prog_ptr; // constant memory: scalar loads
PDEV u32 pk_add_sat(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_add_sat(__builtin_bit_cast(v2s_, a), __builtin_bit_cast(v2s_, b))); }
PDEV u32 pk_sub_sat(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_sub_sat(__builtin_bit_cast(v2s_, a), __builtin_bit_cast(v2s_, b))); }
PDEV u32 pk_min(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_min(__builtin_bit_cast(v2s_, a), __builtin_bit_cast(v2s_, b))); }
PDEV u32 pk_max(u32 a, u32 b) { return __builtin_bit_cast(u32, __builtin_elementwise_max(__builtin_bit_cast(v2s_, a), __builtin_bit_cast(v2s_, b))); }
PDEV u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
// all lanes of the wave have executed everything before this point, LDS and scratch writes
// of one lane are visible to the others: a single wavefront executes its DS / VMEM
// instructions in order, so only the compiler has to be kept from reordering.
PDEV void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
PDEV void team_sync() { __builtin_amdgcn_s_barrier(); } // no data is exchanged: pace only
// nonzero in any lane (wave-uniform result); the host emulation ORs lane after lane instead
PDEV u32 wave_any(u32 v) { return __builtin_amdgcn_ballot_w64(v != 0) != 0; }
#endif

PDEV int chain_depth(u32 pad) { return (int)(pad & 7); }
PDEV int chain_kind(u32 pad, int k) { return k == 1 ? (int)((pad >> 3) & 3) : ((pad >> (4 + k)) & 1 ? 2 : 0); }
PDEV bool chain_skip(u32 pad, int k) { return (pad >> (12 + k)) & 1; }

PDEV u32 bfi(u32 mask, u32 a, u32 b) { return ((a ^ b) & mask) ^ b; } // mask ? a : b (v_bfi_b32)

// ---- packed int8 arithmetic (PolarHelper<SIMD<int8_t,W>>, polar_helper.hh:113-173) ----

static const u32 HI_MASK = 0xFF00FF00u;
static const u32 SIGN_BITS = 0x80808080u;
PDEV u32 split_hi(u32 a) { return a & HI_MASK; }                 // bytes 1,3 stay in the high bytes
PDEV u32 split_lo(u32 a) { return perm(a, a, 0x020C000Cu); }     // bytes 0,2 -> high bytes, zero fill
PDEV u32 merge(u32 hi, u32 lo) { return perm(hi, lo, 0x07030501u); } // high bytes -> b3=hi.3 b2=lo.3 b1=hi.1 b0=lo.1

// prod (polar_helper.hh:153-160) on one 2xi16 register: sgn(x)sgn(y)min(|x|,|y|) = max(min(x,y), -max(x,y))
// (the saturating negation is monotone non-increasing, so min(-x,-y) = -max(x,y) holds for -128 too, which
// it maps to +127 like vqabs): four packed operations.
PDEV u32 prod_half(u32 x, u32 y)
{
	return pk_max(pk_min(x, y), pk_sub_sat(0, pk_max(x, y)));
}
PDEV u32 prod4(u32 a, u32 b)
{
#ifdef POLAR_FAKE_ARITH // diagnostic: the kernel's memory traffic without its arithmetic (wrong results, no erasures)
	return a | (b & 0x01010101u);
#endif
	return merge(prod_half(split_hi(a), split_hi(b)), prod_half(split_lo(a), split_lo(b)));
}
// qadd (polar_helper.hh:141-144)
PDEV u32 qadd4(u32 a, u32 b)
{
#ifdef POLAR_FAKE_ARITH
	return a | (b & 0x01010101u) | 0x01010101u;
#endif
	return merge(pk_add_sat(split_hi(a), split_hi(b)), pk_add_sat(split_lo(a), split_lo(b)));
}
// 16-bit lane masks from sign bits: m_hi from bits 15/31, m_lo from bits 7/23
PDEV u32 mask_hi(u32 s) { return perm(s, s, 0x09090808u); }
PDEV u32 mask_lo(u32 s) { u32 t = s << 8; return perm(t, t, 0x09090808u); }
// madd (polar_helper.hh:161-168): vqadd(vsign(vmax(a,-127), h), b); hs / hz carry the sign and
// zero bits of h on the byte sign bits (other bits are ignored).
PDEV u32 madd4(u32 hs, u32 hz, u32 a, u32 b)
{
#ifdef POLAR_FAKE_ARITH
	return b | (a & 0x01010101u) | ((hs & hz) & 0x02020202u) | 0x01010101u;
#endif
	const u32 K = 0x81008100u; // -127 * 256 per half
	u32 ah = pk_max(split_hi(a), K), al = pk_max(split_lo(a), K);
	u32 bh = split_hi(b), bl = split_lo(b);
	u32 rh = bfi(mask_hi(hs), pk_sub_sat(bh, ah), pk_add_sat(bh, ah));
	u32 rl = bfi(mask_lo(hs), pk_sub_sat(bl, al), pk_add_sat(bl, al));
	rh = bfi(mask_hi(hz), bh, rh);
	rl = bfi(mask_lo(hz), bl, rl);
	return merge(rh, rl);
}
// madd with a partial sum that is known to be +1 / -1 (no erasure): vqadd(vsign(vmax(a,-127), h), b)
PDEV u32 madd4_nz(u32 hs, u32 a, u32 b)
{
#ifdef POLAR_FAKE_ARITH
	return b | (a & 0x01010101u) | (hs & 0x02020202u) | 0x01010101u;
#endif
	const u32 K = 0x81008100u;
	u32 ah = pk_max(split_hi(a), K), al = pk_max(split_lo(a), K);
	u32 bh = split_hi(b), bl = split_lo(b);
	u32 rh = bfi(mask_hi(hs), pk_sub_sat(bh, ah), pk_add_sat(bh, ah));
	u32 rl = bfi(mask_lo(hs), pk_sub_sat(bl, al), pk_add_sat(bl, al));
	return merge(rh, rl);
}
// bit 7 of every byte that equals zero
PDEV u32 zero_bytes(u32 x)
{
	u32 t = (x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu;
	return ~(t | x) & SIGN_BITS;
}
// byte masks 0xFF / 0x00 from bit 7 of every byte
PDEV u32 byte_mask(u32 s) { return ((s & SIGN_BITS) >> 7) * 0xFFu; }
// message symbols (+1 / -1 / 0 per byte) from sign and zero bits on the byte sign bits
PDEV u32 symbols(u32 s, u32 z) { return (byte_mask(s) | 0x01010101u) & ~byte_mask(z); }
// |x| of the scaled halves, low bytes cleared (vqabs)
PDEV u32 abs_half(u32 x) { return pk_max(x, pk_sub_sat(0, x)) & HI_MASK; }
// bits 15/31 where the (low-byte-clean) halves are equal
PDEV u32 eq_half(u32 x, u32 y)
{
	u32 d = x ^ y;
	u32 t = (d & 0x7FFF7FFFu) + 0x7FFF7FFFu;
	return ~(t | d) & 0x80008000u;
}
// byte sign bits of the positions whose magnitude equals the weakest one (wh / wl: scaled halves)
PDEV u32 tie_bits(u32 x, u32 wh, u32 wl)
{
	return eq_half(abs_half(split_hi(x)), wh) | (eq_half(abs_half(split_lo(x)), wl) >> 8);
}
// byte-replicated mask of the `len` (< 8) elements starting at octet offset j0
PDEV u32 span_mask(int j0, int len)
{
	u32 b = (0xFFu >> j0) & ~(0xFFu >> (j0 + len));
	return b * 0x01010101u;
}
// stages h = 1, 2, 4 of the transform (trans, polar_decoder.hh:15-27) inside one octet of planes
PDEV void trans_octet(u32 &s, u32 &z, int len)
{
	if (len > 1) { s ^= (s << 1) & 0xAAAAAAAAu; z |= (z << 1) & 0xAAAAAAAAu; }
	if (len > 2) { s ^= (s << 2) & 0xCCCCCCCCu; z |= (z << 2) & 0xCCCCCCCCu; }
	if (len > 4) { s ^= (s << 4) & 0xF0F0F0F0u; z |= (z << 4) & 0xF0F0F0F0u; }
}
// parity (bit 0 of every byte -> all eight bits) of the element bits of one octet dword
PDEV u32 odd_mask(u32 px)
{
	px ^= (px >> 4) & 0x0F0F0F0Fu;
	px ^= (px >> 2) & 0x03030303u;
	px ^= (px >> 1) & 0x01010101u;
	return (px & 0x01010101u) * 0xFFu;
}

// component-wise forms for 16-byte accesses
PDEV u32x4 prod4(u32x4 a, u32x4 b) { u32x4 r; r.x = prod4(a.x, b.x); r.y = prod4(a.y, b.y); r.z = prod4(a.z, b.z); r.w = prod4(a.w, b.w); return r; }
PDEV u32x4 qadd4(u32x4 a, u32x4 b) { u32x4 r; r.x = qadd4(a.x, b.x); r.y = qadd4(a.y, b.y); r.z = qadd4(a.z, b.z); r.w = qadd4(a.w, b.w); return r; }

// ---- memory views ----

// Scratch vectors and input rows are streamed: written once, read once or twice much later by
// the same wavefront, far beyond what L2 or the Infinity Cache can hold in between. Marking the
// 16-byte accesses non-temporal keeps them from evicting lines that do get reused (+8 % at
// N = 2^16). POLAR_NO_NT restores default-policy accesses for comparison.
#ifndef POLAR_EMU
typedef __attribute__((address_space(1))) u32 glb_u32;
typedef __attribute__((address_space(1))) u32x4 glb_u32x4;
#else
typedef u32 glb_u32;
typedef u32x4 glb_u32x4;
#endif
// Keeps the optimiser from folding "hot ? default-policy access : non-temporal access" into ONE
// access of a selected address, which silently loses the non-temporal hint (the (128, 6) kernel went
// from 197 non-temporal 16-byte stores to 12 that way, and the streamed levels then displaced the hot
// ones from the cache): an empty volatile asm before and after the default-policy side makes that
// block unspeculatable and unlike its twin. It emits no instruction.
#ifndef POLAR_EMU
#define POLAR_KEEP_APART() asm volatile("")
#else
#define POLAR_KEEP_APART() ((void)0)
#endif
struct GVec { // linear dword array in the HBM scratch arena
	glb_u32 *p;
	// hot: a level just above the LDS levels. Its vector is read back a few tens of microseconds after
	// it was written and rewritten in place soon after: with the default cache policy it lives in the
	// Infinity Cache (and never reaches HBM), as long as the streamed levels above it stay non-temporal
	// and so do not displace it (tools/ubench/sc_skeleton.hip). Wave-uniform.
	bool hot;
	PDEV u32 ld(int t) const { return p[t]; }
	PDEV void st(int t, u32 v) const { p[t] = v; }
	PDEV u32x4 ld4h(int t) const { return *reinterpret_cast<const glb_u32x4 *>(p + t); }
	PDEV void st4h(int t, u32x4 v) const { *reinterpret_cast<glb_u32x4 *>(p + t) = v; }
#if !defined(POLAR_EMU) && !defined(POLAR_NO_NT)
	PDEV u32x4 ld4c(int t) const { return __builtin_nontemporal_load(reinterpret_cast<const glb_u32x4 *>(p + t)); }
	PDEV void st4c(int t, u32x4 v) const { __builtin_nontemporal_store(v, reinterpret_cast<glb_u32x4 *>(p + t)); }
#else
	PDEV u32x4 ld4c(int t) const { return ld4h(t); }
	PDEV void st4c(int t, u32x4 v) const { st4h(t, v); }
#endif
	PDEV u32x4 ld4(int t) const
	{
		u32x4 r;
		if (hot) {
			POLAR_KEEP_APART();
			r = ld4h(t);
			POLAR_KEEP_APART();
		} else {
			r = ld4c(t);
		}
		return r;
	}
	PDEV void st4(int t, u32x4 v) const
	{
		if (hot) {
			POLAR_KEEP_APART();
			st4h(t, v);
			POLAR_KEEP_APART();
		} else {
			st4c(t, v);
		}
	}
};

struct LVec { // linear dword array in LDS
	lds_u32 *p;
	PDEV u32 ld(int t) const { return p[t]; }
	PDEV void st(int t, u32 v) const { p[t] = v; }
	PDEV u32x4 ld4(int t) const { return *reinterpret_cast<const lds_u32x4 *>(p + t); }
	PDEV void st4(int t, u32x4 v) const { *reinterpret_cast<lds_u32x4 *>(p + t) = v; }
};

// the caller's codeword buffer: row i of the group starts at p + i*stride (bytes). The host
// guarantees dword-aligned rows and whole groups (ragged batches are staged through a padded
// copy, polar_hip.hip), so a row segment of the group is Q aligned dwords.
// NT = false (team kernels): default-policy loads, so that a line fetched for one wavefront of a
// team still serves the neighbouring groups' segments of it.
template <int Q, bool NT = true> struct InVec {
	const uint8_t *p; // first byte of the group's frames in row 0
	size_t stride;    // bytes per row
	PDEV u32 ld(int t) const
	{
		int i = t / Q, q = t % Q;
		return *reinterpret_cast<const u32 *>(p + (size_t)i * stride + 4 * q);
	}
	PDEV u32x4 ld4(int t) const
	{
		if (Q % 4 == 0) { // four consecutive quads of one row: rows are 16-byte aligned then
			int i = t / Q, q = t % Q;
#if !defined(POLAR_EMU) && !defined(POLAR_NO_NT)
			if (NT)
				return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p + (size_t)i * stride + 4 * q));
#endif
			return *reinterpret_cast<const u32x4 *>(p + (size_t)i * stride + 4 * q);
		}
		u32x4 v;
		v.x = ld(t); v.y = ld(t + 1); v.z = ld(t + 2); v.w = ld(t + 3);
		return v;
	}
};

template <int Q> struct OutRows { // the caller's message buffer, same row layout
	uint8_t *p;
	size_t stride;
	// message rows are written once and never read by the kernel: non-temporal where a group's row
	// segment is a whole 128-byte line, so that they do not displace the hot scratch levels from the
	// caches. Narrower groups leave them to L2, which merges the neighbouring groups' parts of a line
	// (non-temporal partial lines: (64, 7) teams 124 -> 155 ms, measured).
	static const bool NT = Q % 32 == 0;
	PDEV void st(u32 row, int q, u32 v) const
	{
#if !defined(POLAR_EMU) && !defined(POLAR_NO_NT)
		if (NT) {
			__builtin_nontemporal_store(v, reinterpret_cast<u32 *>(p + (size_t)row * stride + 4 * q));
			return;
		}
#endif
		*reinterpret_cast<u32 *>(p + (size_t)row * stride + 4 * q) = v;
	}
	PDEV void st4(u32 row, int q, u32x4 v) const // q % 4 == 0, Q % 4 == 0
	{
#if !defined(POLAR_EMU) && !defined(POLAR_NO_NT)
		if (NT) {
			__builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p + (size_t)row * stride + 4 * q));
			return;
		}
#endif
		*reinterpret_cast<u32x4 *>(p + (size_t)row * stride + 4 * q) = v;
	}
};

// 16-byte accesses through typed pointers (plane arrays, scratch regions): 16-byte aligned
PDEV u32x4 ldp4(const u32 *p) { return *reinterpret_cast<const u32x4 *>(p); }
PDEV void stp4(u32 *p, u32x4 v) { *reinterpret_cast<u32x4 *>(p) = v; }
#ifndef POLAR_EMU
PDEV u32x4 ldp4(const lds_u32 *p) { return *reinterpret_cast<const lds_u32x4 *>(p); }
PDEV void stp4(lds_u32 *p, u32x4 v) { *reinterpret_cast<lds_u32x4 *>(p) = v; }
#endif
// streaming (non-temporal) forms for the long plane arrays in scratch; plain for LDS
#if !defined(POLAR_EMU) && !defined(POLAR_NO_NT)
PDEV u32x4 ldp4s(const u32 *p) { return __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)); }
PDEV void stp4s(u32 *p, u32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p)); }
PDEV u32x4 ldp4s(const lds_u32 *p) { return ldp4(p); }
PDEV void stp4s(lds_u32 *p, u32x4 v) { stp4(p, v); }
#else
template <class T> PDEV u32x4 ldp4s(const T *p) { return ldp4(p); }
template <class T> PDEV void stp4s(T *p, u32x4 v) { stp4(p, v); }
#endif

// 16-byte plane accesses with the policy chosen at run time (hot: default, else streaming)
template <class T> PDEV u32x4 ldp4p(const T *p, bool hot)
{
	u32x4 r;
	if (hot) {
		POLAR_KEEP_APART();
		r = ldp4(p);
		POLAR_KEEP_APART();
	} else {
		r = ldp4s(p);
	}
	return r;
}
template <class T> PDEV void stp4p(T *p, u32x4 v, bool hot)
{
	if (hot) {
		POLAR_KEEP_APART();
		stp4(p, v);
		POLAR_KEEP_APART();
	} else {
		stp4s(p, v);
	}
}

template <class A> struct SrcIsLds { static const bool value = false; };
template <> struct SrcIsLds<LVec> { static const bool value = true; };

template <class T> struct Planes { // partial-sum bit planes of one node
	T *s, *z;
	u32 eb; // position of the node's first element inside the planes
};

// d[t] = fn(t, a[t], a[t + n]) for t < n: the shape of left / right / rate0_right and of the
// halving steps of rep (polar_decoder.hh:28-51, 99-101)
template <class A, class D, class Fn>
PDEV void ew_pairs(const A &a, const D &d, int n, Fn fn)
{
	if (n >= 256) {
		PFOR_LANES(lane) {
			for (int t = lane * 4; t < n; t += 256) {
				u32x4 x = a.ld4(t), y = a.ld4(t + n), r;
				r.x = fn(t, x.x, y.x);
				r.y = fn(t + 1, x.y, y.y);
				r.z = fn(t + 2, x.z, y.z);
				r.w = fn(t + 3, x.w, y.w);
				d.st4(t, r);
			}
		}
	} else {
		PFOR_LANES(lane) {
			for (int t = lane; t < n; t += 64)
				d.st(t, fn(t, a.ld(t), a.ld(t + n)));
		}
	}
	wave_sync();
}

// One group's decoder state.
// TEAM: four wavefronts of one workgroup decode four neighbouring groups side by side (each with
// its own LDS slice and arena) and meet at a barrier before every pass over the input rows, so
// that they read the 128-byte lines their row segments share at the same time.
template <int Q, int RL, bool TEAM = false>
struct Group {
	typedef InVec<Q, !TEAM> In;
	static_assert(RL >= 4, "nodes that share a partial-sum octet must live in LDS");
	static const int W = 1 << (RL + 1); // elements of the LDS partial-sum window
	static const int LDS_SOFT = Q << (RL + 1), LDS_PLANE = (W / 8) * Q;
	static const int LDS_DWORDS = LDS_SOFT + 2 * LDS_PLANE; // (128, 6): exactly 20 KiB, 8 wavefronts per CU
	// deepest fused descent: 16 x 16-byte loads per lane need more than the 128 VGPRs that the
	// variants with 16 wavefronts per CU have (99 spilled registers, 12 % slower when tried)
	static const int CHAIN_DEPTH = LDS_DWORDS * 4 > 10240 ? 4 : 3;
	// Wavefronts per SIMD the kernel is built for (its LDS share allows no more): 4 with ~10 KiB, 2 with
	// ~20 KiB, 1 beyond. Stated outright (__launch_bounds__) the 20 KiB class gets a hard 256-register
	// budget, which the pipelined chain passes (two load buffers of 64 registers) need to stay at two.
	static const int WAVES_PER_SIMD = LDS_DWORDS * 4 <= 10240 ? 4 : LDS_DWORDS * 4 <= 20480 ? 2 : 1;
#ifndef POLAR_PIPELINE
#define POLAR_PIPELINE 1
#endif
	// chain passes over scratch / input rows issue the next block's loads before they fold and store the
	// current one (chain_pipelined): +1.5 % at 2^20 frames of N = 2^16 (profiles/r3_experiment_pipelined_chains.txt).
	// Single-wavefront groups of 64 frames and more in the 20 KiB class only: the team kernels have no registers
	// to spare, and narrower groups have few passes that long.
	static const bool PIPELINE = POLAR_PIPELINE && !TEAM && WAVES_PER_SIMD == 2 && Q >= 16;
	lds_u32 *sl;        // LDS soft arena, LDS_SOFT dwords (level l at Q << l)
	lds_u32 *hls, *hlz; // LDS partial-sum planes (sign, zero), LDS_PLANE dwords each, index by position mod W
	u32 *sg;            // scratch soft arena, Q << top dwords
	u32 *sh;            // hot levels (RL+1 .. hot_top): level l at sh + (Q << l), inside this wavefront's block of the compact hot region
	u32 *hgs, *hgz;     // scratch partial-sum planes, (Q << top) / 8 dwords each
	In in;
	OutRows<Q> out;
	int top;
	// Zero planes in scratch are only written for nodes that hold an erasure (decision 0), which
	// is rare: bit l of zc = the last completed level-l node (l > RL) has a valid, non-empty zero
	// plane; zl = the same for the completed left child waiting at level l. Clear bit: the
	// plane words of that node are undefined and read as "no erasure". Wave-uniform.
	u32 zc, zl;
	int hot_top; // scratch levels RL+1 .. hot_top keep the default cache policy (GVec::hot), the ones above stream
#ifdef POLAR_PROFILE
	unsigned long long *prof; // [16 node kinds][32 levels][cycles, calls], diagnostic build only
#endif

	// level-lvl LLR region in scratch. The hot levels of all resident wavefronts lie side by side in
	// one compact region: spread over the per-wave arenas (10 MiB apart at N = 2^16) they fall into a
	// fraction of the Infinity Cache's sets and evict one another (tools/ubench/sc_skeleton.hip:
	// 5.5 TB/s spread, 6.6 TB/s compact for the same accesses).
	// Not in the team kernels: they sit at the 256-register limit and the second code path spills.
	static const bool HOT = (!TEAM || POLAR_TEAM_HOT) && POLAR_HOT_ENABLE;
	PDEV bool is_hot(int lvl) const { return HOT && lvl <= hot_top; }
	PDEV u32 *gregion(int lvl) const { return (is_hot(lvl) ? sh : sg) + (Q << lvl); }
	PDEV GVec gvec(int lvl) const { return GVec{(glb_u32 *)gregion(lvl), is_hot(lvl)}; }
	// plane dword of element e for frame quad q
	static PDEV int pidx(u32 e, int q) { return (int)(e >> 3) * Q + q; }

	PDEV Planes<lds_u32> lplanes(u32 hrd) const { return Planes<lds_u32>{hls, hlz, hrd & (u32)(W - 1)}; }
	PDEV Planes<u32> gplanes(u32 hrd) const { return Planes<u32>{hgs, hgz, hrd}; }
	// flat-pointer view of either, for the rare nodes above the LDS levels
	PDEV Planes<u32> fplanes(int node_level, u32 hrd) const
	{
		if (node_level <= RL)
			return Planes<u32>{(u32 *)hls, (u32 *)hlz, hrd & (u32)(W - 1)};
		return gplanes(hrd);
	}
	PDEV u32 *fregion(int lvl) const // flat pointer to the level-lvl LLR region (lvl < top)
	{
		if (lvl > RL)
			return gregion(lvl);
		return (u32 *)sl + (Q << lvl);
	}

	// LLR vector of a node above the LDS levels (or the input), runtime placement
	struct AnySrc {
		const Group *g;
		const u32 *base; // null: the caller's input
		PDEV u32 ld(int t) const { return base ? base[t] : g->in.ld(t); }
		PDEV u32x4 ld4(int t) const { return base ? ldp4(base + t) : g->in.ld4(t); }
	};
	PDEV AnySrc any_src(int lvl) const { return AnySrc{this, lvl == top ? nullptr : fregion(lvl)}; }

	// ---- f / g / qadd passes: polar_decoder.hh:28-51 ----
	template <int KIND, class A, class D, class H>
	PDEV void pass(const A &a, const D &d, int n, const H *hs, const H *hz, u32 eb)
	{
		if (KIND == 0)
			ew_pairs(a, d, n, [](int, u32 x, u32 y) { return prod4(x, y); });
		else if (KIND == 2)
			ew_pairs(a, d, n, [](int, u32 x, u32 y) { return qadd4(x, y); });
		else
			ew_pairs(a, d, n, [=](int t, u32 x, u32 y) {
				u32 e = eb + (u32)(t / Q);
				int i = pidx(e, t % Q), sh = (int)(e & 7);
				return madd4(hs[i] << sh, hz[i] << sh, x, y);
			});
	}
	// dispatch on where level `lvl` (source) and lvl-1 (destination) live; KIND 0 = left,
	// 1 = right (hrd = position of the decided left child), 2 = rate0_right / rep halving
	template <int KIND>
	PDEV void soft_pass(int lvl, u32 hrd)
	{
		int n = Q << (lvl - 1);
		if (lvl - 1 <= RL) {
			u32 eb = hrd & (u32)(W - 1);
			LVec d{sl + n};
			if (lvl == top)
				pass<KIND>(in, d, n, hls, hlz, eb);
			else if (lvl > RL)
				pass<KIND>(gvec(lvl), d, n, hls, hlz, eb);
			else
				pass<KIND>(LVec{sl + 2 * n}, d, n, hls, hlz, eb);
		} else {
			GVec d = gvec(lvl - 1);
			if (lvl == top)
				pass<KIND>(in, d, n, hgs, hgz, hrd);
			else
				pass<KIND>(gvec(lvl), d, n, hgs, hgz, hrd);
		}
	}


	// ---- fused descent: polar_decoder.hh:28-51 applied D levels deep in one pass ----
	// Source = level lvl (2^D * nd dwords, nd = Q << (lvl - D)); the lane that owns final index t
	// loads the 2^D source words t + j*nd, folds them stage by stage in registers and stores the
	// level lvl-k words t + j*nd, j < 2^(D-k), after stage k: every level is written once and
	// the re-read by the next `left` of the unfused program never happens.
	// HZ: the zero plane of the decided child is valid and may hold erasures
	template <bool HZ, class H>
	PDEV u32 madd_at(const H *hs, const H *hz, u32 eb, int t, u32 x, u32 y) const
	{
		u32 e = eb + (u32)(t / Q);
		int i = pidx(e, t % Q), sh = (int)(e & 7);
		if (!HZ)
			return madd4_nz(hs[i] << sh, x, y);
		return madd4(hs[i] << sh, hz[i] << sh, x, y);
	}
	template <bool HZ, class H>
	PDEV u32x4 madd_at(const H *hs, const H *hz, u32 eb, int t, u32x4 x, u32x4 y) const
	{
		u32x4 r;
		r.x = madd_at<HZ>(hs, hz, eb, t, x.x, y.x);
		r.y = madd_at<HZ>(hs, hz, eb, t + 1, x.y, y.y);
		r.z = madd_at<HZ>(hs, hz, eb, t + 2, x.z, y.z);
		r.w = madd_at<HZ>(hs, hz, eb, t + 3, x.w, y.w);
		return r;
	}
	static PDEV u32 ldE(const LVec &v, int t, u32) { return v.ld(t); }
	static PDEV u32x4 ldE(const LVec &v, int t, u32x4) { return v.ld4(t); }
	static PDEV u32 ldE(const GVec &v, int t, u32) { return v.ld(t); }
	static PDEV u32x4 ldE(const GVec &v, int t, u32x4) { return v.ld4(t); }
	static PDEV u32 ldE(const In &v, int t, u32) { return v.ld(t); }
	static PDEV u32x4 ldE(const In &v, int t, u32x4) { return v.ld4(t); }
	static PDEV u32x4 ldE(const AnySrc &v, int t, u32x4) { return v.ld4(t); }
	// the same split by cache policy for the scratch vectors (hot: default, cold: non-temporal)
	template <class A> static PDEV bool src_hot(const A &) { return false; }
	static PDEV bool src_hot(const GVec &v) { return v.hot; }
	template <class A, class E> static PDEV E ldH(const A &v, int t, E e) { return ldE(v, t, e); }
	template <class A, class E> static PDEV E ldC(const A &v, int t, E e) { return ldE(v, t, e); }
	static PDEV u32x4 ldH(const GVec &v, int t, u32x4) { return v.ld4h(t); }
	static PDEV u32x4 ldC(const GVec &v, int t, u32x4) { return v.ld4c(t); }
	PDEV void stE(int level, int t, u32 v) const
	{
		if (level > RL)
			gregion(level)[t] = v;
		else
			sl[(Q << level) + t] = v;
	}
	PDEV void stE(int level, int t, u32x4 v) const
	{
		if (level > RL)
			gvec(level).st4(t, v);
		else
			LVec{sl + (Q << level)}.st4(t, v);
	}
	// U independent final indices t, t + step, ... per lane: all 2^D * U loads are issued before
	// the first use, which is what keeps enough bytes in flight for the HBM-resident levels
	template <int D, class E, int U, class A>
	PDEV void chain_load(E (&x)[U][1 << D], const A &a, int t, int nd, int step)
	{
		if (src_hot(a)) { // one branch for the whole block of loads, not one per load
			POLAR_KEEP_APART();
#pragma unroll
			for (int u = 0; u < U; ++u)
#pragma unroll
				for (int j = 0; j < (1 << D); ++j)
					x[u][j] = ldH(a, t + u * step + j * nd, E());
			POLAR_KEEP_APART();
		} else {
#pragma unroll
			for (int u = 0; u < U; ++u)
#pragma unroll
				for (int j = 0; j < (1 << D); ++j)
					x[u][j] = ldC(a, t + u * step + j * nd, E());
		}
	}
	template <int D, bool HZ, class E, int U, class H>
	PDEV void chain_fold(E (&x)[U][1 << D], int lvl, u32 pad, const H *hs, const H *hz, u32 eb, int t, int nd, int step)
	{
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int tu = t + u * step;
#pragma unroll
			for (int k = 1; k <= D; ++k) {
				const int half = 1 << (D - k);
				int kind = chain_kind(pad, k);
				if (kind == 0) {
#pragma unroll
					for (int j = 0; j < half; ++j)
						x[u][j] = prod4(x[u][j], x[u][j + half]);
				} else if (kind == 2) {
#pragma unroll
					for (int j = 0; j < half; ++j)
						x[u][j] = qadd4(x[u][j], x[u][j + half]);
				} else {
#pragma unroll
					for (int j = 0; j < half; ++j)
						x[u][j] = madd_at<HZ>(hs, hz, eb, tu + j * nd, x[u][j], x[u][j + half]);
				}
				if (!chain_skip(pad, k)) {
#pragma unroll
					for (int j = 0; j < half; ++j)
						stE(lvl - k, tu + j * nd, x[u][j]);
				}
			}
		}
	}
	// The same pass with the loads of the next block of final indices issued before the current
	// block is folded and stored (two register buffers): a wavefront's memory stream does not
	// pause for its arithmetic. S = final indices per block, nd a multiple of S (both powers of
	// two times Q), so the trip count is wave-uniform.
	template <int D, bool HZ, class E, int U, class A, class H>
	PDEV void chain_pipelined(const A &a, int lvl, u32 pad, const H *hs, const H *hz, u32 eb, int t, int nd, int step, int S)
	{
		E xa[U][1 << D], xb[U][1 << D];
		chain_load<D, E, U>(xa, a, t, nd, step);
		for (;;) {
			const bool more1 = t + S < nd;
			if (more1)
				chain_load<D, E, U>(xb, a, t + S, nd, step);
			chain_fold<D, HZ, E, U>(xa, lvl, pad, hs, hz, eb, t, nd, step);
			if (!more1)
				break;
			t += S;
			const bool more2 = t + S < nd;
			if (more2)
				chain_load<D, E, U>(xa, a, t + S, nd, step);
			chain_fold<D, HZ, E, U>(xb, lvl, pad, hs, hz, eb, t, nd, step);
			if (!more2)
				break;
			t += S;
		}
	}
	template <int D, bool HZ, class E, int U, class A, class H>
	PDEV void chain_lane(const A &a, int lvl, u32 pad, const H *hs, const H *hz, u32 eb, int t, int nd, int step)
	{
		E x[U][1 << D];
		if (src_hot(a)) { // one branch for the whole block of loads, not one per load
			POLAR_KEEP_APART();
#pragma unroll
			for (int u = 0; u < U; ++u)
#pragma unroll
				for (int j = 0; j < (1 << D); ++j)
					x[u][j] = ldH(a, t + u * step + j * nd, E());
			POLAR_KEEP_APART();
		} else {
#pragma unroll
			for (int u = 0; u < U; ++u)
#pragma unroll
				for (int j = 0; j < (1 << D); ++j)
					x[u][j] = ldC(a, t + u * step + j * nd, E());
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			const int tu = t + u * step;
#pragma unroll
			for (int k = 1; k <= D; ++k) {
				const int half = 1 << (D - k);
				int kind = chain_kind(pad, k);
				if (kind == 0) {
#pragma unroll
					for (int j = 0; j < half; ++j)
						x[u][j] = prod4(x[u][j], x[u][j + half]);
				} else if (kind == 2) {
#pragma unroll
					for (int j = 0; j < half; ++j)
						x[u][j] = qadd4(x[u][j], x[u][j + half]);
				} else {
#pragma unroll
					for (int j = 0; j < half; ++j)
						x[u][j] = madd_at<HZ>(hs, hz, eb, tu + j * nd, x[u][j], x[u][j + half]);
				}
				if (!chain_skip(pad, k)) {
#pragma unroll
					for (int j = 0; j < half; ++j)
						stE(lvl - k, tu + j * nd, x[u][j]);
				}
			}
		}
	}
	template <int D, bool HZ, class A, class H>
	PDEV void chain_run(const A &a, int lvl, u32 pad, const H *hs, const H *hz, u32 eb)
	{
		int nd = Q << (lvl - D);
		// loads in flight per lane from HBM: 16 x 16 bytes
#ifndef POLAR_INFLIGHT
#define POLAR_INFLIGHT 16
#endif
		const int U = SrcIsLds<A>::value ? 1 : (POLAR_INFLIGHT >> D) > 0 ? (POLAR_INFLIGHT >> D) : 1;
		// from two blocks per lane on, scratch / input sources only
		if (PIPELINE && !SrcIsLds<A>::value && Q % 4 == 0 && nd >= 256 * U * 2) {
			PFOR_LANES(lane) {
				chain_pipelined<D, HZ, u32x4, U>(a, lvl, pad, hs, hz, eb, lane * 4, nd, 256, 256 * U);
			}
			wave_sync();
			return;
		}
		if (U > 1 && nd >= 256 * U) {
			PFOR_LANES(lane) {
				for (int t = lane * 4; t < nd; t += 256 * U)
					chain_lane<D, HZ, u32x4, U>(a, lvl, pad, hs, hz, eb, t, nd, 256);
			}
		} else if (nd >= 256) {
			PFOR_LANES(lane) {
				for (int t = lane * 4; t < nd; t += 256)
					chain_lane<D, HZ, u32x4, 1>(a, lvl, pad, hs, hz, eb, t, nd, 0);
			}
		} else {
			PFOR_LANES(lane) {
				for (int t = lane; t < nd; t += 64)
					chain_lane<D, HZ, u32, 1>(a, lvl, pad, hs, hz, eb, t, nd, 0);
			}
		}
		wave_sync();
	}
	template <int D>
	PDEV void chain_at(int lvl, u32 hrd, u32 pad)
	{
		// the decided left child of a leading `right` is a level lvl-1 node
		if (lvl - 1 <= RL) {
			u32 eb = hrd & (u32)(W - 1);
			if (lvl == top)
				chain_run<D, true>(in, lvl, pad, hls, hlz, eb);
			else if (lvl > RL)
				chain_run<D, true>(gvec(lvl), lvl, pad, hls, hlz, eb);
			else
				chain_run<D, true>(LVec{sl + (Q << lvl)}, lvl, pad, hls, hlz, eb);
		} else { // scratch planes: only reached when the child's zero plane is empty (chain())
			if (lvl == top)
				chain_run<D, false>(in, lvl, pad, hgs, hgz, hrd);
			else
				chain_run<D, false>(gvec(lvl), lvl, pad, hgs, hgz, hrd);
		}
	}
	PDEV void chain(int lvl, u32 hrd, u32 pad)
	{
		int depth = chain_depth(pad);
		if (chain_kind(pad, 1) == 1) { // leading `right`: its left sibling (level lvl-1) is decided
			u32 bit = 1u << (lvl - 1);
			zl = (zl & ~bit) | (zc & bit);
			if (lvl - 1 > RL && (zc & bit)) {
				// rare: the sibling holds erasures; run the stages one by one with the zero plane
				soft_pass<1>(lvl, hrd);
				for (int k = 2; k <= depth; ++k) {
					if (chain_kind(pad, k) == 0)
						soft_pass<0>(lvl - k + 1, 0);
					else
						soft_pass<2>(lvl - k + 1, 0);
				}
				return;
			}
		}
		switch (depth) {
		case 1: chain_at<1>(lvl, hrd, pad); break;
		case 2: chain_at<2>(lvl, hrd, pad); break;
		case 3: chain_at<3>(lvl, hrd, pad); break;
		default: chain_at<(CHAIN_DEPTH >= 4 ? 4 : 3)>(lvl, hrd, pad); break;
		}
	}

	// comb / rate0_comb (polar_decoder.hh:52-67): node of level lvl at hrd; RATE0: left half is all +1.
	// Node and children in the LDS window (lvl <= RL), in place.
	template <bool RATE0>
	PDEV void comb_lds(int lvl, u32 hrd)
	{
		Planes<lds_u32> o = lplanes(hrd);
		int h = 1 << (lvl - 1);
		if (h < 8) { // both halves share one octet
			int j0 = (int)(o.eb & 7);
			u32 lowm = span_mask(j0, h);
			PFOR_LANES(lane) {
				if (lane < Q) {
					int i = pidx(o.eb, lane);
					u32 s = o.s[i], z = o.z[i];
					if (RATE0) {
						s = bfi(lowm, s << h, s);
						z = bfi(lowm, z << h, z);
					} else {
						s ^= (s << h) & lowm;
						z |= (z << h) & lowm;
					}
					o.s[i] = s;
					o.z[i] = z;
				}
			}
		} else {
			int n = (h >> 3) * Q; // plane dwords per half
			int l0 = pidx(o.eb, 0), r0 = l0 + n;
			PFOR_LANES(lane) {
				for (int t = lane; t < n; t += 64) {
					u32 rs = o.s[r0 + t], rz = o.z[r0 + t];
					o.s[l0 + t] = RATE0 ? rs : o.s[l0 + t] ^ rs;
					o.z[l0 + t] = RATE0 ? rz : o.z[l0 + t] | rz;
				}
			}
		}
		wave_sync();
	}
	// d[t] = a[t] (op) b[t] over n plane dwords, 16-byte accesses with U of them in flight per
	// array and lane once the arrays are long enough. MODE 0: copy b, 1: a ^ b, 2: a | b,
	// 3: fill with zero. Returns "any result word nonzero" (wave-uniform).
	// hot: default cache policy for the scratch accesses (planes of a hot level), else streaming
	template <int MODE, class TD, class TA, class TB>
	PDEV u32 plane_op(TD *d, const TA *a, const TB *b, int n, bool hot = false)
	{
		u32 any = 0;
		if (Q % 4 == 0 && n >= 256) {
			const int U = 4;
			PFOR_LANES(lane) {
				u32 acc = 0;
				for (int t0 = lane * 4; t0 < n; t0 += 256 * U) {
					u32x4 x[U], y[U];
#pragma unroll
					for (int u = 0; u < U; ++u) {
						int t = t0 + 256 * u;
						if (t < n) {
							if (MODE != 3)
								y[u] = ldp4p(b + t, hot);
							if (MODE == 1 || MODE == 2)
								x[u] = ldp4p(a + t, hot);
						}
					}
#pragma unroll
					for (int u = 0; u < U; ++u) {
						int t = t0 + 256 * u;
						if (t < n) {
							u32x4 r;
							if (MODE == 3) {
								r.x = r.y = r.z = r.w = 0;
							} else if (MODE == 0) {
								r = y[u];
							} else if (MODE == 1) {
								r.x = x[u].x ^ y[u].x; r.y = x[u].y ^ y[u].y; r.z = x[u].z ^ y[u].z; r.w = x[u].w ^ y[u].w;
							} else {
								r.x = x[u].x | y[u].x; r.y = x[u].y | y[u].y; r.z = x[u].z | y[u].z; r.w = x[u].w | y[u].w;
							}
							acc |= r.x | r.y | r.z | r.w;
							stp4p(d + t, r, hot);
						}
					}
				}
				any |= wave_any(acc);
			}
		} else {
			PFOR_LANES(lane) {
				u32 acc = 0;
				for (int t = lane; t < n; t += 64) {
					u32 r = MODE == 3 ? 0u : MODE == 0 ? b[t] : MODE == 1 ? a[t] ^ b[t] : a[t] | b[t];
					acc |= r;
					d[t] = r;
				}
				any |= wave_any(acc);
			}
		}
		return any;
	}
	// any word of the n-dword LDS array nonzero (wave-uniform)
	PDEV u32 lds_any(const lds_u32 *a, int n)
	{
		u32 any = 0;
		PFOR_LANES(lane) {
			u32 acc = 0;
			for (int t = lane; t < n; t += 64)
				acc |= a[t];
			any |= wave_any(acc);
		}
		return any;
	}
	PDEV void set_zc(int lvl, u32 any) { zc = (zc & ~(1u << lvl)) | ((any ? 1u : 0u) << lvl); }
	// Node of level RL+1: the children fill the LDS window, the node's planes go to scratch.
	template <bool RATE0>
	PDEV void comb_up(int lvl, u32 hrd)
	{
		int n = ((1 << (lvl - 1)) >> 3) * Q; // plane dwords per half; the left child starts the window
		u32 *os = hgs + pidx(hrd, 0), *oz = hgz + pidx(hrd, 0);
		const bool hot = is_hot(lvl);
		if (RATE0)
			plane_op<0>(os, hls, hls + n, n, hot);
		else
			plane_op<1>(os, hls, hls + n, n, hot);
		plane_op<0>(os + n, hls, hls + n, n, hot);
		u32 any = RATE0 ? lds_any(hlz + n, n) : lds_any(hlz, 2 * n);
		if (any) {
			if (RATE0)
				plane_op<0>(oz, hlz, hlz + n, n, hot);
			else
				plane_op<2>(oz, hlz, hlz + n, n, hot);
			plane_op<0>(oz + n, hlz, hlz + n, n, hot);
		}
		set_zc(lvl, any);
		wave_sync();
	}
	// Node above level RL+1: children and node in scratch, in place; zero planes only where valid.
	template <bool RATE0>
	PDEV void comb_global(int lvl, u32 hrd)
	{
		int n = ((1 << (lvl - 1)) >> 3) * Q;
		u32 *ls = hgs + pidx(hrd, 0), *lz = hgz + pidx(hrd, 0);
		u32 fl = RATE0 ? 0u : (zl >> (lvl - 1)) & 1u, fr = (zc >> (lvl - 1)) & 1u;
		const bool hot = is_hot(lvl); // the result waits as long as the LLR vector of this level
		if (RATE0)
			plane_op<0>(ls, ls, ls + n, n, hot);
		else
			plane_op<1>(ls, ls, ls + n, n, hot);
		if (fl | fr) {
			if (!fr)
				plane_op<3>(lz + n, lz, lz, n, hot); // the right half becomes valid: no erasures there
			else if (!fl)
				plane_op<0>(lz, lz, lz + n, n, hot); // left = 0 | right
			else
				plane_op<2>(lz, lz, lz + n, n, hot);
		}
		set_zc(lvl, fl | fr);
		wave_sync();
	}
	template <bool RATE0>
	PDEV void comb_any(int lvl, u32 hrd)
	{
		if (lvl <= RL)
			comb_lds<RATE0>(lvl, hrd);
		else if (lvl == RL + 1)
			comb_up<RATE0>(lvl, hrd);
		else
			comb_global<RATE0>(lvl, hrd);
	}

	// store a node's decision bits; s / z hold element j of the item at bit 7-j (normalised)
	template <class T>
	static PDEV void put_bits(const Planes<T> &p, int i, int j0, int len, u32 s, u32 z)
	{
		if (len >= 8) {
			p.s[i] = s;
			p.z[i] = z;
		} else {
			u32 m = span_mask(j0, len);
			p.s[i] = bfi(m, s >> j0, p.s[i]);
			p.z[i] = bfi(m, z >> j0, p.z[i]);
		}
	}

	// message rows row.. for elements first..len-1 of one octet held in registers
	PDEV void emit_octet(u32 s, u32 z, int len, int first, u32 row, int q) const
	{
#pragma unroll
		for (int e = 0; e < 8; ++e)
			if (e >= first && e < len)
				out.st(row + (u32)(e - first), q, symbols(s << e, z << e));
	}

	// ---- leaves of at most 8 elements: one lane per frame quad, everything in registers ----
	// rate0 / rate1 / rep / spc, polar_decoder.hh:68-75, 85-127
	template <class SRC>
	PDEV void leaf_small(u32 op, int lvl, const SRC &src, u32 hrd, u32 row)
	{
		int len = 1 << lvl;
		Planes<lds_u32> p = lplanes(hrd);
		int j0 = (int)(p.eb & 7);
		PFOR_LANES(lane) {
			if (lane < Q) {
				int q = lane;
				u32 x[8];
#pragma unroll
				for (int j = 0; j < 8; ++j)
					x[j] = j < len && op != OP_RATE0 ? src.ld(j * Q + q) : 0;
				u32 s = 0, z = 0;
				if (op == OP_REP) {
#pragma unroll
					for (int h = 4; h >= 1; h /= 2)
						if (h < len) {
#pragma unroll
							for (int i = 0; i < 4; ++i)
								if (i < h)
									x[i] = qadd4(x[i], x[i + h]);
						}
					u32 zb = zero_bytes(x[0]);
					s = byte_mask(x[0]);
					z = byte_mask(zb);
					out.st(row, q, symbols(x[0], zb));
				} else if (op == OP_RATE1) {
#pragma unroll
					for (int j = 0; j < 8; ++j)
						if (j < len) {
							s |= (x[j] & SIGN_BITS) >> j;
							z |= zero_bytes(x[j]) >> j;
						}
				} else if (op == OP_SPC) {
					u32 mh = 0x7F007F00u, ml = 0x7F007F00u;
#pragma unroll
					for (int j = 0; j < 8; ++j)
						if (j < len) {
							s |= (x[j] & SIGN_BITS) >> j;
							mh = pk_min(mh, abs_half(split_hi(x[j])));
							ml = pk_min(ml, abs_half(split_lo(x[j])));
						}
					u32 odd = odd_mask(s);
					if (odd) { // every position tied for the minimum takes the parity (polar_helper.hh:169-172)
						u32 tie = 0;
#pragma unroll
						for (int j = 0; j < 8; ++j)
							if (j < len)
								tie |= tie_bits(x[j], mh, ml) >> j;
						s ^= tie & odd;
					}
				}
				put_bits(p, pidx(p.eb, q), j0, len, s, z);
				if (op == OP_RATE1 || op == OP_SPC) {
					trans_octet(s, z, len);
					emit_octet(s, z, len, op == OP_SPC ? 1 : 0, row, q);
				}
			}
		}
		wave_sync();
	}

	// ---- composite nodes of at most 8 elements, one lane per frame quad, everything in registers ----
	// x: the node's 2^L LLR dwords; fm: frozen mask (bit j = position j). Returns the node's decisions
	// as normalised planes (element j at bit 7-j of every byte, other bits clear) and emits its
	// message rows from `row` on. The case order is the compiler's (polar_compiler.hh:26-48), the
	// node functions are polar_decoder.hh:28-127; a left ... rate1_comb node is the same g pass
	// followed by a rate-1 right child and the comb.
	template <int L>
	PDEV void reg_subtree(u32 (&x)[1 << L], u32 fm, u32 &s, u32 &z, u32 &row, int q) const
	{
		constexpr int len = 1 << L, half = len / 2;
		constexpr u32 all = (1u << len) - 1u, lo = (1u << half) - 1u;
		fm &= all;
		const u32 l = (u32)__builtin_popcount(fm & lo), r = (u32)__builtin_popcount(fm >> half);
		s = z = 0;
		if (l == (u32)half && r == (u32)half) // rate0: all +1
			return;
		if (!l && !r) { // rate1
#pragma unroll
			for (int j = 0; j < len; ++j) {
				s |= (x[j] & SIGN_BITS) >> j;
				z |= zero_bytes(x[j]) >> j;
			}
			u32 ts = s, tz = z;
			trans_octet(ts, tz, len);
			emit_octet(ts, tz, len, 0, row, q);
			row += len;
			return;
		}
		if (l == (u32)half && r == (u32)half - 1 && !((fm >> (len - 1)) & 1u)) { // rep: pairwise halving sums
#pragma unroll
			for (int h = half; h >= 1; h /= 2)
#pragma unroll
				for (int i = 0; i < h; ++i)
					x[i] = qadd4(x[i], x[i + h]);
			const u32 zb = zero_bytes(x[0]), m = span_mask(0, len);
			s = byte_mask(x[0]) & m;
			z = byte_mask(zb) & m;
			out.st(row, q, symbols(x[0], zb));
			row += 1;
			return;
		}
		if (l == 1 && !r && (fm & 1u)) { // spc: every position tied for the minimum takes the parity
			u32 mh = 0x7F007F00u, ml = 0x7F007F00u;
#pragma unroll
			for (int j = 0; j < len; ++j) {
				s |= (x[j] & SIGN_BITS) >> j;
				mh = pk_min(mh, abs_half(split_hi(x[j])));
				ml = pk_min(ml, abs_half(split_lo(x[j])));
			}
			const u32 odd = odd_mask(s);
			if (odd) {
				u32 tie = 0;
#pragma unroll
				for (int j = 0; j < len; ++j)
					tie |= tie_bits(x[j], mh, ml) >> j;
				s ^= tie & odd;
			}
			u32 ts = s, tz = 0;
			trans_octet(ts, tz, len);
			emit_octet(ts, tz, len, 1, row, q);
			row += len - 1;
			return;
		}
		if constexpr (L > 1) {
			u32 y[half], cs, cz;
			if (l == (u32)half) { // rate0_right, child, rate0_comb
#pragma unroll
				for (int i = 0; i < half; ++i)
					y[i] = qadd4(x[i], x[i + half]);
				reg_subtree<L - 1>(y, fm >> half, cs, cz, row, q);
				s = cs | (cs >> half);
				z = cz | (cz >> half);
				return;
			}
#pragma unroll
			for (int i = 0; i < half; ++i) // left
				y[i] = prod4(x[i], x[i + half]);
			reg_subtree<L - 1>(y, fm, cs, cz, row, q);
#pragma unroll
			for (int i = 0; i < half; ++i) // right
				y[i] = madd4(cs << i, cz << i, x[i], x[i + half]);
			u32 rs, rz;
			reg_subtree<L - 1>(y, fm >> half, rs, rz, row, q);
			const u32 m = span_mask(0, half); // comb
			s = ((cs ^ rs) & m) | (rs >> half);
			z = ((cz | rz) & m) | (rz >> half);
		}
	}
	template <class SRC>
	PDEV void subtree_from(int lvl, const SRC &src, u32 hrd, u32 row, u32 fm)
	{
		Planes<lds_u32> p = lplanes(hrd);
		PFOR_LANES(lane) {
			if (lane < Q) {
				int q = lane, i = pidx(p.eb, q);
				u32 r = row;
				if (lvl == 3) {
					u32 x[8], s, z;
#pragma unroll
					for (int j = 0; j < 8; ++j)
						x[j] = src.ld(j * Q + q);
					reg_subtree<3>(x, fm, s, z, r, q);
					put_bits(p, i, 0, 8, s, z);
				} else { // level 4: a composite root over two level-3 children, one plane octet each
					u32 x[16], y[8], ls, lz, rs, rz;
#pragma unroll
					for (int j = 0; j < 16; ++j)
						x[j] = src.ld(j * Q + q);
					if (__builtin_popcount(fm & 0xFFu) == 8) { // rate0_right, child, rate0_comb
#pragma unroll
						for (int j = 0; j < 8; ++j)
							y[j] = qadd4(x[j], x[j + 8]);
						reg_subtree<3>(y, fm >> 8, rs, rz, r, q);
						ls = rs;
						lz = rz;
					} else {
#pragma unroll
						for (int j = 0; j < 8; ++j)
							y[j] = prod4(x[j], x[j + 8]);
						reg_subtree<3>(y, fm, ls, lz, r, q);
#pragma unroll
						for (int j = 0; j < 8; ++j)
							y[j] = madd4(ls << j, lz << j, x[j], x[j + 8]);
						reg_subtree<3>(y, fm >> 8, rs, rz, r, q);
						ls ^= rs;
						lz |= rz;
					}
					put_bits(p, i, 0, 8, ls, lz);
					put_bits(p, i + Q, 0, 8, rs, rz);
				}
			}
		}
		wave_sync();
	}
	PDEV void subtree(int lvl, u32 hrd, u32 row, u32 fm)
	{
		if (lvl == top)
			subtree_from(lvl, in, hrd, row, fm);
		else
			subtree_from(lvl, LVec{sl + (Q << lvl)}, hrd, row, fm);
	}

	// one or two butterfly stages (octet distance ho, and 2*ho when TWO) of the transform on one
	// plane array of noct*Q dwords; OR instead of XOR for the zero plane
	template <bool TWO, bool IS_OR, class E, class T>
	PDEV void trans_stage(T *w, int n, int ho, int lane_step)
	{
		const int per = TWO ? 4 : 2, V = sizeof(E) / 4;
		PFOR_LANES(lane) {
			for (int t = lane * V; t < n / per; t += lane_step) {
				int o = t / Q, q = t % Q;
				int i0 = ((o / ho) * per * ho + (o % ho)) * Q + q, d = ho * Q;
				E x0 = ldE_p(w + i0, E()), x1 = ldE_p(w + i0 + d, E());
				if (TWO) {
					E x2 = ldE_p(w + i0 + 2 * d, E()), x3 = ldE_p(w + i0 + 3 * d, E());
					x0 = IS_OR ? (x0 | x1) | (x2 | x3) : (x0 ^ x1) ^ (x2 ^ x3);
					x1 = IS_OR ? x1 | x3 : x1 ^ x3;
					x2 = IS_OR ? x2 | x3 : x2 ^ x3;
					stE_p(w + i0 + d, x1);
					stE_p(w + i0 + 2 * d, x2);
				} else {
					x0 = IS_OR ? x0 | x1 : x0 ^ x1;
				}
				stE_p(w + i0, x0);
			}
		}
	}
	template <class T> static PDEV u32 ldE_p(const T *p, u32) { return *p; }
	template <class T> static PDEV u32x4 ldE_p(const T *p, u32x4) { return ldp4(p); }
	template <class T> static PDEV void stE_p(T *p, u32 v) { *p = v; }
	template <class T> static PDEV void stE_p(T *p, u32x4 v) { stp4(p, v); }

	// polar transform of a node's decisions (trans, polar_decoder.hh:15-27) on bit planes held in
	// ts / tz (noct*Q dwords each, element 0 at bit 7), then emission of elements first..len-1
	// as message rows row, row+1, ... (len >= 16). hz: tz holds erasures; without it tz is
	// neither read nor written. Two stages per pass where the node is deep enough.
	template <class T>
	PDEV void trans_emit(T *ts, T *tz, int len, int first, u32 row, bool hz)
	{
		int noct = len >> 3;
		int n = noct * Q;
		for (int ho = 1; ho < noct;) { // stages h = 8*ho
			bool two = 4 * ho <= noct;
			if (two) {
				if (Q % 4 == 0 && n >= 1024) {
					trans_stage<true, false, u32x4>(ts, n, ho, 256);
					if (hz)
						trans_stage<true, true, u32x4>(tz, n, ho, 256);
				} else {
					trans_stage<true, false, u32>(ts, n, ho, 64);
					if (hz)
						trans_stage<true, true, u32>(tz, n, ho, 64);
				}
			} else {
				if (Q % 4 == 0 && n >= 512) {
					trans_stage<false, false, u32x4>(ts, n, ho, 256);
					if (hz)
						trans_stage<false, true, u32x4>(tz, n, ho, 256);
				} else {
					trans_stage<false, false, u32>(ts, n, ho, 64);
					if (hz)
						trans_stage<false, true, u32>(tz, n, ho, 64);
				}
			}
			wave_sync();
			ho *= two ? 4 : 2;
		}
		int cnt = (len - first) * Q;
		if (Q % 4 == 0) {
			PFOR_LANES(lane) {
				for (int t = lane * 4; t < cnt; t += 256) {
					int e = first + t / Q, q = t % Q;
					int i = pidx((u32)e, q), sh = e & 7;
					u32x4 a = ldp4(ts + i), b, v;
					if (hz)
						b = ldp4(tz + i);
					else
						b.x = b.y = b.z = b.w = 0;
					v.x = symbols(a.x << sh, b.x << sh);
					v.y = symbols(a.y << sh, b.y << sh);
					v.z = symbols(a.z << sh, b.z << sh);
					v.w = symbols(a.w << sh, b.w << sh);
					out.st4(row + (u32)(e - first), q, v);
				}
			}
		} else {
			PFOR_LANES(lane) {
				for (int t = lane; t < cnt; t += 64) {
					int e = first + t / Q, q = t % Q;
					int i = pidx((u32)e, q), sh = e & 7;
					out.st(row + (u32)(e - first), q, symbols(ts[i] << sh, hz ? tz[i] << sh : 0u));
				}
			}
		}
		wave_sync();
	}

	// ---- leaves of 16 elements and more: all lanes, one (octet, quad) item each ----
	// tmp: 4*(len/8)*Q dwords of scratch (the dead LLR region below the node)
	// wz: write the zero plane even when it is empty (LDS window); without it (scratch planes) an
	// empty zero plane is left unwritten. Returns "the node holds an erasure" (wave-uniform).
	template <class SRC, class T>
	PDEV u32 leaf_big(u32 op, int lvl, const SRC &src, Planes<T> p, T *tmp, u32 row, bool wz)
	{
		int len = 1 << lvl, noct = len >> 3, n = noct * Q;
		int i0 = pidx(p.eb, 0);
		u32 any = 0;
		if (op == OP_RATE0) {
			plane_op<3>(p.s + i0, p.s, p.s, n);
			if (wz)
				plane_op<3>(p.z + i0, p.z, p.z, n);
			wave_sync();
			return 0;
		}
		if (op == OP_REP) { // the halving passes have left the sum in the level-0 region
			lds_u32 *s0 = sl + Q;
			PFOR_LANES(lane) {
				u32 acc = 0;
				for (int t = lane; t < n; t += 64) {
					int q = t % Q;
					u32 x = s0[q], zb = zero_bytes(x);
					p.s[i0 + t] = byte_mask(x);
					p.z[i0 + t] = byte_mask(zb);
					acc |= zb;
					if (t < Q)
						out.st(row, q, symbols(x, zb));
				}
				any |= wave_any(acc);
			}
			wave_sync();
			return any;
		}
		T *ts = tmp, *tz = tmp + n, *mins = tmp + 2 * n;
		if (op == OP_RATE1) {
			PFOR_LANES(lane) {
				u32 acc = 0;
				for (int t = lane; t < n; t += 64) {
					int o = t / Q, q = t % Q;
					u32 s = 0, z = 0;
#pragma unroll
					for (int j = 0; j < 8; ++j) {
						u32 x = src.ld((8 * o + j) * Q + q);
						s |= (x & SIGN_BITS) >> j;
						z |= zero_bytes(x) >> j;
					}
					p.s[i0 + t] = s;
					p.z[i0 + t] = z;
					acc |= z;
					trans_octet(s, z, 8);
					ts[t] = s;
					tz[t] = z;
				}
				any |= wave_any(acc);
			}
			wave_sync();
			trans_emit(ts, tz, len, 0, row, any != 0);
			return any;
		}
		// spc (polar_decoder.hh:107-127)
		if (WIDE_LEAVES && n >= 256) {
			spc_wide(lvl, src, p, tmp, sl + Q, row, wz);
			return 0;
		}
		PFOR_LANES(lane) {
			for (int t = lane; t < n; t += 64) {
				int o = t / Q, q = t % Q;
				u32 s = 0, mh = 0x7F007F00u, ml = 0x7F007F00u;
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					u32 x = src.ld((8 * o + j) * Q + q);
					s |= (x & SIGN_BITS) >> j;
					mh = pk_min(mh, abs_half(split_hi(x)));
					ml = pk_min(ml, abs_half(split_lo(x)));
				}
				ts[t] = s;
				mins[2 * t] = mh;
				mins[2 * t + 1] = ml;
			}
		}
		wave_sync();
		lds_u32 *wk = sl + Q; // per quad: weakest magnitude (2 dwords) and odd-parity mask, in the dead levels 0-1
		PFOR_LANES(lane) {
			if (lane < Q) {
				u32 mh = 0x7F007F00u, ml = 0x7F007F00u, px = 0;
				for (int o = 0; o < noct; ++o) {
					int t = o * Q + lane;
					mh = pk_min(mh, mins[2 * t]);
					ml = pk_min(ml, mins[2 * t + 1]);
					px ^= ts[t];
				}
				wk[3 * lane] = mh;
				wk[3 * lane + 1] = ml;
				wk[3 * lane + 2] = odd_mask(px);
			}
		}
		wave_sync();
		PFOR_LANES(lane) {
			for (int t = lane; t < n; t += 64) {
				int o = t / Q, q = t % Q;
				u32 wh = wk[3 * q], wl = wk[3 * q + 1], odd = wk[3 * q + 2];
				u32 s = ts[t], z = 0;
				if (odd) {
					u32 tie = 0;
#pragma unroll
					for (int j = 0; j < 8; ++j)
						tie |= tie_bits(src.ld((8 * o + j) * Q + q), wh, wl) >> j;
					s ^= tie & odd;
				}
				p.s[i0 + t] = s;
				if (wz)
					p.z[i0 + t] = 0;
				trans_octet(s, z, 8);
				ts[t] = s;
			}
		}
		wave_sync();
		trans_emit(ts, tz, len, 1, row, false);
		return 0; // a parity-check node never decides 0 (polar_decoder.hh:107-127)
	}

	// spc (polar_decoder.hh:107-127) for nodes of >= 256 plane dwords: every lane owns four frame
	// quads of every (256/Q)-th octet, keeps the running weakest magnitude and parity in registers
	// and has its eight 16-byte loads in flight together. red: 768 + 3Q dwords of (dead) LDS.
	static const bool WIDE_LEAVES = Q % 4 == 0 && LDS_SOFT >= 4 * Q + 768;
	template <class SRC, class T>
	PDEV void spc_wide(int lvl, const SRC &src, Planes<T> p, T *tmp, lds_u32 *red, u32 row, bool wz)
	{
		int len = 1 << lvl, n = (len >> 3) * Q;
		int i0 = pidx(p.eb, 0);
		T *ts = tmp, *tz = tmp + n;
		PFOR_LANES(lane) {
			u32 amh[4], aml[4], apx[4];
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				amh[c] = aml[c] = 0x7F007F00u;
				apx[c] = 0;
			}
			for (int t = lane * 4; t < n; t += 256) {
				int o = t / Q, q = t % Q;
				u32x4 x[8];
#pragma unroll
				for (int j = 0; j < 8; ++j)
					x[j] = ldE(src, (8 * o + j) * Q + q, u32x4());
				u32 sg4[4] = {0, 0, 0, 0};
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					u32 xc[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
					for (int c = 0; c < 4; ++c) {
						sg4[c] |= (xc[c] & SIGN_BITS) >> j;
						amh[c] = pk_min(amh[c], abs_half(split_hi(xc[c])));
						aml[c] = pk_min(aml[c], abs_half(split_lo(xc[c])));
					}
				}
				u32x4 sv;
				sv.x = sg4[0]; sv.y = sg4[1]; sv.z = sg4[2]; sv.w = sg4[3];
				stp4(ts + t, sv);
#pragma unroll
				for (int c = 0; c < 4; ++c)
					apx[c] ^= sg4[c];
			}
#pragma unroll
			for (int c = 0; c < 4; ++c) {
				red[lane * 4 + c] = amh[c];
				red[256 + lane * 4 + c] = aml[c];
				red[512 + lane * 4 + c] = apx[c];
			}
		}
		wave_sync();
		lds_u32 *wk = red + 768; // per quad: weakest magnitude (two arrays) and odd-parity mask
		PFOR_LANES(lane) {
			if (lane < Q) {
				u32 mh = 0x7F007F00u, ml = 0x7F007F00u, px = 0;
				for (int i = lane; i < 256; i += Q) { // entry i belongs to quad i % Q
					mh = pk_min(mh, red[i]);
					ml = pk_min(ml, red[256 + i]);
					px ^= red[512 + i];
				}
				wk[lane] = mh;
				wk[Q + lane] = ml;
				wk[2 * Q + lane] = odd_mask(px);
			}
		}
		wave_sync();
		PFOR_LANES(lane) {
			for (int t = lane * 4; t < n; t += 256) {
				int o = t / Q, q = t % Q;
				u32x4 whv = ldp4(wk + q), wlv = ldp4(wk + Q + q), odv = ldp4(wk + 2 * Q + q), sv = ldp4(ts + t);
				u32 wh[4] = {whv.x, whv.y, whv.z, whv.w}, wl[4] = {wlv.x, wlv.y, wlv.z, wlv.w};
				u32 od[4] = {odv.x, odv.y, odv.z, odv.w}, sg4[4] = {sv.x, sv.y, sv.z, sv.w};
				if (od[0] | od[1] | od[2] | od[3]) {
					u32x4 x[8];
#pragma unroll
					for (int j = 0; j < 8; ++j)
						x[j] = ldE(src, (8 * o + j) * Q + q, u32x4());
					u32 tie[4] = {0, 0, 0, 0};
#pragma unroll
					for (int j = 0; j < 8; ++j) {
						u32 xc[4] = {x[j].x, x[j].y, x[j].z, x[j].w};
#pragma unroll
						for (int c = 0; c < 4; ++c)
							tie[c] |= tie_bits(xc[c], wh[c], wl[c]) >> j;
					}
#pragma unroll
					for (int c = 0; c < 4; ++c)
						sg4[c] ^= tie[c] & od[c];
				}
				u32x4 zero;
				zero.x = zero.y = zero.z = zero.w = 0;
				sv.x = sg4[0]; sv.y = sg4[1]; sv.z = sg4[2]; sv.w = sg4[3];
				stp4(p.s + i0 + t, sv);
				if (wz)
					stp4(p.z + i0 + t, zero);
#pragma unroll
				for (int c = 0; c < 4; ++c) {
					u32 z = 0;
					trans_octet(sg4[c], z, 8);
				}
				sv.x = sg4[0]; sv.y = sg4[1]; sv.z = sg4[2]; sv.w = sg4[3];
				stp4(ts + t, sv);
			}
		}
		wave_sync();
		trans_emit(ts, tz, len, 1, row, false);
	}

	PDEV void leaf(u32 op, int lvl, u32 hrd, u32 row)
	{
		if (op == OP_REP && lvl > 3)
			for (int l = lvl; l >= 1; --l)
				soft_pass<2>(l, 0);
		int n4 = Q << (lvl - 1); // 4 * (len/8) * Q: the whole level lvl-1 region
		if (lvl <= 3) {
			if (lvl == top)
				leaf_small(op, lvl, in, hrd, row);
			else
				leaf_small(op, lvl, LVec{sl + (Q << lvl)}, hrd, row);
		} else if (lvl <= RL) {
			if (lvl == top)
				leaf_big(op, lvl, in, lplanes(hrd), sl + n4, row, true);
			else
				leaf_big(op, lvl, LVec{sl + (Q << lvl)}, lplanes(hrd), sl + n4, row, true);
		} else {
			set_zc(lvl, leaf_big(op, lvl, any_src(lvl), gplanes(hrd), fregion(lvl - 1), row, false));
		}
	}

	// rate1_comb (polar_decoder.hh:76-84): node of level lvl at hrd, left child decided.
	// c: children's planes (left child at c.eb), o: the node's planes, tmp: 2*max(1,h/8)*Q dwords
	// fl: the left child's zero plane is valid. Returns "the node holds an erasure".
	template <class SRC, class TC, class TO, class T>
	PDEV u32 rate1_comb(int lvl, const SRC &src, Planes<TC> c, Planes<TO> o, T *tmp, u32 row, u32 fl)
	{
		u32 any = 0;
		int h = 1 << (lvl - 1);
		int noct = h >= 8 ? h >> 3 : 1, per = h >= 8 ? 8 : h;
		int n = noct * Q;
		T *ts = tmp, *tz = tmp + n;
		int j0 = (int)(o.eb & 7); // nonzero only when the whole node fits one octet
		int il0 = pidx(c.eb, 0), ol0 = pidx(o.eb, 0), or0 = pidx(o.eb + h, 0);
		u32 anyr = 0;
		if (Q % 4 == 0 && n >= 256) { // h >= 8: four frame quads per lane, sixteen 16-byte loads in flight
			PFOR_LANES(lane) {
				u32 acc = 0, accr = 0;
				for (int t = lane * 4; t < n; t += 256) {
					int oc = t / Q, q = t % Q;
					u32x4 x[8], y[8];
#pragma unroll
					for (int j = 0; j < 8; ++j) {
						x[j] = ldE(src, (8 * oc + j) * Q + q, u32x4());
						y[j] = ldE(src, (8 * oc + j + h) * Q + q, u32x4());
					}
					u32x4 lsv = ldp4(c.s + il0 + t), lzv;
					if (fl)
						lzv = ldp4(c.z + il0 + t);
					else
						lzv.x = lzv.y = lzv.z = lzv.w = 0;
					u32 ls[4] = {lsv.x, lsv.y, lsv.z, lsv.w}, lz[4] = {lzv.x, lzv.y, lzv.z, lzv.w};
					u32 rs[4] = {0, 0, 0, 0}, rz[4] = {0, 0, 0, 0};
#pragma unroll
					for (int j = 0; j < 8; ++j) {
						u32 xc[4] = {x[j].x, x[j].y, x[j].z, x[j].w}, yc[4] = {y[j].x, y[j].y, y[j].z, y[j].w};
#pragma unroll
						for (int k = 0; k < 4; ++k) {
							u32 r = madd4(ls[k] << j, lz[k] << j, xc[k], yc[k]);
							rs[k] |= (r & SIGN_BITS) >> j;
							rz[k] |= zero_bytes(r) >> j;
						}
					}
					u32x4 v;
					v.x = ls[0] ^ rs[0]; v.y = ls[1] ^ rs[1]; v.z = ls[2] ^ rs[2]; v.w = ls[3] ^ rs[3];
					stp4(o.s + ol0 + t, v);
					v.x = lz[0] | rz[0]; v.y = lz[1] | rz[1]; v.z = lz[2] | rz[2]; v.w = lz[3] | rz[3];
					stp4(o.z + ol0 + t, v);
					acc |= v.x | v.y | v.z | v.w;
					v.x = rs[0]; v.y = rs[1]; v.z = rs[2]; v.w = rs[3];
					stp4(o.s + or0 + t, v);
					v.x = rz[0]; v.y = rz[1]; v.z = rz[2]; v.w = rz[3];
					stp4(o.z + or0 + t, v);
					accr |= v.x | v.y | v.z | v.w;
#pragma unroll
					for (int k = 0; k < 4; ++k)
						trans_octet(rs[k], rz[k], 8);
					v.x = rs[0]; v.y = rs[1]; v.z = rs[2]; v.w = rs[3];
					stp4(ts + t, v);
					v.x = rz[0]; v.y = rz[1]; v.z = rz[2]; v.w = rz[3];
					stp4(tz + t, v);
				}
				any |= wave_any(acc);
				anyr |= wave_any(accr);
			}
			wave_sync();
			trans_emit(ts, tz, h, 0, row, anyr != 0);
			return any;
		}
		PFOR_LANES(lane) {
			u32 acc = 0, accr = 0;
			for (int t = lane; t < n; t += 64) {
				int oc = t / Q, q = t % Q;
				u32 ls = c.s[il0 + t], lz = fl ? c.z[il0 + t] : 0u;
				u32 rs = 0, rz = 0;
#pragma unroll
				for (int j = 0; j < 8; ++j)
					if (j < per) {
						int i = 8 * oc + j;
						int sh = h >= 8 ? j : j0 + j;
						u32 r = madd4(ls << sh, lz << sh, src.ld(i * Q + q), src.ld((i + h) * Q + q));
						rs |= (r & SIGN_BITS) >> j;
						rz |= zero_bytes(r) >> j;
					}
				if (h >= 8) {
					o.s[ol0 + t] = ls ^ rs;
					o.z[ol0 + t] = lz | rz;
					o.s[or0 + t] = rs;
					o.z[or0 + t] = rz;
				} else {
					u32 lm = span_mask(j0, h), rm = span_mask(j0 + h, h);
					u32 s = ls ^ ((rs >> j0) & lm), z = lz | ((rz >> j0) & lm);
					s = bfi(rm, rs >> (j0 + h), s);
					z = bfi(rm, rz >> (j0 + h), z);
					o.s[ol0 + t] = s;
					o.z[ol0 + t] = z;
				}
				acc |= lz | rz;
				accr |= rz;
				trans_octet(rs, rz, per);
				if (h <= 8) {
					emit_octet(rs, rz, h, 0, row, q);
				} else {
					ts[t] = rs;
					tz[t] = rz;
				}
			}
			any |= wave_any(acc);
			anyr |= wave_any(accr);
		}
		wave_sync();
		if (h > 8)
			trans_emit(ts, tz, h, 0, row, anyr != 0);
		return any;
	}
	PDEV void rate1_comb_any(int lvl, u32 hrd, u32 row)
	{
		// scratch: the left child's LLR region (level lvl-1), dead by now
		if (lvl <= RL) {
			lds_u32 *tmp = sl + (Q << (lvl - 1));
			if (lvl == top)
				rate1_comb(lvl, in, lplanes(hrd), lplanes(hrd), tmp, row, 1u);
			else
				rate1_comb(lvl, LVec{sl + (Q << lvl)}, lplanes(hrd), lplanes(hrd), tmp, row, 1u);
		} else {
			u32 fl = lvl - 1 > RL ? (zc >> (lvl - 1)) & 1u : 1u;
			set_zc(lvl, rate1_comb(lvl, any_src(lvl), fplanes(lvl - 1, hrd), gplanes(hrd), fregion(lvl - 1), row, fl));
		}
	}

	PDEV void node(u32 op, int lvl, u32 hrd, u32 row, u32 pad)
	{
		switch (op) {
		case OP_CHAIN: chain(lvl, hrd, pad); break;
		case OP_SUBTREE: // compiled only into the narrow-group kernels the host emits it for (program.hpp)
			if constexpr (Q <= SUBTREE_MAX_Q)
				subtree(lvl, hrd, row, pad);
			break;
		case OP_COMB: comb_any<false>(lvl, hrd); break;
		case OP_RATE0_COMB: comb_any<true>(lvl, hrd); break;
		case OP_RATE1_COMB: rate1_comb_any(lvl, hrd, row); break;
		default: leaf(op, lvl, hrd, row); break;
		}
	}

	template <class P>
	static PDEV DevOp fetch(P prog, int pc)
	{
		DevOp d;
		d.op_level = prog[pc].op_level;
		d.hrd = prog[pc].hrd;
		d.row = prog[pc].row;
		d.pad = prog[pc].pad;
		return d;
	}

	// the interpreter (polar_decoder.hh:131-464); prog is terminated by two OP_END words
	template <class P>
	PDEV void run(P prog)
	{
		zc = zl = 0;
		DevOp cur = fetch(prog, 0);
		for (int pc = 1;; ++pc) {
			u32 op = cur.op_level & 255, hrd = cur.hrd, row = cur.row, pad = cur.pad;
			int lvl = (int)(cur.op_level >> 8);
			if (op == OP_END)
				break;
			cur = fetch(prog, pc); // fetched while the current node runs
#ifdef POLAR_PROFILE
			if (TEAM && lvl == top)
				team_sync();
			unsigned long long t0_ = __builtin_amdgcn_s_memtime();
			node(op, lvl, hrd, row, pad);
			__builtin_amdgcn_s_waitcnt(0);
			unsigned long long dt_ = __builtin_amdgcn_s_memtime() - t0_;
			if (prof && (threadIdx.x & 63) == 0) {
				u32 slot = ((op == OP_CHAIN ? 10 + chain_depth(pad) : op == OP_SUBTREE ? 15 : op) * 32 + (u32)lvl) * 2;
				atomicAdd(&prof[slot], dt_);
				atomicAdd(&prof[slot + 1], 1ull);
			}
#else
			if (TEAM && lvl == top)
				team_sync();
			node(op, lvl, hrd, row, pad);
#endif

		}
	}
};

} // namespace polar
