// Micro-benchmark: the memory skeleton of the decode kernel. Persistent single-wave workgroups walk a
// successive-cancellation schedule over per-wave arenas: node(l) = pass(l) [reads the level-l vector,
// writes level l-1], node(l-1), pass(l), node(l-1); levels <= RL are a timed wait (the LDS subtree).
// No decoding, only the traffic: which cache policies / layouts / queue depths move it fastest?
//   ./sc_skeleton <hot_top> <U> <leaf_cycles> <waves> [mode]
//   hot_top: levels RL+1..hot_top use default-policy accesses, the rest non-temporal (RL = none)
//   U: pairs of 16-byte loads in flight per lane;  mode bit 0: every access default policy, bit 1: row layout, bits 8-: cold source levels from this one up are read with the default policy (pollution), bit 3: waves out of step (each starts at its own place in the program), bit 2: the hot levels of the per-wave layout in one compact region (+ pad vectors per wave), bits 4-7: log2 of the elements per wave piece in it
//   (element-major / wave-minor: all waves' copies of an element adjacent); jitter: random extra leaf cycles
//   hipcc --offload-arch=gfx950 -O3 -o sc_skeleton sc_skeleton.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

enum { TOP = 16, RL = 6, QV = 8 }; // 128 frames per wave: 8 vectors per element

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4 *p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// vector v of a wave's level in the row layout with blocks of 2^lb elements: [element block][wave][2^lb elements][QV]
// (rs = resident waves * QV vectors per element row; the wave's column offset is already in the base pointer, scaled by 2^lb)
__device__ __forceinline__ size_t rowaddr(int v, size_t rs, int lb)
{
	size_t e = (size_t)(v / QV);
	return ((e >> lb) * rs << lb) + ((e & (((size_t)1 << lb) - 1)) * QV) + (v % QV);
}
// ROWS: element-major / wave-minor layout: vector v of a level lives at (v / QV) * rs + (v % QV) from
// the wave's column (rs = resident waves * QV), so that all waves' copies of element i are adjacent
template <int U, bool SNT, bool DNT, bool ROWS>
__device__ __forceinline__ u32 pass(const u32x4 *src, u32x4 *dst, int n, int lane, size_t rs, int lb)
{
	u32 acc = 0;
	for (int t = lane; t < n; t += 64 * U) { // n is a multiple of 64 * U (checked on the host)
		u32x4 x[U], y[U];
#pragma unroll
		for (int u = 0; u < U; ++u) {
			int v = t + 64 * u;
			size_t a = ROWS ? rowaddr(v, rs, lb) : (size_t)v, b = ROWS ? rowaddr(v + n, rs, lb) : (size_t)(v + n);
			x[u] = ld<SNT>(src + a);
			y[u] = ld<SNT>(src + b);
		}
#pragma unroll
		for (int u = 0; u < U; ++u) {
			int v = t + 64 * u;
			u32x4 r = x[u] ^ y[u];
			r.x += 0x9E3779B9u;
			acc ^= r.y;
			if (dst)
				st<DNT>(dst + (ROWS ? rowaddr(v, rs, lb) : (size_t)v), r);
		}
	}
	return acc;
}

template <int U, bool ROWS>
__global__ void __launch_bounds__(64) skel(u32x4 *arenas, size_t arena_vec, const u32x4 *input, size_t in_vec, const int *prog, int nops,
	int hot_top, int leaf_cycles, int all_default, int jitter, int lb, int compact, int rotate, int pollute, u32x4 *hotbase, size_t pad, u32 *sink)
{
	const int lane = threadIdx.x;
	// ROWS: level l of all waves is one region of gridDim * (QV << l) vectors at gridDim * (QV << l)
	const size_t nw = gridDim.x, rs = nw * QV;
	u32x4 *arena = ROWS ? arenas + ((size_t)blockIdx.x * QV << lb) : arenas + (size_t)blockIdx.x * arena_vec;
	const u32x4 *in = ROWS ? input + ((size_t)blockIdx.x * QV << lb) : input + (size_t)blockIdx.x * in_vec;
	u32 acc = 0, rnd = blockIdx.x * 2654435761u + 12345u;
	// mode bit 3: every wave starts at its own place in the program and wraps around (no data depends
	// on the order here), so that the waves are out of step with one another from the start
	const int pc0 = rotate ? (int)((blockIdx.x * 2654435761u >> 8) % (unsigned)nops) : 0;
	for (int k = 0; k < nops; ++k) {
		int pc = k + pc0 < nops ? k + pc0 : k + pc0 - nops;
		int l = prog[pc]; // level of the pass, or 0 = leaf
		if (l == 0) {
			rnd = rnd * 1664525u + 1013904223u;
			unsigned long long wait = (unsigned long long)leaf_cycles + (jitter ? (rnd >> 8) % (unsigned)jitter : 0u);
			unsigned long long t0 = __builtin_amdgcn_s_memtime();
			while (__builtin_amdgcn_s_memtime() - t0 < wait)
				__builtin_amdgcn_s_sleep(8);
			continue;
		}
		const int n = QV << (l - 1);
		const u32x4 *src = l == TOP ? in : arena + (ROWS ? nw : 1) * ((size_t)QV << l);
		u32x4 *dst = l - 1 > RL ? arena + (ROWS ? nw : 1) * ((size_t)QV << (l - 1)) : nullptr;
		if (compact && !ROWS) { // hot levels RL+1..hot_top: wave b's block of (QV << (hot_top + 1)) - (QV << (RL + 1)) vectors
			const size_t hv = ((size_t)QV << (hot_top + 1)) - ((size_t)QV << (RL + 1));
			u32x4 *hb = hotbase + (size_t)blockIdx.x * (hv + pad) - ((size_t)QV << (RL + 1));
			if (l != TOP && l <= hot_top)
				src = hb + ((size_t)QV << l);
			if (dst && l - 1 <= hot_top)
				dst = hb + ((size_t)QV << (l - 1));
		}
		const bool shot = all_default || (l != TOP && l <= hot_top) || (l >= pollute), dhot = all_default || l - 1 <= hot_top; // pollute: source levels from this one up are read with the default policy
		if (shot && dhot) acc ^= pass<U, false, false, ROWS>(src, dst, n, lane, rs, lb);
		else if (shot) acc ^= pass<U, false, true, ROWS>(src, dst, n, lane, rs, lb);
		else if (dhot) acc ^= pass<U, true, false, ROWS>(src, dst, n, lane, rs, lb);
		else acc ^= pass<U, true, true, ROWS>(src, dst, n, lane, rs, lb);
	}
	sink[blockIdx.x * 64 + lane] = acc;
}

static void gen(std::vector<int> &p, int l)
{
	if (l == RL) {
		p.push_back(0);
		return;
	}
	p.push_back(l);
	gen(p, l - 1);
	p.push_back(l);
	gen(p, l - 1);
}

int main(int argc, char **argv)
{
	int hot_top = argc > 1 ? atoi(argv[1]) : RL, U = argc > 2 ? atoi(argv[2]) : 8, leaf = argc > 3 ? atoi(argv[3]) : 40000;
	int waves = argc > 4 ? atoi(argv[4]) : 2048, mode = argc > 5 ? atoi(argv[5]) : 0, jitter = argc > 6 ? atoi(argv[6]) : 0, pad = argc > 7 ? atoi(argv[7]) : 0;
	if (hot_top > 13 || pad < 0 || pad > 4096) { printf("hot_top <= 13, pad 0..4096 vectors\n"); return 1; }
	std::vector<int> prog;
	gen(prog, TOP);
	const size_t arena_vec = (size_t)QV << TOP, in_vec = (size_t)QV << TOP; // 8 MiB each per wave
	if (U != 1 && U != 2 && U != 4 && U != 8) { printf("U must be 1, 2, 4 or 8\n"); return 1; }
	if ((QV << RL) % (64 * U)) { printf("bad U\n"); return 1; } // smallest pass: n = QV << RL
	if (waves < 1 || waves > 2048) { printf("waves 1..2048\n"); return 1; }
	u32x4 *arenas, *input;
	u32 *sink;
	int *dprog;
	CK(hipMalloc(&arenas, waves * arena_vec * 16));
	CK(hipMalloc(&input, waves * in_vec * 16));
	CK(hipMalloc(&sink, waves * 64 * 4));
	u32x4 *hot;
	CK(hipMalloc(&hot, (size_t)waves * (((size_t)QV << 14) + 4096) * 16)); // room for hot_top <= 13 plus padding
	CK(hipMemset(hot, 3, (size_t)waves * (((size_t)QV << 14) + 4096) * 16));
	CK(hipMalloc(&dprog, prog.size() * 4));
	CK(hipMemset(arenas, 1, waves * arena_vec * 16));
	CK(hipMemset(input, 2, waves * in_vec * 16));
	CK(hipMemcpy(dprog, prog.data(), prog.size() * 4, hipMemcpyHostToDevice));
	double bytes = 0;
	for (int l : prog)
		if (l)
			bytes += 16.0 * ((QV << l) + (l - 1 > RL ? QV << (l - 1) : 0));
	bytes *= waves;
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	float best = 1e30f;
	for (int rep = 0; rep < 3; ++rep) {
		CK(hipEventRecord(a));
#define LAUNCH(UU, RR) hipLaunchKernelGGL((skel<UU, RR>), dim3(waves), dim3(64), 0, 0, arenas, arena_vec, input, in_vec, dprog, (int)prog.size(), hot_top, leaf, mode & 1, jitter, (mode >> 4) & 15, (mode >> 2) & 1, (mode >> 3) & 1, (mode >> 8) ? (mode >> 8) : 99, hot, (size_t)pad, sink)
		if (mode & 2) {
			switch (U) { case 1: LAUNCH(1, true); break; case 2: LAUNCH(2, true); break; case 4: LAUNCH(4, true); break; default: LAUNCH(8, true); break; }
		} else {
			switch (U) { case 1: LAUNCH(1, false); break; case 2: LAUNCH(2, false); break; case 4: LAUNCH(4, false); break; default: LAUNCH(8, false); break; }
		}
		CK(hipEventRecord(b));
		CK(hipEventSynchronize(b));
		CK(hipGetLastError());
		float ms;
		CK(hipEventElapsedTime(&ms, a, b));
		if (ms < best)
			best = ms;
	}
	printf("hot_top %2d U %d leaf %6d (+%d jitter) cycles waves %4d mode %d: %8.2f ms  %.2f TB/s (%.1f GB per launch)\n", hot_top, U, leaf, jitter, waves, mode, best, bytes / best / 1e9, bytes / 1e9);
	return 0;
}
