// Micro-benchmark: what the HBM / Infinity Cache path of an MI355X gives to kernels shaped like the
// decode kernel, and what it gives to the textbook shapes the microarchitecture guide quotes.
//   part 1: classic grid (one 16-byte vector per thread, 256-thread workgroups): read-only (loads
//           consumed), write-only, copy, 3R:2W; default and non-temporal policy
//   part 2: persistent single-wave workgroups, 3R:2W, chip-wide interleave at 1 KiB / 4 KiB
//   part 3: short-lived "hot" vectors (written, then read back after D bytes of other traffic, like
//           the scratch levels just above the LDS levels) beside streaming "cold" traffic: does the
//           256 MiB Infinity Cache serve them, and does the cache policy of either kind matter?
//   hipcc --offload-arch=gfx950 -O3 -o hbm_ceiling hbm_ceiling.hip && ./hbm_ceiling
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld(const u32x4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <bool NT> __device__ __forceinline__ void st(u32x4 *p, u32x4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// ---- part 1: classic grids ----
template <bool NT, int R, int W>
__global__ void __launch_bounds__(256) classic(const u32x4 *in, u32x4 *out, size_t per_stream, u32x4 *sink)
{
	size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
	u32x4 acc = {(u32)t, 1, 2, 3};
#pragma unroll
	for (int r = 0; r < R; ++r)
		acc ^= ld<NT>(in + r * per_stream + t);
#pragma unroll
	for (int w = 0; w < W; ++w) {
		acc.x += w;
		st<NT>(out + w * per_stream + t, acc);
	}
	if (W == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) // consumes the loads; practically never true
		sink[0] = acc;
}
// grid-stride form of the same: B blocks, each thread walks t, t + B*256, ...
template <bool NT, int R, int W, int U>
__global__ void __launch_bounds__(256) strided(const u32x4 *in, u32x4 *out, size_t per_stream, u32x4 *sink)
{
	const size_t step = (size_t)gridDim.x * 256;
	u32x4 tot = {0, 0, 0, 0};
	for (size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x; t0 + (U - 1) * step < per_stream; t0 += U * step) {
		u32x4 x[R > 0 ? R : 1][U];
#pragma unroll
		for (int r = 0; r < R; ++r)
#pragma unroll
			for (int u = 0; u < U; ++u)
				x[r][u] = ld<NT>(in + r * per_stream + t0 + u * step);
#pragma unroll
		for (int u = 0; u < U; ++u) {
			u32x4 acc = {(u32)t0, 1, 2, 3};
#pragma unroll
			for (int r = 0; r < R; ++r)
				acc ^= x[r][u];
#pragma unroll
			for (int w = 0; w < W; ++w) {
				acc.x += w;
				st<NT>(out + w * per_stream + t0 + u * step, acc);
			}
			tot ^= acc;
		}
	}
	if (W == 0 && (tot.x ^ tot.y ^ tot.z ^ tot.w) == 0x12345u)
		sink[0] = tot;
}

static float time_launches(void (*launch)(void *), void *ctx, int reps)
{
	hipEvent_t a, b;
	CK(hipEventCreate(&a));
	CK(hipEventCreate(&b));
	launch(ctx);
	CK(hipEventRecord(a));
	for (int i = 0; i < reps; ++i)
		launch(ctx);
	CK(hipEventRecord(b));
	CK(hipEventSynchronize(b));
	CK(hipGetLastError());
	float ms;
	CK(hipEventElapsedTime(&ms, a, b));
	CK(hipEventDestroy(a));
	CK(hipEventDestroy(b));
	return ms / reps;
}

struct Ctx { const u32x4 *in; u32x4 *out, *sink; size_t per_stream; int blocks; };

template <bool NT, int R, int W> static void run_classic(const char *name, Ctx c)
{
	auto l = [](void *p) {
		Ctx *c = (Ctx *)p;
		hipLaunchKernelGGL((classic<NT, R, W>), dim3((unsigned)(c->per_stream / 256)), dim3(256), 0, 0, c->in, c->out, c->per_stream, c->sink);
	};
	float ms = time_launches(l, &c, 3);
	printf("classic   %-28s %s: %.2f TB/s\n", name, NT ? "nt     " : "default", (double)(R + W) * c.per_stream * 16 / ms / 1e9);
}
template <bool NT, int R, int W, int U> static void run_strided(const char *name, Ctx c, int blocks)
{
	c.blocks = blocks;
	auto l = [](void *p) {
		Ctx *c = (Ctx *)p;
		hipLaunchKernelGGL((strided<NT, R, W, U>), dim3(c->blocks), dim3(256), 0, 0, c->in, c->out, c->per_stream, c->sink);
	};
	float ms = time_launches(l, &c, 3);
	printf("strided   %-28s %s %5d blocks x%d: %.2f TB/s\n", name, NT ? "nt     " : "default", blocks, U, (double)(R + W) * c.per_stream * 16 / ms / 1e9);
}

// ---- part 2: persistent single-wave workgroups, chip-wide interleave at CH vectors ----
template <bool NT, int CH>
__global__ void __launch_bounds__(64) persistent32(const u32x4 *in, u32x4 *out, size_t per_stream)
{
	// CH vectors of every stream per step and wave (CH = 64: 1 KiB, 256: 4 KiB); wave b takes chunks b, b + nb, ...
	const size_t chunks = per_stream / CH, nb = gridDim.x;
	for (size_t c = blockIdx.x; c < chunks; c += nb) {
		size_t t = c * CH + threadIdx.x;
		u32x4 x[3][CH / 64];
#pragma unroll
		for (int r = 0; r < 3; ++r)
#pragma unroll
			for (int u = 0; u < CH / 64; ++u)
				x[r][u] = ld<NT>(in + r * per_stream + t + 64 * u);
#pragma unroll
		for (int u = 0; u < CH / 64; ++u) {
			u32x4 acc = x[0][u] ^ x[1][u] ^ x[2][u];
			st<NT>(out + t + 64 * u, acc);
			acc.x += 1;
			st<NT>(out + per_stream + t + 64 * u, acc);
		}
	}
}

// ---- part 2b: what separates the classic grid from the persistent one? ----
// persistent 256-thread workgroups doing exactly the classic kernel's work per step (3 loads, 2 stores
// per thread), chunk c = step * nb + b (chip-wide moving window) or a contiguous slice per workgroup
template <bool NT, bool SLICE>
__global__ void __launch_bounds__(256) persistent_classic(const u32x4 *in, u32x4 *out, size_t per_stream)
{
	const size_t chunks = per_stream / 256, nb = gridDim.x, mine = chunks / nb;
	for (size_t k = 0; k < mine; ++k) {
		size_t c = SLICE ? blockIdx.x * mine + k : k * nb + blockIdx.x;
		size_t t = c * 256 + threadIdx.x;
		u32x4 acc = ld<NT>(in + t) ^ ld<NT>(in + per_stream + t) ^ ld<NT>(in + 2 * per_stream + t);
		st<NT>(out + t, acc);
		acc.x += 1;
		st<NT>(out + per_stream + t, acc);
	}
}
// classic grid with V vectors per thread (V consecutive 4 KiB chunks per workgroup)
template <bool NT, int V>
__global__ void __launch_bounds__(256) classic_v(const u32x4 *in, u32x4 *out, size_t per_stream)
{
	size_t t0 = ((size_t)blockIdx.x * V) * 256 + threadIdx.x;
	u32x4 x[3][V];
#pragma unroll
	for (int r = 0; r < 3; ++r)
#pragma unroll
		for (int v = 0; v < V; ++v)
			x[r][v] = ld<NT>(in + r * per_stream + t0 + 256 * v);
#pragma unroll
	for (int v = 0; v < V; ++v) {
		u32x4 acc = x[0][v] ^ x[1][v] ^ x[2][v];
		st<NT>(out + t0 + 256 * v, acc);
		acc.x += 1;
		st<NT>(out + per_stream + t0 + 256 * v, acc);
	}
}
// classic grid, one vector per thread, but workgroup b works on chunk perm(b): scattered instead of a moving window
template <bool NT>
__global__ void __launch_bounds__(256) classic_scattered(const u32x4 *in, u32x4 *out, size_t per_stream, unsigned mask, int gs)
{
	// groups of 2^gs neighbouring workgroups (dispatched together) keep neighbouring chunks; the groups are scattered
	unsigned b = blockIdx.x;
	unsigned c = ((((b >> gs) * 2654435761u) << gs) | (b & ((1u << gs) - 1))) & mask; // a permutation of the chunk index (chunks = mask + 1, a power of two)
	size_t t = (size_t)c * 256 + threadIdx.x;
	u32x4 acc = ld<NT>(in + t) ^ ld<NT>(in + per_stream + t) ^ ld<NT>(in + 2 * per_stream + t);
	st<NT>(out + t, acc);
	acc.x += 1;
	st<NT>(out + per_stream + t, acc);
}

// ---- part 3: hot vectors beside cold streams ----
// Per iteration and wave: write hv hot vectors, read cr_n cold vectors, write cw_n cold vectors, read
// the hv hot vectors back. REUSE: the hot vectors live at a fixed place per wave (total hot footprint
// = waves * hv * 16 bytes, rewritten every iteration); otherwise they advance through a large region
// like the cold streams (same instructions, nothing to reuse). Policy per access kind: hot load (HL),
// hot store (HS), cold load (CL), cold store (CS); true = non-temporal.
template <bool HL, bool HS, bool CL, bool CS, bool REUSE>
__global__ void __launch_bounds__(64) hot_cold(u32x4 *hot, const u32x4 *cold_r, u32x4 *cold_w, u32x4 *sink, int hv, int cr_n, int cw_n, int iters,
	size_t hot_pw, size_t cr_pw, size_t cw_pw)
{
	const size_t b = blockIdx.x;
	const int lane = threadIdx.x;
	u32x4 acc = {(u32)b, (u32)lane, 7, 9};
	size_t hpos = 0, cr = 0, cw = 0;
	u32x4 *hbase = hot + b * hot_pw;
	const u32x4 *rbase = cold_r + b * cr_pw;
	u32x4 *wbase = cold_w + b * cw_pw;
	for (int it = 0; it < iters; ++it) {
		u32x4 *h = hbase + (REUSE ? 0 : hpos);
		for (int t = lane; t < hv; t += 256) {
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				acc.x += 0x9E3779B9u;
				st<HS>(h + t + 64 * u, acc);
			}
		}
		for (int t = lane; t < cr_n; t += 256) { // hv, cr_n, cw_n: multiples of 256 (checked on the host)
			u32x4 x[4];
#pragma unroll
			for (int u = 0; u < 4; ++u)
				x[u] = ld<CL>(rbase + cr + t + 64 * u);
#pragma unroll
			for (int u = 0; u < 4; ++u)
				acc.y ^= x[u].x ^ x[u].z;
		}
		for (int t = lane; t < cw_n; t += 256) {
#pragma unroll
			for (int u = 0; u < 4; ++u) {
				acc.w += 0x7F4A7C15u;
				st<CS>(wbase + cw + t + 64 * u, acc);
			}
		}
		for (int t = lane; t < hv; t += 256) {
			u32x4 x[4];
#pragma unroll
			for (int u = 0; u < 4; ++u)
				x[u] = ld<HL>(h + t + 64 * u);
#pragma unroll
			for (int u = 0; u < 4; ++u)
				acc.z ^= x[u].x ^ x[u].w;
		}
		hpos += hv; if (hpos + hv > hot_pw) hpos = 0;
		cr += cr_n; if (cr + cr_n > cr_pw) cr = 0;
		cw += cw_n; if (cw + cw_n > cw_pw) cw = 0;
	}
	sink[b * 64 + lane] = acc;
}

struct HC { u32x4 *hot; const u32x4 *cr; u32x4 *cw, *sink; int hv, cr_n, cw_n, iters, waves; size_t hot_pw, cr_pw, cw_pw; };
template <bool HL, bool HS, bool CL, bool CS, bool REUSE> static double run_hc(HC c)
{
	auto l = [](void *p) {
		HC *c = (HC *)p;
		hipLaunchKernelGGL((hot_cold<HL, HS, CL, CS, REUSE>), dim3(c->waves), dim3(64), 0, 0, c->hot, c->cr, c->cw, c->sink, c->hv, c->cr_n, c->cw_n, c->iters, c->hot_pw, c->cr_pw, c->cw_pw);
	};
	if (c.hv % 256 || c.cr_n % 256 || c.cw_n % 256 || (size_t)c.hv > c.hot_pw || (size_t)c.cr_n > c.cr_pw || (size_t)c.cw_n > c.cw_pw) {
		printf("bad shape\n");
		exit(1);
	}
	float ms = time_launches(l, &c, 2);
	return (double)c.waves * c.iters * (2.0 * c.hv + c.cr_n + c.cw_n) * 16 / ms / 1e9;
}

int main(int argc, char **argv)
{
	int parts = argc > 1 ? atoi(argv[1]) : 7;
	const size_t per_stream = (size_t)1 << 27; // 2 GiB per stream
	u32x4 *in, *out, *sink;
	CK(hipMalloc(&in, 4 * per_stream * 16));   // 8 GiB
	CK(hipMalloc(&out, 2 * per_stream * 16));  // 4 GiB
	CK(hipMalloc(&sink, 1 << 22));
	CK(hipMemset(in, 1, 4 * per_stream * 16));
	CK(hipMemset(out, 2, 2 * per_stream * 16));
	Ctx c{in, out, sink, per_stream, 0};
	if (parts & 1) {
		run_classic<false, 1, 0>("read only (consumed)", c);
		run_classic<true, 1, 0>("read only (consumed)", c);
		run_classic<false, 0, 1>("write only", c);
		run_classic<true, 0, 1>("write only", c);
		run_classic<false, 1, 1>("copy 1R:1W", c);
		run_classic<true, 1, 1>("copy 1R:1W", c);
		run_classic<false, 3, 2>("3R:2W", c);
		run_classic<true, 3, 2>("3R:2W", c);
		for (int blocks : {512, 1024, 2048, 4096}) {
			run_strided<false, 1, 0, 4>("read only (consumed)", c, blocks);
			run_strided<true, 1, 0, 4>("read only (consumed)", c, blocks);
			run_strided<false, 1, 1, 4>("copy 1R:1W", c, blocks);
			run_strided<true, 1, 1, 4>("copy 1R:1W", c, blocks);
			run_strided<false, 3, 2, 4>("3R:2W", c, blocks);
			run_strided<true, 3, 2, 4>("3R:2W", c, blocks);
		}
		run_strided<true, 1, 0, 8>("read only (consumed)", c, 2048);
		run_strided<true, 3, 2, 8>("3R:2W", c, 2048);
	}
	if (parts & 2) {
		for (int waves : {2048, 4096}) {
			Ctx d = c;
			d.blocks = waves;
			auto l1 = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((persistent32<true, 64>), dim3(c->blocks), dim3(64), 0, 0, c->in, c->out, c->per_stream); };
			auto l4 = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((persistent32<true, 256>), dim3(c->blocks), dim3(64), 0, 0, c->in, c->out, c->per_stream); };
			auto l4d = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((persistent32<false, 256>), dim3(c->blocks), dim3(64), 0, 0, c->in, c->out, c->per_stream); };
			printf("persistent 3R:2W %d waves, interleave 1 KiB nt: %.2f TB/s\n", waves, 5.0 * per_stream * 16 / time_launches(l1, &d, 3) / 1e9);
			printf("persistent 3R:2W %d waves, interleave 4 KiB nt: %.2f TB/s\n", waves, 5.0 * per_stream * 16 / time_launches(l4, &d, 3) / 1e9);
			printf("persistent 3R:2W %d waves, interleave 4 KiB default: %.2f TB/s\n", waves, 5.0 * per_stream * 16 / time_launches(l4d, &d, 3) / 1e9);
		}
	}
	if (parts & 8) {
		Ctx d = c;
		for (int per_cu : {2, 4, 8}) {
			d.blocks = 256 * per_cu;
			auto a1 = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((persistent_classic<true, false>), dim3(c->blocks), dim3(256), 0, 0, c->in, c->out, c->per_stream); };
			auto a2 = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((persistent_classic<true, true>), dim3(c->blocks), dim3(256), 0, 0, c->in, c->out, c->per_stream); };
			auto a3 = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((persistent_classic<false, false>), dim3(c->blocks), dim3(256), 0, 0, c->in, c->out, c->per_stream); };
			printf("persistent classic-step 3R:2W, %d x 256 threads per CU: moving window nt %.2f, slices nt %.2f, moving window default %.2f TB/s\n", per_cu,
				5.0 * per_stream * 16 / time_launches(a1, &d, 3) / 1e9, 5.0 * per_stream * 16 / time_launches(a2, &d, 3) / 1e9, 5.0 * per_stream * 16 / time_launches(a3, &d, 3) / 1e9);
		}
		auto v1 = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((classic_v<true, 1>), dim3((unsigned)(c->per_stream / 256)), dim3(256), 0, 0, c->in, c->out, c->per_stream); };
		auto v2 = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((classic_v<true, 2>), dim3((unsigned)(c->per_stream / 512)), dim3(256), 0, 0, c->in, c->out, c->per_stream); };
		auto v4 = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((classic_v<true, 4>), dim3((unsigned)(c->per_stream / 1024)), dim3(256), 0, 0, c->in, c->out, c->per_stream); };
		auto v8 = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((classic_v<true, 8>), dim3((unsigned)(c->per_stream / 2048)), dim3(256), 0, 0, c->in, c->out, c->per_stream); };
		printf("classic 3R:2W nt, vectors per thread 1 / 2 / 4 / 8: %.2f %.2f %.2f %.2f TB/s\n", 5.0 * per_stream * 16 / time_launches(v1, &d, 3) / 1e9,
			5.0 * per_stream * 16 / time_launches(v2, &d, 3) / 1e9, 5.0 * per_stream * 16 / time_launches(v4, &d, 3) / 1e9, 5.0 * per_stream * 16 / time_launches(v8, &d, 3) / 1e9);
		// the same scattered-chunk kernel confined to smaller regions (5 streams of 2^lg vectors; repeated to move as many bytes)
		for (int lg = 22; lg <= 27; ++lg) {
			Ctx e = c;
			e.per_stream = (size_t)1 << lg;
			e.blocks = 0;
			auto sc = [](void *p) {
				Ctx *c = (Ctx *)p;
				for (size_t r = 0; r < ((size_t)1 << 27) / c->per_stream; ++r)
					hipLaunchKernelGGL((classic_scattered<true>), dim3((unsigned)(c->per_stream / 256)), dim3(256), 0, 0, c->in, c->out, c->per_stream, (unsigned)(c->per_stream / 256 - 1), c->blocks);
			};
			auto io = [](void *p) {
				Ctx *c = (Ctx *)p;
				for (size_t r = 0; r < ((size_t)1 << 27) / c->per_stream; ++r)
					hipLaunchKernelGGL((classic_v<true, 1>), dim3((unsigned)(c->per_stream / 256)), dim3(256), 0, 0, c->in, c->out, c->per_stream);
			};
			printf("5 streams of %5zu MiB: scattered 4 KiB chunks %.2f TB/s, in order %.2f TB/s\n", e.per_stream * 16 >> 20, 5.0 * per_stream * 16 / time_launches(sc, &e, 3) / 1e9, 5.0 * per_stream * 16 / time_launches(io, &e, 3) / 1e9);
		}
		for (int gs = 0; gs <= 10; gs += 5) {
			d.blocks = gs;
			auto sc = [](void *p) { Ctx *c = (Ctx *)p; hipLaunchKernelGGL((classic_scattered<true>), dim3((unsigned)(c->per_stream / 256)), dim3(256), 0, 0, c->in, c->out, c->per_stream, (unsigned)(c->per_stream / 256 - 1), c->blocks); };
			printf("classic 3R:2W nt, scattered pieces of %4d KiB per stream: %.2f TB/s\n", 4 << gs, 5.0 * per_stream * 16 / time_launches(sc, &d, 3) / 1e9);
		}
	}
	if (parts & 4) {
		const int waves = 2048;
		HC h;
		h.waves = waves;
		h.sink = sink;
		h.cr = in;                                  // 8 GiB: 4 MiB per wave
		h.cr_pw = 4 * per_stream / waves;
		h.cw = out;                                 // first 2 GiB of out: 1 MiB per wave
		h.cw_pw = per_stream / waves;
		h.hot = out + per_stream;                   // second 2 GiB of out: 1 MiB per wave
		h.hot_pw = per_stream / waves;
		printf("hot/cold, %d persistent waves; policy letters: hot load, hot store, cold load, cold store (d = default, n = non-temporal); TB/s of all bytes\n", waves);
		// cold reads : cold writes per hot vector, x2 (hot share of the bytes: 25 %, 50 %, 67 %)
		const int mixes[3][2] = {{8, 4}, {2, 2}, {2, 0}};
		for (int m = 0; m < 3; ++m) {
			int a = mixes[m][0], b = mixes[m][1];
			printf("cold per hot vector: %.1f read, %.1f written (hot share %.0f %%)\n", a / 2.0, b / 2.0, 100.0 * 4 / (4 + a + b));
			printf("%-9s %-9s | reuse: ddnn   ddnd   ddnn->nodirty dndn  ndnn   nnnn   dddd | no reuse: ddnn   nnnn   dddd   nndd(ld nt, st dflt)  ddnn\n", "hot/wave", "hot total");
			for (int hkb : {4, 16, 32, 64, 128}) {
				h.hv = hkb * 1024 / 16;
				h.cr_n = h.hv * a / 2;
				h.cw_n = h.hv * b / 2;
				h.iters = (int)(32.0 * 1024 * 1024 / 16 / (2.0 * h.hv + h.cr_n + h.cw_n)); // 32 MiB per wave and launch
				double r1 = run_hc<false, false, true, true, true>(h);
				double r2 = run_hc<false, false, true, false, true>(h);
				double r3 = run_hc<false, true, false, true, true>(h);
				double r4 = run_hc<true, false, true, true, true>(h);
				double r5 = run_hc<true, true, true, true, true>(h);
				double r6 = run_hc<false, false, false, false, true>(h);
				double n1 = run_hc<false, false, true, true, false>(h);
				double n2 = run_hc<true, true, true, true, false>(h);
				double n3 = run_hc<false, false, false, false, false>(h);
				double n4 = run_hc<true, false, true, false, false>(h);
				printf("%4d KiB  %5d MiB | %6.2f %6.2f %6s %6.2f %6.2f %6.2f %6.2f | %6.2f %6.2f %6.2f %6.2f\n", hkb, hkb * waves / 1024, r1, r2, "", r3, r4, r5, r6, n1, n2, n3, n4);
			}
		}
	}
	return 0;
}
