// kernels_transform.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): encoder / systematic-extraction transform kernels, plane unpacking, error counting.
#pragma once

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
