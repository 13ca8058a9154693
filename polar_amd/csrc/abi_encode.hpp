// abi_encode.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): encoder, systematic extraction and plane entry points.
#pragma once

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
