// abi_float.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): the float bench path around the float decoder: encoders, channel, simulate.
#pragma once

// ---------------------------------------------------------------------------------------
// float path around the float decoder: the reference selects it with
// `typedef float code_type` (testbench.cc:49-53) and every functor follows. Symbols (+1 / -1 / 0)
// are exact in either type, so the encoders run on the int8 kernels between two conversions; the
// channel writes the unquantised LLR (PolarHelper<float>::quant is the identity, polar_helper.hh:28-32).
// ---------------------------------------------------------------------------------------

// out = in as int8 symbols; *bad is raised when a value is none of +1 / -1 / 0 (the int8 kernels
// multiply symbols, the reference's float encoder would multiply whatever it is given)
__global__ void __launch_bounds__(256) f32_to_symbols_kernel(int8_t *out, const float *in, size_t n, unsigned *bad)
{
	unsigned b = 0;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
		float v = in[i];
		b |= !(v == 1.0f || v == -1.0f || v == 0.0f);
		out[i] = (int8_t)((v > 0.0f) - (v < 0.0f));
	}
	if (b && bad)
		atomicOr(bad, 1u);
}
__global__ void __launch_bounds__(256) symbols_to_f32_kernel(float *out, const int8_t *in, size_t n)
{
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
		out[i] = (float)in[i];
}
// llr = (float)(fact * (x + sigma * n)) with the int8 channel's draws (channel.hpp), in double as the
// bench does (testbench.cc:151-165); counts[0] += #(llr * x < 0), counts[1] += #(llr == 0)
__global__ void __launch_bounds__(256) channel_f32_kernel(float *out, const int8_t *cw, size_t rows, size_t frames, double sigma, double fact, uint64_t seed, uint64_t frame0, unsigned long long *counts)
{
	unsigned n_flipped = 0, n_zero = 0;
	const uint64_t q0 = frame0 >> 2, nq = ((frame0 + frames - 1) >> 2) - q0 + 1;
	const size_t items = rows * nq, nt = (size_t)gridDim.x * blockDim.x;
	for (size_t it = (size_t)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += nt) {
		const size_t i = it / nq;
		const uint64_t quad = q0 + it % nq;
		PhiloxBlock b = draw(seed, STREAM_NOISE, (uint32_t)i, quad);
		double n[4];
		normal_pair(b.v[0], b.v[1], n[0], n[1]);
		normal_pair(b.v[2], b.v[3], n[2], n[3]);
#pragma unroll
		for (int c = 0; c < 4; ++c) {
			uint64_t g = 4 * quad + c;
			if (g >= frame0 && g < frame0 + frames) {
				size_t at = i * frames + (size_t)(g - frame0);
				int8_t x = cw[at];
				float l = (float)(fact * ((double)x + sigma * n[c]));
				out[at] = l;
				n_flipped += l * (float)x < 0.0f;
				n_zero += l == 0.0f;
			}
		}
	}
	if (counts) {
		if (n_flipped)
			atomicAdd(&counts[0], (unsigned long long)n_flipped);
		if (n_zero)
			atomicAdd(&counts[1], (unsigned long long)n_zero);
	}
}

static unsigned grid_for(const polar_hip_handle *h, size_t n)
{
	size_t want = (n + 255) / 256, cap = (size_t)h->cus * 32;
	return (unsigned)(want < cap ? (want ? want : 1) : cap);
}

/* PolarEncoder / PolarSysEnc with TYPE = float lanes (polar_encoder.hh:9-59 is generic in TYPE). */
extern "C" int polar_hip_encode_f32(polar_hip_handle *h, int systematic, float *codeword, const float *message, const uint8_t *frozen, int level, size_t frames, int where)
{
	if (!h || !frozen || level < 1 || level > MAX_LEVEL || (where != POLAR_HOST && where != POLAR_DEVICE))
		return fail(POLAR_E_ARG, "polar_hip_encode_f32: bad argument");
	if (frames && !codeword)
		return fail(POLAR_E_ARG, "polar_hip_encode_f32: null codeword");
	int rc = use_device(h), K = 0;
	if (rc)
		return rc;
	if (!frames)
		return POLAR_OK;
	if ((rc = upload_tables(h, frozen, level, &K)))
		return rc;
	if (K && !message)
		return fail(POLAR_E_ARG, "polar_hip_encode_f32: null message");
	const size_t n = (size_t)1 << level, km = (size_t)(K ? K : 1) * frames;
	// f32_a: int8 codeword then int8 message then the flag word; f32_b: float staging for host callers
	if ((rc = h->f32_a.reserve(n * frames + km + 16)))
		return rc;
	int8_t *cw8 = (int8_t *)h->f32_a.p, *msg8 = cw8 + n * frames;
	unsigned *bad = (unsigned *)(((uintptr_t)(msg8 + km) + 3) / 4 * 4);
	HIP_TRY(hipMemsetAsync(bad, 0, 4, h->stream));
	const float *msg = message;
	float *cw = codeword;
	if (where == POLAR_HOST) {
		if ((rc = h->f32_b.reserve((n > (size_t)K ? n : (size_t)K) * frames * sizeof(float))))
			return rc;
		if (K)
			HIP_TRY(hipMemcpyAsync(h->f32_b.p, message, (size_t)K * frames * sizeof(float), hipMemcpyHostToDevice, h->stream));
		msg = (const float *)h->f32_b.p;
		cw = (float *)h->f32_b.p;
	}
	if (K) {
		hipLaunchKernelGGL(f32_to_symbols_kernel, dim3(grid_for(h, (size_t)K * frames)), dim3(256), 0, h->stream, msg8, msg, (size_t)K * frames, bad);
		HIP_TRY(hipGetLastError());
	}
	if ((rc = encode_device(h, systematic, cw8, msg8, level, frames)))
		return rc;
	hipLaunchKernelGGL(symbols_to_f32_kernel, dim3(grid_for(h, n * frames)), dim3(256), 0, h->stream, cw, cw8, n * frames);
	HIP_TRY(hipGetLastError());
	unsigned flag = 0;
	HIP_TRY(hipMemcpyAsync(&flag, bad, 4, hipMemcpyDeviceToHost, h->stream));
	if (where == POLAR_HOST)
		HIP_TRY(hipMemcpyAsync(codeword, cw, n * frames * sizeof(float), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	if (flag)
		return fail(POLAR_E_ARG, "polar_hip_encode_f32: message symbols must be +1, -1 or 0");
	return POLAR_OK;
}

/* The bench loop of one SNR point with code_type = float (testbench.cc:49-53, 125-193): message
 * source -> encoder -> BPSK/AWGN with unquantised float LLRs -> float SC decoder -> (systematic:
 * re-encode + gather) -> the four counters. Same counter-based message and noise streams as
 * polar_hip_simulate, so the two decoders meet the very same frames. counters as polar_hip_simulate. */
extern "C" int polar_hip_simulate_f32(polar_hip_handle *h, int program_id, const uint8_t *frozen, int level, int systematic, double snr_db, uint64_t seed, uint64_t first_frame, size_t frames, int64_t *counters, int where)
{
	if (!h || !frozen || !counters || level < 1 || level > MAX_LEVEL || (where != POLAR_HOST && where != POLAR_DEVICE))
		return fail(POLAR_E_ARG, "polar_hip_simulate_f32: bad argument");
	if (program_id < 0 || (size_t)program_id >= h->programs.size() || h->programs[program_id].host.top != level)
		return fail(POLAR_E_ARG, "polar_hip_simulate_f32: program id unknown or not of this level");
	int rc = use_device(h), K = 0;
	if (rc)
		return rc;
	if ((rc = upload_tables(h, frozen, level, &K)))
		return rc;
	if ((uint32_t)K != h->programs[program_id].host.rows)
		return fail(POLAR_E_PROGRAM, "polar_hip_simulate_f32: program and frozen set disagree on K");
	const size_t n = (size_t)1 << level, kk = (size_t)(K ? K : 1);
	if ((rc = h->counters.reserve(64)))
		return rc;
	unsigned long long *c = where == POLAR_DEVICE ? (unsigned long long *)counters : (unsigned long long *)h->counters.p;
	if (where == POLAR_HOST)
		HIP_TRY(hipMemsetAsync(c, 0, 48, h->stream));
	// chunks of whole 64-frame wavefront groups: the float decoder keeps 3 N floats per frame in its arena
	size_t chunk = (size_t)8 * h->cus * 64 * 4;
	while (chunk > 64 && chunk * n * 6 > ((size_t)8 << 30))
		chunk /= 2;
	if (chunk > frames)
		chunk = frames;
	if (frames) {
		size_t tmpb = packed_tmp_bytes(level, chunk);
		if ((rc = h->sim_cw.reserve(n * chunk > tmpb ? n * chunk : tmpb)) || (rc = h->sim_msg.reserve(kk * chunk)) || (rc = h->sim_dec.reserve(kk * chunk)) ||
		    (rc = h->f32_a.reserve(n * chunk * sizeof(float))) || (rc = h->f32_b.reserve(kk * chunk * sizeof(float))))
			return rc;
	}
	int8_t *cw = (int8_t *)h->sim_cw.p, *msg = (int8_t *)h->sim_msg.p, *dec = (int8_t *)h->sim_dec.p;
	float *llr = (float *)h->f32_a.p, *decf = (float *)h->f32_b.p;
	const double sigma = channel_sigma(snr_db), fact = 2.0 / (sigma * sigma);
	for (size_t a = 0; a < frames; a += chunk) {
		size_t f = frames - a < chunk ? frames - a : chunk;
		uint64_t f0 = first_frame + a;
		if (K && (rc = launch_channel(h, 1, msg, nullptr, (size_t)K, f, 0.0, seed, f0)))
			return rc;
		if ((rc = encode_device(h, systematic, cw, msg, level, f)))
			return rc;
		size_t items = n * (((f0 + f - 1) >> 2) - (f0 >> 2) + 1);
		hipLaunchKernelGGL(channel_f32_kernel, dim3(grid_for(h, items)), dim3(256), 0, h->stream, llr, cw, n, f, sigma, fact, seed, f0, c);
		HIP_TRY(hipGetLastError());
		if ((rc = launch_decode_f32(h, program_id, decf, llr, f)))
			return rc;
		if (K) {
			hipLaunchKernelGGL(f32_to_symbols_kernel, dim3(grid_for(h, (size_t)K * f)), dim3(256), 0, h->stream, dec, decf, (size_t)K * f, (unsigned *)nullptr);
			HIP_TRY(hipGetLastError());
			if (systematic && (rc = extract_device(h, dec, nullptr, nullptr, dec, cw, level, f)))
				return rc;
			if ((rc = count_device(h, c, nullptr, nullptr, 0, (const uint8_t *)dec, (const uint8_t *)msg, (size_t)K * f)))
				return rc;
		}
	}
	hipLaunchKernelGGL(add_counters_kernel, dim3(1), dim3(1), 0, h->stream, c, (unsigned long long)frames, (unsigned long long)frames * (unsigned long long)K);
	HIP_TRY(hipGetLastError());
	if (where == POLAR_HOST) {
		int64_t tmp[6];
		HIP_TRY(hipMemcpyAsync(tmp, c, 48, hipMemcpyDeviceToHost, h->stream));
		HIP_TRY(hipStreamSynchronize(h->stream));
		for (int k = 0; k < 6; ++k)
			counters[k] += tmp[k];
		return POLAR_OK;
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

/* The float bench's channel alone (testbench.cc:151-165 with code_type = float): llr = (float)(2 / sigma^2 *
 * (codeword + sigma * n)), the same noise draws as polar_hip_channel. codeword: +1 / -1 / 0 floats;
 * llr may be the same buffer (the reference overwrites its codeword, testbench.cc:164-165). */
extern "C" int polar_hip_channel_f32(polar_hip_handle *h, float *llr, const float *codeword, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t first_frame, int where)
{
	if (!h || (where != POLAR_HOST && where != POLAR_DEVICE))
		return fail(POLAR_E_ARG, "polar_hip_channel_f32: bad argument");
	const size_t n = rows * frames;
	if (n && (!llr || !codeword))
		return fail(POLAR_E_ARG, "polar_hip_channel_f32: null buffer");
	if (rows > 0xFFFFFFFFull)
		return fail(POLAR_E_ARG, "polar_hip_channel_f32: too many rows");
	int rc = use_device(h);
	if (rc)
		return rc;
	if (!n)
		return POLAR_OK;
	if ((rc = h->f32_a.reserve(n + 16)))
		return rc;
	int8_t *cw8 = (int8_t *)h->f32_a.p;
	unsigned *bad = (unsigned *)(((uintptr_t)(cw8 + n) + 3) / 4 * 4);
	HIP_TRY(hipMemsetAsync(bad, 0, 4, h->stream));
	const float *src = codeword;
	float *dst = llr;
	if (where == POLAR_HOST) {
		if ((rc = h->f32_b.reserve(n * sizeof(float))))
			return rc;
		HIP_TRY(hipMemcpyAsync(h->f32_b.p, codeword, n * sizeof(float), hipMemcpyHostToDevice, h->stream));
		src = dst = (float *)h->f32_b.p;
	}
	hipLaunchKernelGGL(f32_to_symbols_kernel, dim3(grid_for(h, n)), dim3(256), 0, h->stream, cw8, src, n, bad);
	HIP_TRY(hipGetLastError());
	const double sigma = channel_sigma(snr_db), fact = 2.0 / (sigma * sigma);
	size_t items = rows * (((first_frame + frames - 1) >> 2) - (first_frame >> 2) + 1);
	hipLaunchKernelGGL(channel_f32_kernel, dim3(grid_for(h, items)), dim3(256), 0, h->stream, dst, cw8, rows, frames, sigma, fact, seed, first_frame, (unsigned long long *)nullptr);
	HIP_TRY(hipGetLastError());
	unsigned flag = 0;
	HIP_TRY(hipMemcpyAsync(&flag, bad, 4, hipMemcpyDeviceToHost, h->stream));
	if (where == POLAR_HOST)
		HIP_TRY(hipMemcpyAsync(llr, dst, n * sizeof(float), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	if (flag)
		return fail(POLAR_E_ARG, "polar_hip_channel_f32: codeword symbols must be +1, -1 or 0");
	return POLAR_OK;
}
