// abi_simulate.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): counters, message source, channel and the bench loop of one SNR point.
#pragma once

// ---- bench counters ----

static int count_device(polar_hip_handle *h, unsigned long long *c, const uint8_t *noisy, const uint8_t *orig, size_t n_code, const uint8_t *decoded, const uint8_t *message, size_t n_mesg)
{
	size_t work = (n_code > n_mesg ? n_code : n_mesg) / 16 / 256 + 1;
	unsigned blocks = (unsigned)(work < (size_t)h->cus * 8 ? work : (size_t)h->cus * 8);
	hipLaunchKernelGGL(count_kernel, dim3(blocks), dim3(256), 0, h->stream, c, noisy, orig, n_code, decoded, message, n_mesg);
	HIP_TRY(hipGetLastError());
	return POLAR_OK;
}

extern "C" int polar_hip_count(polar_hip_handle *h, int64_t *counters, const int8_t *noisy, const int8_t *orig, size_t n_code, const int8_t *decoded, const int8_t *message, size_t n_mesg, int where)
{
	if (!h || !counters || (where != POLAR_HOST && where != POLAR_DEVICE) || (n_code && (!noisy || !orig)) || (n_mesg && (!decoded || !message)))
		return fail(POLAR_E_ARG, "polar_hip_count: bad argument");
	int rc = use_device(h);
	if (rc)
		return rc;
	const uint8_t *p[4] = {(const uint8_t *)noisy, (const uint8_t *)orig, (const uint8_t *)decoded, (const uint8_t *)message};
	unsigned long long *c = (unsigned long long *)counters;
	if (where == POLAR_HOST) {
		// four staged arrays share stage_in; counters are accumulated on the device then added
		size_t off[5] = {0, n_code, 2 * n_code, 2 * n_code + n_mesg, 2 * (n_code + n_mesg)};
		for (int k = 0; k < 5; ++k)
			off[k] = (off[k] + 15) / 16 * 16 + 16 * k;
		if ((rc = h->stage_in.reserve(off[4] + 16)) || (rc = h->counters.reserve(32)))
			return rc;
		for (int k = 0; k < 4; ++k) {
			size_t n = k < 2 ? n_code : n_mesg;
			if (n)
				HIP_TRY(hipMemcpyAsync((uint8_t *)h->stage_in.p + off[k], p[k], n, hipMemcpyHostToDevice, h->stream));
			p[k] = (const uint8_t *)h->stage_in.p + off[k];
		}
		c = (unsigned long long *)h->counters.p;
		HIP_TRY(hipMemsetAsync(c, 0, 32, h->stream));
	}
	if ((rc = count_device(h, c, p[0], p[1], n_code, p[2], p[3], n_mesg)))
		return rc;
	if (where == POLAR_HOST) {
		int64_t tmp[4];
		HIP_TRY(hipMemcpyAsync(tmp, c, 32, hipMemcpyDeviceToHost, h->stream));
		HIP_TRY(hipStreamSynchronize(h->stream));
		for (int k = 0; k < 4; ++k)
			counters[k] += tmp[k];
		return POLAR_OK;
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

// ---- message source, channel, and the whole bench loop on the device ----

static int launch_channel(polar_hip_handle *h, int mode, int8_t *out, const int8_t *cw, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t frame0, unsigned long long *counts = nullptr)
{
	double sigma = channel_sigma(snr_db), fact = 2.0 / (sigma * sigma);
	size_t items = rows * (((frame0 + frames - 1) >> 2) - (frame0 >> 2) + 1);
	size_t want = (items + 255) / 256, cap = (size_t)h->cus * 32;
	unsigned blocks = (unsigned)(want < cap ? want : cap);
	// Screening threshold in LLR units. The single-precision normal is good to 1e-5 absolute
	// (|n| < 6.7; logf, sqrtf, sincospif within 2 ulp; u1 near 1 excluded in normal_pair_f32), which
	// the LLR scale multiplies by fact * sigma; the LLR arithmetic itself adds 128 * 2^-22. Four
	// times that bound is asked for. No screening when it would reject most values anyway.
	const bool f64_only = getenv("POLAR_HIP_CHANNEL_F64") != nullptr; // tests and experiments
	double bound = 4.0 * (1e-5 * fact * sigma + 128.0 / 4194304.0);
	float thr = f64_only || bound > 0.05 ? 0.0f : (float)bound;
	int wide = frames % 4 == 0 && frame0 % 4 == 0 && (uintptr_t)out % 4 == 0 && (uintptr_t)cw % 4 == 0;
	if (mode)
		hipLaunchKernelGGL(channel_kernel<1>, dim3(blocks), dim3(256), 0, h->stream, out, cw, rows, frames, sigma, fact, seed, frame0, thr, wide, counts);
	else
		hipLaunchKernelGGL(channel_kernel<0>, dim3(blocks), dim3(256), 0, h->stream, out, cw, rows, frames, sigma, fact, seed, frame0, thr, wide, counts);
	HIP_TRY(hipGetLastError());
	return POLAR_OK;
}

static int channel_entry(polar_hip_handle *h, int mode, int8_t *out, const int8_t *cw, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t frame0, int where, const char *who)
{
	if (!h || (where != POLAR_HOST && where != POLAR_DEVICE) || rows > 0xFFFFFFFFu || (rows && frames && (!out || (!mode && !cw))))
		return fail(POLAR_E_ARG, who);
	int rc = use_device(h);
	if (rc)
		return rc;
	if (!rows || !frames)
		return POLAR_OK;
	size_t bytes = rows * frames;
	if (where == POLAR_DEVICE) {
		if ((rc = launch_channel(h, mode, out, cw, rows, frames, snr_db, seed, frame0)))
			return rc;
		HIP_TRY(hipStreamSynchronize(h->stream));
		return POLAR_OK;
	}
	if ((rc = h->stage_in.reserve(bytes)) || (rc = h->stage_out.reserve(bytes)))
		return rc;
	if (!mode)
		HIP_TRY(hipMemcpyAsync(h->stage_in.p, cw, bytes, hipMemcpyHostToDevice, h->stream));
	if ((rc = launch_channel(h, mode, (int8_t *)h->stage_out.p, (const int8_t *)h->stage_in.p, rows, frames, snr_db, seed, frame0)))
		return rc;
	HIP_TRY(hipMemcpyAsync(out, h->stage_out.p, bytes, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

extern "C" int polar_hip_random_message(polar_hip_handle *h, int8_t *message, size_t rows, size_t frames, uint64_t seed, uint64_t first_frame, int where)
{
	return channel_entry(h, 1, message, nullptr, rows, frames, 0.0, seed, first_frame, where, "polar_hip_random_message: bad argument");
}

extern "C" int polar_hip_channel(polar_hip_handle *h, int8_t *llr, const int8_t *codeword, size_t rows, size_t frames, double snr_db, uint64_t seed, uint64_t first_frame, int where)
{
	return channel_entry(h, 0, llr, codeword, rows, frames, snr_db, seed, first_frame, where, "polar_hip_channel: bad argument");
}

extern "C" int polar_hip_simulate(polar_hip_handle *h, int program_id, const uint8_t *frozen, int level, int systematic, double snr_db, uint64_t seed, uint64_t first_frame, size_t frames, int64_t *counters, int where)
{
	if (!h || !frozen || !counters || level < 1 || level > MAX_LEVEL || (where != POLAR_HOST && where != POLAR_DEVICE))
		return fail(POLAR_E_ARG, "polar_hip_simulate: bad argument");
	if (program_id < 0 || (size_t)program_id >= h->programs.size() || h->programs[program_id].host.top != level)
		return fail(POLAR_E_ARG, "polar_hip_simulate: program id unknown or not of this level");
	int rc = use_device(h), K = 0;
	if (rc)
		return rc;
	if ((rc = upload_tables(h, frozen, level, &K)))
		return rc;
	if ((uint32_t)K != h->programs[program_id].host.rows)
		return fail(POLAR_E_PROGRAM, "polar_hip_simulate: program and frozen set disagree on K");
	const size_t n = (size_t)1 << level;
	if ((rc = h->counters.reserve(64)))
		return rc;
	unsigned long long *c = where == POLAR_DEVICE ? (unsigned long long *)counters : (unsigned long long *)h->counters.p;
	if (where == POLAR_HOST)
		HIP_TRY(hipMemsetAsync(c, 0, 48, h->stream));
	// chunk: whole 1024-frame blocks within a quarter of the free memory (and at most 64 GiB)
	size_t free_b = 0, total_b = 0;
	HIP_TRY(hipMemGetInfo(&free_b, &total_b));
	size_t held = h->sim_cw.bytes + h->sim_llr.bytes + h->sim_msg.bytes + h->sim_dec.bytes;
	size_t budget = held + free_b / 4;
	if (budget > ((size_t)64 << 30))
		budget = (size_t)64 << 30;
	size_t per_frame = 2 * n + 2 * (size_t)(K ? K : 1);
	size_t chunk = budget / per_frame;
	if (chunk >= 1024)
		chunk -= chunk % 1024;
	if (chunk < 1)
		return fail(POLAR_E_NOMEM, "polar_hip_simulate: not enough device memory for one frame");
	if (chunk > frames)
		chunk = frames;
	if (chunk >= 1024 && frames > chunk) { // equal chunks instead of full ones and a small rest (which would decode in the small-batch regime)
		size_t pieces = (frames + chunk - 1) / chunk, even = (frames + pieces - 1) / pieces;
		even = (even + 1023) / 1024 * 1024;
		if (even < chunk)
			chunk = even;
	}
	if (frames) {
		if ((rc = h->sim_cw.reserve(n * chunk > packed_tmp_bytes(level, chunk) ? n * chunk : packed_tmp_bytes(level, chunk))) || (rc = h->sim_llr.reserve(n * chunk)) || (rc = h->sim_msg.reserve((size_t)(K ? K : 1) * chunk)) || (rc = h->sim_dec.reserve((size_t)(K ? K : 1) * chunk)))
			return rc;
	}
	int8_t *cw = (int8_t *)h->sim_cw.p, *llr = (int8_t *)h->sim_llr.p, *msg = (int8_t *)h->sim_msg.p, *dec = (int8_t *)h->sim_dec.p;
	for (size_t a = 0; a < frames; a += chunk) {
		size_t f = frames - a < chunk ? frames - a : chunk;
		uint64_t f0 = first_frame + a;
		if (K && (rc = launch_channel(h, 1, msg, nullptr, (size_t)K, f, 0.0, seed, f0)))
			return rc;
		if ((rc = encode_device(h, systematic, cw, msg, level, f)))
			return rc;
		if ((rc = launch_channel(h, 0, llr, cw, n, f, snr_db, seed, f0, c))) // counts the channel errors and erasures as it goes
			return rc;
		if ((rc = launch_decode(h, program_id, dec, llr, f, h->stream, true))) // timed: polar_hip_last_decode_ms reports the last chunk
			return rc;
		if (systematic && K && (rc = extract_device(h, dec, nullptr, nullptr, dec, cw, level, f)))
			return rc;
		if (K && (rc = count_device(h, c, nullptr, nullptr, 0, (const uint8_t *)dec, (const uint8_t *)msg, (size_t)K * f)))
			return rc;
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
