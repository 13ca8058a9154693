// abi_decode.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): decode entry points (int8, asynchronous, allocation-free, float).
#pragma once

extern "C" int polar_hip_decode(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, int where)
{
	int rc = check_decode_args(h, program_id, message, codeword, frames);
	if (rc)
		return rc;
	if (where != POLAR_HOST && where != POLAR_DEVICE)
		return fail(POLAR_E_ARG, "polar_hip_decode: bad `where`");
	if ((rc = use_device(h)))
		return rc;
	if (!frames)
		return POLAR_OK;
	const HostProgram &hp = h->programs[program_id].host;
	if (where == POLAR_DEVICE) {
		if ((rc = launch_decode(h, program_id, message, codeword, frames, h->stream, true, true))) // a resident batch: the arenas' placement is watched
			return rc;
		HIP_TRY(hipStreamSynchronize(h->stream));
		return POLAR_OK;
	}
	const size_t chunk = pipeline_chunk((size_t)1 << hp.top);
	if (frames >= 2 * chunk) {
		if (host_pinned(codeword) && (!hp.rows || host_pinned(message)))
			return decode_host_pipelined(h, program_id, message, codeword, frames, chunk);
		if ((chunk << hp.top) <= ((size_t)1 << 30)) // two page-locked slots of a chunk each way: 3 GiB at most
			return decode_host_staged(h, program_id, message, codeword, frames, chunk);
	}
	size_t in_bytes = ((size_t)1 << hp.top) * frames, out_bytes = (size_t)hp.rows * frames;
	if ((rc = h->stage_in.reserve(in_bytes)) || (rc = h->stage_out.reserve(out_bytes ? out_bytes : 4)))
		return rc;
	HIP_TRY(hipMemcpyAsync(h->stage_in.p, codeword, in_bytes, hipMemcpyHostToDevice, h->stream));
	if ((rc = launch_decode(h, program_id, (int8_t *)h->stage_out.p, (const int8_t *)h->stage_in.p, frames, h->stream, true)))
		return rc;
	if (out_bytes)
		HIP_TRY(hipMemcpyAsync(message, h->stage_out.p, out_bytes, hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

// The caller's rows may be further apart than `frames` bytes (views into wider tensors, or rows padded on purpose:
// a pitch of frames + 4352 instead of 2^20 takes 1.5 % off the decode of N = 2^16, profiles/r3_row_pitch.txt).
extern "C" int polar_hip_decode_pitched(polar_hip_handle *h, int program_id, int8_t *message, size_t message_pitch, const int8_t *codeword, size_t codeword_pitch, size_t frames)
{
	int rc = check_decode_args(h, program_id, message, codeword, frames);
	if (rc)
		return rc;
	if ((codeword_pitch && codeword_pitch < frames) || (message_pitch && message_pitch < frames))
		return fail(POLAR_E_ARG, "polar_hip_decode_pitched: a row pitch below the row length");
	if ((rc = use_device(h)))
		return rc;
	if (!frames)
		return POLAR_OK;
	if ((rc = launch_decode(h, program_id, message, codeword, frames, h->stream, true, true, codeword_pitch, message_pitch)))
		return rc;
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

extern "C" size_t polar_hip_row_pitch(size_t frames)
{
	// rows a multiple of 4 KiB apart all start in the same place of the memory system's address interleaving
	return frames && frames % 4096 == 0 ? frames + 4352 : (frames + 15) / 16 * 16;
}

extern "C" int polar_hip_decode_async(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, void *stream)
{
	int rc = check_decode_args(h, program_id, message, codeword, frames);
	if (rc)
		return rc;
	if ((rc = use_device(h)))
		return rc;
	return launch_decode(h, program_id, message, codeword, frames, (hipStream_t)stream, false);
}

// No allocation on this path: fixed kernel shape (resident since polar_hip_load_program), arenas in
// the file-scope `scratch` array through the per-device pointer taken at polar_hip_create, no
// staging copies (so only whole groups and aligned rows), occupancy known since create.
extern "C" int polar_hip_decode_static(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, void *stream)
{
	int rc = check_decode_args(h, program_id, message, codeword, frames);
	if (rc)
		return rc;
	if ((rc = use_device(h)))
		return rc;
	if (!frames)
		return POLAR_OK;
	const size_t fpw = 4 * (size_t)STATIC_Q;
	LoadedProgram &lp = h->programs[program_id];
	// teams of four for whole teams whose arenas fit the array together, else single wavefronts
	const bool teams = frames % (fpw * STATIC_TEAM) == 0 &&
		(lp.host.top <= STATIC_RL || STATIC_SCRATCH_BYTES / (scratch_bytes_per_wave(STATIC_Q, lp.host.top) * STATIC_TEAM) >= 1);
	const KernelVariant *v = find_variant(STATIC_Q, STATIC_RL, teams ? STATIC_TEAM : 1);
	if (!v)
		return fail(POLAR_E_HIP, "polar_hip_decode_static: kernel shape not built into this library");
	const size_t team = (size_t)v->team;
	if (frames % fpw || (uintptr_t)codeword % 16 || (uintptr_t)message % 16)
		return fail(POLAR_E_ARG, "polar_hip_decode_static: frames must be a multiple of 32 and both buffers 16-byte aligned (this entry point cannot stage ragged batches: it does not allocate)");
	void *arena = h->device < MAX_DEVICES ? g_static_scratch[h->device] : nullptr;
	if (!arena)
		return fail(POLAR_E_HIP, "polar_hip_decode_static: no static scratch on this device");
	auto dev = lp.dev.find(program_shape(v));
	auto occ = h->occupancy.find((const void *)v->fn);
	if (dev == lp.dev.end() || occ == h->occupancy.end() || occ->second < 1)
		return fail(POLAR_E_ARG, "polar_hip_decode_static: program not resident for this entry point");
	const size_t groups = frames / fpw;
	if (groups > 0x7FFFFFFFu)
		return fail(POLAR_E_ARG, "too many frames for one call");
	size_t blocks = (size_t)occ->second * h->cus; // workgroups of `team` wavefronts
	DecodeArgs a = {};
	a.scratch = nullptr;
	a.block_dwords = 0;
	if (lp.host.top > v->rl) {
		const size_t per_wave = scratch_bytes_per_wave(v->q, lp.host.top);
		const size_t fit = STATIC_SCRATCH_BYTES / (per_wave * team);
		if (fit < 1)
			return fail(POLAR_E_ARG, "polar_hip_decode_static: one wavefront of this code needs " + std::to_string(per_wave) + " bytes of scratch, the static array holds " + std::to_string((size_t)STATIC_SCRATCH_BYTES) + " (codes up to N = 2^20)");
		if (blocks > fit)
			blocks = fit;
		a.scratch = (u32 *)arena; // the pointer, never the symbol
		a.block_dwords = per_wave / sizeof(u32);
	}
	if (blocks > groups / team)
		blocks = groups / team;
	a.prog = dev->second;
	a.cw = (const uint8_t *)codeword;
	a.msg = (uint8_t *)message;
	a.stride = frames;
	a.msg_stride = frames;
	a.groups = (u32)groups;
	a.top = lp.host.top;
	a.hot_top = v->rl; // no hot region here: it would have to be carved out of the static array
	a.hot = nullptr;
	a.hot_dwords = 0;
	a.prof = nullptr;
	h->timed = false;
	hipLaunchKernelGGL(v->fn, dim3((unsigned)blocks), dim3(64 * (unsigned)team), 0, (hipStream_t)stream, a);
	HIP_TRY(hipGetLastError());
	return POLAR_OK;
}

extern "C" int polar_hip_last_decode_shape(polar_hip_handle *h, int *frames_per_wave, int *lds_level, int *team)
{
	if (!h || !h->last_shape[0])
		return fail(POLAR_E_ARG, "polar_hip_last_decode_shape: no decode on this handle");
	if (frames_per_wave)
		*frames_per_wave = h->last_shape[0];
	if (lds_level)
		*lds_level = h->last_shape[1];
	if (team)
		*team = h->last_shape[2];
	return POLAR_OK;
}

extern "C" int polar_hip_last_decode_launches(polar_hip_handle *h)
{
	if (!h || !h->last_shape[0])
		return fail(POLAR_E_ARG, "polar_hip_last_decode_launches: no decode on this handle");
	return h->last_launches;
}

extern "C" float polar_hip_last_decode_ms(polar_hip_handle *h)
{
	if (!h || !h->timed)
		return (float)fail(POLAR_E_ARG, "polar_hip_last_decode_ms: no timed decode on this handle");
	float ms = 0;
	if (hipEventSynchronize(h->ev1) != hipSuccess || hipEventElapsedTime(&ms, h->ev0, h->ev1) != hipSuccess)
		return (float)fail(POLAR_E_HIP, "polar_hip_last_decode_ms: event query failed");
	return ms;
}

extern "C" int polar_hip_synchronize(polar_hip_handle *h)
{
	if (!h)
		return fail(POLAR_E_ARG, "polar_hip_synchronize: null handle");
	int rc = use_device(h);
	if (rc)
		return rc;
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}

// ---- float decoder ----

// enqueues the float decode of device buffers on the handle's stream (timed: polar_hip_last_decode_ms)
static int launch_decode_f32(polar_hip_handle *h, int program_id, float *msg, const float *cw, size_t frames)
{
	LoadedProgram &lp = h->programs[program_id];
	const size_t n = (size_t)1 << lp.host.top;
	const size_t groups = (frames + 63) / 64, per_wave = 3 * n * 64 * sizeof(float);
	if (groups > 0x7FFFFFFFu)
		return fail(POLAR_E_ARG, "too many frames for one call");
	if (!lp.dev_bytes) { // the float kernel interprets the reference byte-code as it is
		HIP_TRY(hipMalloc((void **)&lp.dev_bytes, lp.bytes.size()));
		HIP_TRY(hipMemcpy(lp.dev_bytes, lp.bytes.data(), lp.bytes.size(), hipMemcpyHostToDevice));
	}
	ScratchBudget budget{h};
	size_t blocks = (size_t)8 * h->cus < groups ? (size_t)8 * h->cus : groups;
	if (!budget.fits(blocks * per_wave)) {
		blocks = budget.limit() / per_wave;
		if (blocks < 1)
			return fail(POLAR_E_NOMEM, "not enough device memory for one wavefront's float arena");
	}
	int rc;
	if ((rc = h->scratch.reserve(blocks * per_wave)))
		return rc;
	FloatArgs a;
	a.prog = lp.dev_bytes;
	a.cw = cw;
	a.msg = msg;
	a.arena = (float *)h->scratch.p;
	a.frames = frames;
	a.groups = (u32)groups;
	a.top = lp.host.top;
	h->timed = false;
	h->placement.pending = false; // the timing events now belong to this launch
	HIP_TRY(hipEventRecord(h->ev0, h->stream));
	hipLaunchKernelGGL(sc_decode_f32_kernel, dim3((unsigned)blocks), dim3(64), 0, h->stream, a);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipEventRecord(h->ev1, h->stream));
	h->timed = true;
	return POLAR_OK;
}

extern "C" int polar_hip_decode_f32(polar_hip_handle *h, int program_id, float *message, const float *codeword, size_t frames, int where)
{
	int rc = check_decode_args(h, program_id, message, codeword, frames);
	if (rc)
		return rc;
	if (where != POLAR_HOST && where != POLAR_DEVICE)
		return fail(POLAR_E_ARG, "polar_hip_decode_f32: bad `where`");
	if ((rc = use_device(h)))
		return rc;
	if (!frames)
		return POLAR_OK;
	LoadedProgram &lp = h->programs[program_id];
	const size_t n = (size_t)1 << lp.host.top, rows = lp.host.rows;
	const float *cw = codeword;
	float *msg = message;
	if (where == POLAR_HOST) {
		if ((rc = h->stage_in.reserve(n * frames * sizeof(float))) || (rc = h->stage_out.reserve((rows ? rows : 1) * frames * sizeof(float))))
			return rc;
		HIP_TRY(hipMemcpyAsync(h->stage_in.p, codeword, n * frames * sizeof(float), hipMemcpyHostToDevice, h->stream));
		cw = (const float *)h->stage_in.p;
		msg = (float *)h->stage_out.p;
	}
	if ((rc = launch_decode_f32(h, program_id, msg, cw, frames)))
		return rc;
	if (where == POLAR_HOST && rows)
		HIP_TRY(hipMemcpyAsync(message, msg, rows * frames * sizeof(float), hipMemcpyDeviceToHost, h->stream));
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}
