// abi_set.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): decoding a set of resident batches of different codes in one launch.
#pragma once

// A batch of a large code holds few frames for its bytes and is bound by the serial depth of its program
// (8.9 ms for 8 192 frames of N = 2^16 whatever the kernel shape), so one such launch leaves most of the
// device idle. polar_hip_decode_set (per-SNR frozen sets, mixed N = 2^10 .. 2^16 batches) sorts
// its jobs in two: those that fill the device on their own (two sweeps and more of 128-frame groups) are
// launched one after the other as polar_hip_decode would; all the others share ONE launch of a team kernel
// ((64, 7) or (32, 8), see below): their teams form one queue, the largest codes first, and every resident
// workgroup takes the next team from it when it has finished one (decode_kernel.hpp), so the workgroups finish
// together whatever the mix. One launch on one queue: several launches on streams of their own overlap only as
// far as the streams happen to sit on different hardware queues (four per process here;
// profiles/r3_bench_E_streams.txt).
enum { SET_Q = 8, SET_RL = 8, SET_TEAM = 4 };

extern "C" int polar_hip_decode_set(polar_hip_handle *h, const polar_hip_job *jobs, int count)
{
	if (!h || count < 0 || (count && !jobs))
		return fail(POLAR_E_ARG, "polar_hip_decode_set: bad argument");
	int rc = use_device(h);
	if (rc)
		return rc;
	for (int i = 0; i < count; ++i) {
		if ((rc = check_decode_args(h, jobs[i].program_id, jobs[i].message, jobs[i].codeword, jobs[i].frames)))
			return rc;
		if (jobs[i].frames && (polar_hip_where(jobs[i].codeword) != POLAR_DEVICE || (h->programs[jobs[i].program_id].host.rows && polar_hip_where(jobs[i].message) != POLAR_DEVICE)))
			return fail(POLAR_E_ARG, "polar_hip_decode_set: the batches of a set are device-resident (device pointers only)");
		if ((jobs[i].codeword_pitch && jobs[i].codeword_pitch < jobs[i].frames) || (jobs[i].message_pitch && jobs[i].message_pitch < jobs[i].frames))
			return fail(POLAR_E_ARG, "polar_hip_decode_set: a row pitch below the row length");
	}
	// the common launch: (64, 7) teams when the jobs that will share it bring at least one team per resident
	// workgroup (8 % faster than (32, 8) teams under load: config E 118 -> 127 Gbit/s), else (32, 8) teams (more,
	// narrower teams for a set that is short of frames altogether)
	int set_q = SET_Q, set_rl = SET_RL;
	{
		size_t wide_teams = 0;
		bool whole = true;
		for (int i = 0; i < count; ++i) {
			if (!jobs[i].frames || h->cfg_frames_per_wave)
				continue;
			if (jobs[i].frames / 128 >= (size_t)16 * h->cus)
				continue; // fills the device on its own
			whole = whole && jobs[i].frames % 256 == 0;
			wide_teams += jobs[i].frames / 256;
		}
		if (whole && wide_teams >= (size_t)2 * h->cus && find_variant(16, 7, SET_TEAM)) {
			set_q = 16;
			set_rl = 7;
		}
	}
	if (const char *e = getenv("POLAR_HIP_SET_SHAPE")) // experiments: "8,8" or "16,7"
		sscanf(e, "%d,%d", &set_q, &set_rl);
	const KernelVariant *v = find_variant(set_q, set_rl, SET_TEAM);
	std::vector<int> shared; // jobs of the common launch
	const size_t fpw = 4 * (size_t)set_q * SET_TEAM;
	for (int i = 0; i < count; ++i) {
		if (!jobs[i].frames)
			continue;
		const LoadedProgram &lp = h->programs[jobs[i].program_id];
		const bool direct = jobs[i].frames % fpw == 0 && (uintptr_t)jobs[i].codeword % 16 == 0 && (uintptr_t)jobs[i].message % 16 == 0 && jobs[i].codeword_pitch % 16 == 0 && jobs[i].message_pitch % 16 == 0;
		// a configured shape, or two sweeps and more of the widest one: the job is not short of frames
		const bool fills = h->cfg_frames_per_wave || jobs[i].frames / 128 >= (size_t)16 * h->cus;
		if (v && direct && !fills && lp.host.top > set_rl)
			shared.push_back(i);
		else if ((rc = launch_decode(h, jobs[i].program_id, jobs[i].message, jobs[i].codeword, jobs[i].frames, h->stream, false, false, jobs[i].codeword_pitch, jobs[i].message_pitch)))
			return rc;
	}
	if (shared.size() == 1) { // nothing to share a launch with
		int i = shared[0];
		if ((rc = launch_decode(h, jobs[i].program_id, jobs[i].message, jobs[i].codeword, jobs[i].frames, h->stream, false, false, jobs[i].codeword_pitch, jobs[i].message_pitch)))
			return rc;
		shared.clear();
	}
	if (!shared.empty()) {
		int per_cu = 0;
		auto occ = h->occupancy.find((const void *)v->fn);
		if (occ == h->occupancy.end()) {
			HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, v->fn, 64 * v->team, 0));
			h->occupancy[(const void *)v->fn] = per_cu;
		} else {
			per_cu = occ->second;
		}
		if (per_cu < 1)
			return fail(POLAR_E_HIP, "decode kernel does not fit on a CU");
		size_t resident = (size_t)per_cu * h->cus;
		if (h->cfg_max_waves && (size_t)h->cfg_max_waves / SET_TEAM < resident)
			resident = (size_t)h->cfg_max_waves / SET_TEAM ? (size_t)h->cfg_max_waves / SET_TEAM : 1;
		// the common queue: the jobs with the longest walks (largest codes) first, so that the last teams taken are short ones
		std::sort(shared.begin(), shared.end(), [&](int x, int y) {
			int tx = h->programs[jobs[x].program_id].host.top, ty = h->programs[jobs[y].program_id].host.top;
			return tx != ty ? tx > ty : x < y;
		});
		const size_t m = shared.size();
		std::vector<DecodeArgs> table(m + 1);
		size_t first = 0, block_dwords = 0;
		for (size_t k = 0; k < m; ++k) {
			LoadedProgram &lp = h->programs[jobs[shared[k]].program_id];
			if ((rc = upload_shape(lp, v)))
				return rc;
			const size_t soft = (size_t)set_q << lp.host.top;
			if (soft + 2 * (soft / 8) > block_dwords)
				block_dwords = soft + 2 * (soft / 8); // every workgroup's arenas hold the largest code of the set
			DecodeArgs &t = table[k];
			t = DecodeArgs{};
			t.prog = lp.dev[program_shape(v)];
			t.cw = (const uint8_t *)jobs[shared[k]].codeword;
			t.msg = (uint8_t *)jobs[shared[k]].message;
			t.stride = jobs[shared[k]].codeword_pitch ? jobs[shared[k]].codeword_pitch : jobs[shared[k]].frames;
			t.msg_stride = jobs[shared[k]].message_pitch ? jobs[shared[k]].message_pitch : jobs[shared[k]].frames;
			t.top = lp.host.top;
			t.set_first = (u32)first;
			first += jobs[shared[k]].frames / fpw;
		}
		table[m] = DecodeArgs{};
		table[m].set_first = (u32)first;
		if (first > 0x7FFFFFFFu)
			return fail(POLAR_E_ARG, "polar_hip_decode_set: too many frames for one call");
		const size_t grid = first < resident ? first : resident;
		const size_t table_bytes = (table.size() * sizeof(DecodeArgs) + 255) / 256 * 256;
		HIP_TRY(hipStreamSynchronize(h->stream)); // the jobs launched above may still use the arenas that the reserve replaces
		if ((rc = h->scratch.reserve_arenas(grid * SET_TEAM * block_dwords * sizeof(u32), h->device)) || (rc = h->set_table.reserve(table_bytes + 256)))
			return rc;
		u32 *counter = (u32 *)((char *)h->set_table.p + table_bytes);
		HIP_TRY(hipMemcpyAsync(h->set_table.p, table.data(), table.size() * sizeof(DecodeArgs), hipMemcpyHostToDevice, h->stream));
		HIP_TRY(hipMemsetAsync(counter, 0, sizeof(u32), h->stream));
		HIP_TRY(hipStreamSynchronize(h->stream)); // the table is a local
		if (getenv("POLAR_HIP_DEBUG"))
			fprintf(stderr, "polar_hip: decode set: %zu of %d jobs share one launch: %zu teams for %zu workgroups (Q=%d)\n", m, count, first, grid, set_q);
		DecodeArgs a = {};
		a.set = (const DecodeArgs *)h->set_table.p;
		a.set_jobs = (u32)m;
		a.set_counter = counter;
		a.scratch = (u32 *)h->scratch.p;
		a.block_dwords = block_dwords;
		a.top = table[0].top;
		a.hot_top = set_rl;
		h->placement.pending = false;
		h->timed = false;
		h->last_shape[0] = 4 * v->q;
		h->last_shape[1] = v->rl;
		h->last_shape[2] = v->team;
		h->last_launches = 1;
		hipLaunchKernelGGL(v->fn, dim3((unsigned)grid), dim3(64 * (unsigned)v->team), 0, h->stream, a);
		HIP_TRY(hipGetLastError());
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}
