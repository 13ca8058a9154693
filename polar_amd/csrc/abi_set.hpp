// abi_set.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): decoding a set of resident batches of different codes in one launch.
#pragma once

// A batch of a large code holds few frames for its bytes and is bound by the serial depth of its program
// (8.9 ms for 8 192 frames of N = 2^16 whatever the kernel shape), so one such launch leaves most of the
// device idle. polar_hip_decode_set (per-SNR frozen sets, mixed N = 2^10 .. 2^16 batches) sorts
// its jobs in two: those that fill the device on their own (two sweeps and more of 128-frame groups) are
// launched one after the other as polar_hip_decode would; all the others share ONE launch of a team kernel
// ((64, 7) or (32, 8), see below), whose workgroups are dealt to the jobs so that all of them finish together: a job of
// `units` teams that gets `blocks` workgroups takes ceil(units / blocks) walks of its program, a walk is
// about N / 16 device operations of 2 us, and the smallest common finishing time for which the workgroups
// add up to the resident ones is found by bisection. One launch on one queue: several launches on streams
// of their own overlap only as far as the streams happen to sit on different hardware queues (four per
// process here; profiles/r3_bench_E_streams.txt).
enum { SET_Q = 8, SET_RL = 8, SET_TEAM = 4 };

static double set_walk_ms(int top) { return (double)((size_t)1 << top) * 1.25e-4; }

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
		const bool direct = jobs[i].frames % fpw == 0 && (uintptr_t)jobs[i].codeword % 16 == 0 && (uintptr_t)jobs[i].message % 16 == 0;
		// a configured shape, or two sweeps and more of the widest one: the job is not short of frames
		const bool fills = h->cfg_frames_per_wave || jobs[i].frames / 128 >= (size_t)16 * h->cus;
		if (v && direct && !fills && lp.host.top > set_rl)
			shared.push_back(i);
		else if ((rc = launch_decode(h, jobs[i].program_id, jobs[i].message, jobs[i].codeword, jobs[i].frames, h->stream, false)))
			return rc;
	}
	if (shared.size() == 1) { // nothing to share a launch with
		int i = shared[0];
		if ((rc = launch_decode(h, jobs[i].program_id, jobs[i].message, jobs[i].codeword, jobs[i].frames, h->stream, false)))
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
		const size_t m = shared.size();
		std::vector<size_t> units(m), blocks(m);
		std::vector<double> walk(m);
		for (size_t k = 0; k < m; ++k) {
			units[k] = jobs[shared[k]].frames / fpw;
			walk[k] = set_walk_ms(h->programs[jobs[shared[k]].program_id].host.top);
		}
		// blocks for a common finishing time T: every job takes floor(T / walk) walks at most
		auto need = [&](double T, std::vector<size_t> *out) {
			size_t sum = 0;
			for (size_t k = 0; k < m; ++k) {
				size_t walks = (size_t)(T / walk[k]);
				if (!walks)
					return (size_t)-1;
				size_t b = (units[k] + walks - 1) / walks;
				if (out)
					(*out)[k] = b;
				sum += b;
			}
			return sum;
		};
		double lo = 0, hi = 0;
		for (size_t k = 0; k < m; ++k) {
			lo = walk[k] > lo ? walk[k] : lo;
			hi += walk[k] * (double)units[k];
		}
		if (need(lo, nullptr) > resident) {
			for (int it = 0; it < 60; ++it) {
				double mid = 0.5 * (lo + hi);
				(need(mid, nullptr) <= resident ? hi : lo) = mid;
			}
			lo = hi;
		}
		need(lo, &blocks);
		// what is left over goes to the jobs with the most walks per workgroup
		size_t used = 0;
		for (size_t b : blocks)
			used += b;
		// arenas and the argument table
		std::vector<DecodeArgs> table(m + 1);
		size_t arena_dwords = 0, first = 0;
		for (size_t k = 0; k < m; ++k) {
			LoadedProgram &lp = h->programs[jobs[shared[k]].program_id];
			if ((rc = upload_shape(lp, v)))
				return rc;
			const size_t soft = (size_t)set_q << lp.host.top, block_dwords = soft + 2 * (soft / 8);
			DecodeArgs &t = table[k];
			t = DecodeArgs{};
			t.prog = lp.dev[program_shape(v)];
			t.cw = (const uint8_t *)jobs[shared[k]].codeword;
			t.msg = (uint8_t *)jobs[shared[k]].message;
			t.block_dwords = block_dwords;
			t.stride = jobs[shared[k]].frames;
			t.groups = (u32)(units[k] * SET_TEAM);
			t.top = lp.host.top;
			t.hot_top = set_rl;
			t.set_first = (u32)first;
			t.scratch = (u32 *)(uintptr_t)arena_dwords; // offset for now: the base is known after the reserve
			arena_dwords += blocks[k] * SET_TEAM * block_dwords;
			first += blocks[k];
		}
		table[m] = DecodeArgs{};
		table[m].set_first = (u32)first;
		HIP_TRY(hipStreamSynchronize(h->stream)); // the jobs launched above may still use the arenas that the reserve replaces
		if ((rc = h->scratch.reserve_arenas(arena_dwords * sizeof(u32), h->device)) || (rc = h->set_table.reserve(table.size() * sizeof(DecodeArgs))))
			return rc;
		for (size_t k = 0; k < m; ++k)
			table[k].scratch = (u32 *)h->scratch.p + (size_t)(uintptr_t)table[k].scratch;
		HIP_TRY(hipMemcpyAsync(h->set_table.p, table.data(), table.size() * sizeof(DecodeArgs), hipMemcpyHostToDevice, h->stream));
		HIP_TRY(hipStreamSynchronize(h->stream)); // the table is a local
		if (getenv("POLAR_HIP_DEBUG"))
			fprintf(stderr, "polar_hip: decode set: %zu of %d jobs share one launch of %zu workgroups (%zu resident), common time %.2f ms by the model\n", m, count, first, resident, lo);
		DecodeArgs a = {};
		a.set = (const DecodeArgs *)h->set_table.p;
		a.set_jobs = (u32)m;
		h->placement.pending = false;
		h->timed = false;
		h->last_shape[0] = 4 * v->q;
		h->last_shape[1] = v->rl;
		h->last_shape[2] = v->team;
		hipLaunchKernelGGL(v->fn, dim3((unsigned)first), dim3(64 * (unsigned)v->team), 0, h->stream, a);
		HIP_TRY(hipGetLastError());
		(void)used;
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}
