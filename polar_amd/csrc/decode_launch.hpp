// decode_launch.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): launch of the decode kernel: shape, arenas, hot levels, staging of ragged batches.
#pragma once

// Where the driver puts the arenas decides up to 5 % of the kernel time: the same 20 GiB allocation, at the
// same virtual address, gives 219 or 231 ms for 2^20 frames of N = 2^16 depending on the physical memory
// behind it, reproducibly per placement (profiles/r3_placement.txt). Nothing in user space tells a good
// placement from a bad one except the clock, so a handle that decodes the same resident batch again and
// again (polar_hip_decode with device pointers: the bench loop) measures: the first call runs on the
// placement at hand; it is then set aside and the next call runs on newly allocated arenas; the faster of
// the two stays, the other is freed; PLACEMENT_CANDIDATES new placements are looked at this way, one call
// each, the best one is kept. Results do not depend on it. Arenas of 1 GiB and more, and only while a
// second copy fits the free device memory; POLAR_HIP_PLACEMENT=0 turns it off.
enum { PLACEMENT_CANDIDATES = 3 };
static void placement_step(polar_hip_handle *h, int program_id, size_t frames, const KernelVariant *v, size_t bytes, bool relay_possible)
{
	typedef polar_hip_handle::Placement P;
	static const bool off = getenv("POLAR_HIP_PLACEMENT") && atoi(getenv("POLAR_HIP_PLACEMENT")) == 0;
	polar_hip_handle::Placement &t = h->placement;
	const bool same = t.program == program_id && t.frames == frames && t.fn == (const void *)v->fn && t.bytes == bytes;
	if (off || bytes < ((size_t)1 << 30) || !same) { // nothing to watch, or another workload: start over with the placement at hand
		if (h->scratch_spare.p) { // a comparison was under way: the arenas set aside were the better ones so far
			if (hipStreamSynchronize(h->stream) == hipSuccess) {
				h->scratch.release();
				std::swap(h->scratch, h->scratch_spare);
			}
		}
		t = polar_hip_handle::Placement();
		if (!off && bytes >= ((size_t)1 << 30)) {
			t.program = program_id;
			t.frames = frames;
			t.fn = (const void *)v->fn;
			t.bytes = bytes;
		}
		return;
	}
	if (t.done) { // the placement is chosen; what is left is the call that ran as relay launches, if there was one
		float trial = 0;
		if (t.relay == P::RELAY_TRIAL && t.pending) {
			if (hipEventSynchronize(h->ev1) == hipSuccess && hipEventElapsedTime(&trial, h->ev0, h->ev1) == hipSuccess && trial > 0)
				t.relay = trial < 0.99f * t.best_ms ? P::RELAY_ON : P::RELAY_OFF; // it has to gain a percent
			else
				t.relay = P::RELAY_OFF;
			if (getenv("POLAR_HIP_DEBUG"))
				fprintf(stderr, "polar_hip: relay launches %.2f ms against %.2f ms for one launch: %s\n", trial, t.best_ms, t.relay == P::RELAY_ON ? "kept" : "dropped");
		}
		t.pending = false;
		return;
	}
	if (!t.pending)
		return;
	t.pending = false;
	struct Finish { // whichever way the comparison of placements ends: the next call tries relay launches
		P &t;
		bool relay_possible;
		~Finish()
		{
			if (t.done && relay_possible && t.relay == P::RELAY_UNTESTED)
				t.relay = P::RELAY_TRIAL;
		}
	} finish{t, relay_possible};
	float ms = 0;
	bool ok = hipEventSynchronize(h->ev1) == hipSuccess && hipEventElapsedTime(&ms, h->ev0, h->ev1) == hipSuccess && ms > 0;
	ok = ok && hipStreamSynchronize(h->stream) == hipSuccess;
	const bool debug = getenv("POLAR_HIP_DEBUG") != nullptr;
	if (ok && h->scratch_spare.p) { // the call ran on a candidate: keep the faster of it and the one set aside
		if (ms < t.best_ms) {
			t.best_ms = ms;
			h->scratch_spare.release();
		} else {
			h->scratch.release();
			std::swap(h->scratch, h->scratch_spare);
		}
	} else if (ok) {
		t.best_ms = ms;
	}
	if (debug)
		fprintf(stderr, "polar_hip: arena placement %d: %.2f ms, best %.2f ms\n", t.tries, ms, t.best_ms);
	size_t free_b = 0, total_b = 0;
	if (!ok || t.tries >= PLACEMENT_CANDIDATES || hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)4 << 30)) {
		t.done = true;
		return;
	}
	++t.tries;
	std::swap(h->scratch, h->scratch_spare); // set the best aside; the reserve that follows allocates the candidate
	if (h->scratch.reserve_arenas(bytes, h->device) != POLAR_OK) { // no room after all: go on with what we have
		(void)hipGetLastError();
		std::swap(h->scratch, h->scratch_spare);
		t.done = true;
	}
}

// Relay launches need kernels on `stream` and on the handle's second stream to run side by side. Streams share
// the process's few hardware queues (four here) by an assignment nobody can ask for, and two streams on one queue
// run their kernels one after the other - half a grid at a time, for the relay: much slower than one launch. So
// the pair is tried once per caller's stream: one wavefront spins for a fraction of a millisecond, then one on each
// stream; when the two took twice as long the second stream is replaced by a newly created one (consecutive streams go to different queues), four
// tries; if none runs beside `stream`, relay launches stay off for it. Synchronises `stream` (once).
__global__ void relay_probe_kernel(unsigned long long ticks)
{
	const unsigned long long t0 = __builtin_amdgcn_s_memtime(); // constant-rate counter: the loop ends whatever the clocks do
	while (__builtin_amdgcn_s_memtime() - t0 < ticks)
		__builtin_amdgcn_s_sleep(8);
}

static int relay_ready(polar_hip_handle *h, hipStream_t stream, bool *usable)
{
	if (h->relay_partner_known && h->relay_partner == stream) {
		*usable = h->relay_usable;
		return POLAR_OK;
	}
	if (!h->relay_fork) {
		HIP_TRY(hipEventCreateWithFlags(&h->relay_fork, hipEventDisableTiming));
		HIP_TRY(hipEventCreateWithFlags(&h->relay_join, hipEventDisableTiming));
	}
	const unsigned long long spin = 500000; // 0.2 - 0.5 ms of the shader clock's counter
	auto probe = [&](bool both, float *ms) -> int {
		HIP_TRY(hipEventRecord(h->ev0, stream));
		HIP_TRY(hipEventRecord(h->relay_fork, stream));
		HIP_TRY(hipStreamWaitEvent(h->relay, h->relay_fork, 0));
		hipLaunchKernelGGL(relay_probe_kernel, dim3(1), dim3(64), 0, stream, spin);
		if (both)
			hipLaunchKernelGGL(relay_probe_kernel, dim3(1), dim3(64), 0, h->relay, spin);
		HIP_TRY(hipGetLastError());
		HIP_TRY(hipEventRecord(h->relay_join, h->relay));
		HIP_TRY(hipStreamWaitEvent(stream, h->relay_join, 0));
		HIP_TRY(hipEventRecord(h->ev1, stream));
		HIP_TRY(hipEventSynchronize(h->ev1));
		HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
		return POLAR_OK;
	};
	h->relay_usable = false;
	std::vector<hipStream_t> rejected; // kept until the search is over: a stream created while they live goes to another queue
	struct Drop {
		std::vector<hipStream_t> &v;
		~Drop()
		{
			for (hipStream_t s : v) {
				(void)hipStreamSynchronize(s);
				(void)hipStreamDestroy(s);
			}
		}
	} drop{rejected};
	for (int attempt = 0; attempt < 4 && !h->relay_usable; ++attempt) {
		if (!h->relay || attempt) {
			if (h->relay)
				rejected.push_back(h->relay);
			h->relay = nullptr;
			HIP_TRY(hipStreamCreateWithFlags(&h->relay, hipStreamNonBlocking));
		}
		float one = 0, two = 0, warm = 0;
		int rc;
		if ((rc = probe(true, &warm)) || (rc = probe(false, &one)) || (rc = probe(true, &two))) // the first one loads the kernel
			return rc;
		h->relay_usable = two < 1.5f * one;
		if (getenv("POLAR_HIP_DEBUG"))
			fprintf(stderr, "polar_hip: relay stream %d: one spinning wavefront %.3f ms, one on each stream %.3f ms: %s\n", attempt, one, two, h->relay_usable ? "side by side" : "one after the other");
	}
	h->timed = false;
	h->relay_partner = stream;
	h->relay_partner_known = true;
	*usable = h->relay_usable;
	return POLAR_OK;
}

// Enqueues the decode of `frames` frames on `stream`. The kernel wants whole groups of 4Q frames
// and dword-aligned rows; anything else is staged through padded device copies (zero-filled
// tail frames decode to garbage that is never copied back).
// cw_pitch / msg_pitch: bytes from one row of the caller's buffer to the next (0: the rows are contiguous, `frames` apart).
static int launch_decode(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, hipStream_t stream, bool timed, bool tune_placement = false, size_t cw_pitch = 0, size_t msg_pitch = 0)
{
	if (!cw_pitch)
		cw_pitch = frames;
	if (!msg_pitch)
		msg_pitch = frames;
	LoadedProgram &lp = h->programs[program_id];
	ScratchBudget budget{h};
	const KernelVariant *v = pick_variant(h, lp.host.top, frames, budget);
	if (!v)
		return fail(POLAR_E_ARG, "no kernel variant for the configured (frames_per_wave, lds_level)");
	const int shape = program_shape(v);
	int rc;
	if ((rc = upload_shape(lp, v))) // first use with this wavefront shape: fuse and upload
		return rc;
	// a workgroup takes `team` neighbouring groups at a time (1, or TEAM_WAVES in the team form)
	const size_t team = (size_t)v->team, fpw = 4 * (size_t)v->q * team, n = (size_t)1 << lp.host.top, rows = lp.host.rows;
	const size_t units = (frames + fpw - 1) / fpw, groups = units * team;
	if (groups > 0x7FFFFFFFu)
		return fail(POLAR_E_ARG, "too many frames for one call");
	h->timed = false;
	if (!groups)
		return POLAR_OK;
	const size_t padded = units * fpw;
	const bool direct = padded == frames && (uintptr_t)codeword % 16 == 0 && (uintptr_t)message % 16 == 0 && cw_pitch % 16 == 0 && msg_pitch % 16 == 0;
	const uint8_t *cw = (const uint8_t *)codeword;
	uint8_t *msg = (uint8_t *)message;
	if (!direct) {
		if ((rc = h->pad_in.reserve(n * padded)) || (rc = h->pad_out.reserve((rows ? rows : 1) * padded)))
			return rc;
		HIP_TRY(hipMemsetAsync(h->pad_in.p, 0, n * padded, stream));
		HIP_TRY(hipMemcpy2DAsync(h->pad_in.p, padded, codeword, cw_pitch, frames, n, hipMemcpyDeviceToDevice, stream));
		cw = (const uint8_t *)h->pad_in.p;
		msg = (uint8_t *)h->pad_out.p;
	}
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
	size_t blocks = (size_t)per_cu * h->cus;
	if (h->cfg_max_waves && (size_t)h->cfg_max_waves / team < blocks)
		blocks = (size_t)h->cfg_max_waves / team ? (size_t)h->cfg_max_waves / team : 1;
	size_t block_dwords = 0, max_blocks = ~(size_t)0 >> 8;
	if (lp.host.top > v->rl) { // LLR levels above the LDS levels and the partial-sum planes live in scratch
		size_t soft = (size_t)v->q << lp.host.top;
		block_dwords = soft + 2 * (soft / 8);
		if (const char *e = getenv("POLAR_HIP_ARENA_PAD")) // experiments: dwords between the arenas (a multiple of 4)
			block_dwords += (size_t)atoi(e) / 4 * 4;
		if (!budget.fits(blocks * team * block_dwords * sizeof(u32))) { // fewer resident wavefronts, more sweeps
			max_blocks = budget.limit() / (team * block_dwords * sizeof(u32));
			if (max_blocks < 1)
				return fail(POLAR_E_NOMEM, "not enough device memory for one wavefront's scratch arena");
			if (blocks > max_blocks)
				blocks = max_blocks;
		}
	}
	// every wavefront walks the same number of groups: no idle tail behind a last partial sweep
	size_t sweeps = (units + blocks - 1) / blocks;
	blocks = (units + sweeps - 1) / sweeps;
	if (blocks >= 8 && blocks % 8) // multiple of 8 for the XCD-aware group order
		blocks = blocks + 8 - blocks % 8 <= max_blocks ? blocks + 8 - blocks % 8 : blocks - blocks % 8;
	// Relay launches. The SIMD favours the older of its two wavefronts: in one launch of s sweeps the wavefronts
	// of the first half of the grid finish 11 % before the others and their places stay empty for the rest of
	// the kernel (profiles/r3_wave_lifetimes.txt). So a batch of two sweeps and more goes out as two cohorts of
	// half a grid each - one wavefront per SIMD - on two streams, the caller's and a second one that is forked
	// from it and joined to it again, each with its own half of the arenas. Per segment of about four sweeps one
	// cohort is sent as two launches and the other as one: the cohort that was launched first is the older and
	// faster one on every SIMD until its first launch is through; its second launch is then the younger one,
	// and the other cohort runs the second half of its work at the faster pace. Both finish together (496 instead
	// of 512 in units of 1/120 of a walk; measured 212 instead of 219 ms at the headline). From segment to
	// segment the cohorts change roles. Where the two streams share a hardware queue the launches run one after
	// the other on half the device each, which is slower than one launch: the pair of streams is tried first (relay_ready). Not with a configured share of
	// the device (max_waves: the lanes of a caller who overlaps launches himself), not for ragged batches.
	size_t segments = 0; // 0: one launch
	const size_t half = blocks / 2 / 8 * 8; // workgroups per cohort
	bool relay_possible = false, relay_forced = false;
#if !defined(POLAR_PROFILE) && !defined(POLAR_SPAN)
	{
		const char *relay_env = getenv("POLAR_HIP_RELAY"); // 0 / 1: never; 2: whenever possible; unset: where the clock says so (placement_step)
		relay_possible = (!relay_env || atoi(relay_env) > 1) && direct && sweeps >= 2 && !h->cfg_max_waves && half && blocks >= (size_t)per_cu * h->cus / 2;
		relay_forced = relay_possible && relay_env;
	}
#endif
	if (getenv("POLAR_HIP_DEBUG"))
		fprintf(stderr, "polar_hip: scratch %p (%zu MiB) hot %p (%zu MiB)\n", h->scratch.p, h->scratch.bytes >> 20, h->hot.p, h->hot.bytes >> 20);
	if (getenv("POLAR_HIP_DEBUG"))
		fprintf(stderr, "polar_hip: decode Q=%d RL=%d team=%d groups=%zu blocks=%zu (%d per CU x %d CUs) lds=%zu B direct=%d hot levels <= %d\n", v->q, v->rl, v->team, groups, blocks, per_cu, h->cus, v->lds_bytes, (int)direct, hot_top_for(v, lp.host.top, blocks * team));
	h->last_shape[0] = 4 * v->q;
	h->last_shape[1] = v->rl;
	h->last_shape[2] = v->team;
	DecodeArgs a = {};
	a.prog = lp.dev[shape];
	a.cw = cw;
	a.msg = msg;
	a.stride = direct ? cw_pitch : padded;
	a.msg_stride = direct ? msg_pitch : padded;
	// experiments: the caller's rows are this many bytes apart (a multiple of 16); _IN / _OUT: one side only
	for (int side = 0; side < 3; ++side)
		if (const char *e = getenv(side == 0 ? "POLAR_HIP_ROW_STRIDE" : side == 1 ? "POLAR_HIP_ROW_STRIDE_IN" : "POLAR_HIP_ROW_STRIDE_OUT"))
			if (direct && cw_pitch == frames && msg_pitch == frames && (size_t)atol(e) >= padded && atol(e) % 16 == 0) {
				if (side != 2)
					a.stride = (size_t)atol(e);
				if (side != 1)
					a.msg_stride = (size_t)atol(e);
			}
	a.groups = (u32)groups;
	a.top = lp.host.top;
	a.hot_top = v->rl;
	a.hot = nullptr;
	a.hot_dwords = 0;
	a.scratch = nullptr;
	a.block_dwords = 0;
	if (block_dwords) {
		a.block_dwords = block_dwords;
		// experiments: byte offsets (multiples of 256, up to 64 MiB) of the arenas / the hot region inside their allocations
		const char *skew_e = getenv("POLAR_HIP_SCRATCH_SKEW"), *hskew_e = getenv("POLAR_HIP_HOT_SKEW");
		const size_t slack = skew_e || hskew_e ? (size_t)64 << 20 : 0;
		const size_t skew = skew_e ? (size_t)atol(skew_e) / 256 * 256 % (slack + 1) : 0, hskew = hskew_e ? (size_t)atol(hskew_e) / 256 * 256 % (slack + 1) : 0;
		if (const char *e = getenv("POLAR_HIP_REALLOC")) { // experiments: free and allocate anew: "scratch", "hot" or "both"
			HIP_TRY(hipStreamSynchronize(stream));
			if (e[0] == 's' || e[0] == 'b')
				h->scratch.release();
			if (e[0] == 'h' || e[0] == 'b')
				h->hot.release();
		}
		if (tune_placement && timed && direct && stream == h->stream)
			placement_step(h, program_id, frames, v, blocks * team * block_dwords * sizeof(u32) + slack, relay_possible);
		else
			h->placement.pending = false;
		// experiments: POLAR_HIP_ARENA_PIECES = "<MiB per piece>[s]": the arenas as separately allocated pieces of
		// device memory mapped into one range, in a shuffled order with an 's', the range aligned to the piece size with an 'a'
		if (const char *pe = getenv("POLAR_HIP_ARENA_PIECES")) {
			if (atoi(pe) > 0 && !strchr(pe, 'd') && (rc = h->scratch.reserve_chunked(blocks * team * block_dwords * sizeof(u32) + slack, (size_t)atoi(pe) << 20, strchr(pe, 's') != nullptr, h->device, strchr(pe, 'a') != nullptr)))
				return rc;
		}
		if ((rc = h->scratch.reserve_arenas(blocks * team * block_dwords * sizeof(u32) + slack, h->device)))
			return rc;
		a.scratch = (u32 *)((char *)h->scratch.p + skew);
		a.hot_top = hot_top_for(v, lp.host.top, blocks * team);
		if (a.hot_top > v->rl) { // the wavefronts' hot levels side by side in one compact region
			a.hot_dwords = ((size_t)v->q << (a.hot_top + 1)) - ((size_t)v->q << (v->rl + 1));
			if ((rc = h->hot.reserve(blocks * team * a.hot_dwords * sizeof(u32) + slack)))
				return rc;
			a.hot = (u32 *)((char *)h->hot.p + hskew);
		}
	}
	// Relay launches pay for some workloads and cost for others (N = 2^16: -2.3 % with 2^20 frames, +2.5 % with 2^21,
	// +12 % with 3 * 2^18; N = 2^12 ... 2^15: +1 to +6 %, profiles/r3_relay_sizes.txt), so they are not a default
	// but a measurement: a handle that decodes the same resident batch again and again (the one whose arena
	// placement is chosen by the clock) runs one call as relay launches once the placement stands, and keeps them
	// if that call was a percent faster than the best single launch.
	{
		typedef polar_hip_handle::Placement P;
		P &t = h->placement;
		const bool watched = tune_placement && timed && direct && stream == h->stream && t.program == program_id && t.done;
		if (relay_forced || (relay_possible && watched && (t.relay == P::RELAY_TRIAL || t.relay == P::RELAY_ON))) {
			bool usable = false;
			if ((rc = relay_ready(h, stream, &usable)))
				return rc;
			if (usable)
				segments = (sweeps + 3) / 4;
			else if (!relay_forced)
				t.relay = P::RELAY_OFF;
		}
		if (getenv("POLAR_HIP_DEBUG") && segments)
			fprintf(stderr, "polar_hip: relay launches: %zu segment(s) of three launches, %zu workgroups per cohort\n", segments, half);
		h->last_launches = segments ? (int)(3 * segments) : 1;
	}
	a.prof = nullptr;
#if defined(POLAR_PROFILE) || defined(POLAR_SPAN)
	static unsigned long long *prof_dev = nullptr;
	if (!prof_dev)
		HIP_TRY(hipMalloc((void **)&prof_dev, (16 * 32 * 2 + 2 * 4096) * 8));
	HIP_TRY(hipMemsetAsync(prof_dev, 0, (16 * 32 * 2 + 2 * 4096) * 8, stream));
	a.prof = prof_dev;
#endif
	if (timed)
		HIP_TRY(hipEventRecord(h->ev0, stream));
	if (segments) {
		HIP_TRY(hipEventRecord(h->relay_fork, stream));
		HIP_TRY(hipStreamWaitEvent(h->relay, h->relay_fork, 0));
		// `count` units from unit `first` on as one launch of half a grid: the cohort on the second stream uses the arenas from `half` on
		auto go = [&](size_t first, size_t count, bool on_relay) -> int {
			if (!count)
				return POLAR_OK;
			DecodeArgs ap = a;
			ap.cw = a.cw + first * fpw;
			ap.msg = a.msg + first * fpw;
			ap.groups = (u32)(count * team);
			if (on_relay) {
				if (ap.scratch)
					ap.scratch += half * team * block_dwords;
				if (ap.hot)
					ap.hot += half * team * a.hot_dwords;
			}
			hipLaunchKernelGGL(v->fn, dim3((unsigned)half), dim3(64 * (unsigned)team), 0, on_relay ? h->relay : stream, ap);
			HIP_TRY(hipGetLastError());
			return POLAR_OK;
		};
		// a segment's units in whole walks of half a grid: cohort Y takes s walks per wavefront in one launch, cohort X
		// the rest as floor(s / 2) walks in its first launch and what is left in its second
		const size_t seg_units = (units + segments - 1) / segments;
		for (size_t seg = 0; seg < segments && seg * seg_units < units; ++seg) {
			const size_t u0 = seg * seg_units, n = (u0 + seg_units < units ? u0 + seg_units : units) - u0;
			const size_t walks = (n + 2 * half - 1) / (2 * half);
			const size_t once = n < walks * half ? n : walks * half, twice = n - once;
			const size_t first_part = twice < walks / 2 * half ? twice : walks / 2 * half, second_part = twice - first_part;
			const bool twice_on_relay = seg & 1; // the cohort that goes out as two launches: launched first, so the older one to begin with
			if ((rc = go(u0, first_part, twice_on_relay)) || (rc = go(u0 + twice, once, !twice_on_relay)) || (rc = go(u0 + first_part, second_part, twice_on_relay)))
				return rc;
		}
		HIP_TRY(hipEventRecord(h->relay_join, h->relay));
		HIP_TRY(hipStreamWaitEvent(stream, h->relay_join, 0));
	} else {
		hipLaunchKernelGGL(v->fn, dim3((unsigned)blocks), dim3(64 * (unsigned)team), 0, stream, a);
		HIP_TRY(hipGetLastError());
	}
#if defined(POLAR_PROFILE) || defined(POLAR_SPAN)
	{
		static unsigned long long host[16 * 32 * 2];
		HIP_TRY(hipMemcpyAsync(host, prof_dev, sizeof(host), hipMemcpyDeviceToHost, stream));
		HIP_TRY(hipStreamSynchronize(stream));
		static const char *names[16] = {"left", "right", "comb", "rate0", "rate1", "rep", "spc", "r0right", "r0comb", "r1comb", "?", "chain1", "chain2", "chain3", "chain4", "subtree"};
		unsigned long long total = 0;
		for (int i = 0; i < 16 * 32; ++i)
			total += host[2 * i];
		{
			static unsigned long long span[2 * 4096];
			HIP_TRY(hipMemcpy(span, prof_dev + 16 * 32 * 2, sizeof(span), hipMemcpyDeviceToHost));
			// the counter is per XCD (different offsets): lifetimes, not absolute times; all wavefronts of a launch
			// that fits the device start within microseconds of one another
			std::vector<double> life;
			for (size_t i = 0; i < blocks && i < 4096; ++i)
				if (span[2 * i])
					life.push_back((double)(span[2 * i + 1] - span[2 * i]));
			{
				double by8[8] = {}, n8[8] = {};
				for (size_t i = 0; i < blocks && i < 4096; ++i)
					if (span[2 * i]) {
						by8[i % 8] += (double)(span[2 * i + 1] - span[2 * i]);
						n8[i % 8] += 1;
					}
				fprintf(stderr, "polar_hip profile: mean lifetime by workgroup index mod 8 (XCD under round-robin dispatch):");
				for (int k = 0; k < 8; ++k)
					fprintf(stderr, " %.0f", n8[k] ? by8[k] / n8[k] : 0.0);
				fprintf(stderr, "\npolar_hip profile: mean lifetime by workgroup index / 64 (arena address order):");
				for (size_t k = 0; k * 64 < blocks && k < 64; ++k) {
					double sum = 0, n = 0;
					for (size_t i = k * 64; i < (k + 1) * 64 && i < blocks; ++i)
						if (span[2 * i]) {
							sum += (double)(span[2 * i + 1] - span[2 * i]);
							n += 1;
						}
					fprintf(stderr, " %.0f", n ? sum / n / 1e6 : 0.0);
				}
				fprintf(stderr, "\n");
			}
			if (!life.empty()) {
				std::sort(life.begin(), life.end());
				double sum = 0;
				for (double e : life)
					sum += e;
				auto q = [&](double f) { return life[(size_t)(f * (life.size() - 1))]; };
				fprintf(stderr, "polar_hip profile: wavefront lifetimes (ticks of s_memtime), %zu wavefronts: min %.0f  5%% %.0f  25%% %.0f  median %.0f  75%% %.0f  95%% %.0f  max %.0f  mean %.0f (%.1f %% of max)\n",
					life.size(), life.front(), q(0.05), q(0.25), q(0.5), q(0.75), q(0.95), life.back(), sum / life.size(), 100.0 * sum / life.size() / life.back());
			}
		}
		fprintf(stderr, "polar_hip profile: wave-cycles per node kind and level (blocks=%zu, total %.3e cycles = %.1f Mcycles per wave)\n", blocks, (double)total, total / 1e6 / blocks);
		for (int k = 0; k < 16; ++k)
			for (int l = 0; l < 32; ++l)
				if (host[2 * (k * 32 + l) + 1])
					fprintf(stderr, "  %-8s L%-2d calls/wave %8.1f  cycles/call %9.0f  share %5.2f%%\n", names[k], l, (double)host[2 * (k * 32 + l) + 1] / blocks,
						(double)host[2 * (k * 32 + l)] / host[2 * (k * 32 + l) + 1], 100.0 * host[2 * (k * 32 + l)] / total);
	}
#endif
	if (timed) {
		HIP_TRY(hipEventRecord(h->ev1, stream));
		h->timed = true;
		if (tune_placement && direct && stream == h->stream && h->placement.program == program_id && (!h->placement.done || (h->placement.relay == polar_hip_handle::Placement::RELAY_TRIAL && segments)))
			h->placement.pending = true;
	}
	if (!direct && rows)
		HIP_TRY(hipMemcpy2DAsync(message, msg_pitch, msg, padded, frames, rows, hipMemcpyDeviceToDevice, stream));
	return POLAR_OK;
}

static int check_decode_args(polar_hip_handle *h, int program_id, const void *message, const void *codeword, size_t frames)
{
	if (!h)
		return fail(POLAR_E_ARG, "polar_hip_decode: null handle");
	if (program_id < 0 || (size_t)program_id >= h->programs.size())
		return fail(POLAR_E_ARG, "polar_hip_decode: unknown program id");
	if (frames && (!codeword || (!message && h->programs[program_id].host.rows)))
		return fail(POLAR_E_ARG, "polar_hip_decode: null buffer");
	return POLAR_OK;
}

// True when the runtime knows `p` as page-locked host memory (polar_hip_host_alloc, hipHostMalloc,
// hipHostRegister): only then do asynchronous copies run beside the host thread.
static bool host_pinned(const void *p)
{
	hipPointerAttribute_t attr;
	if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
		(void)hipGetLastError(); // pageable memory is "invalid value" to this query
		return false;
	}
	return attr.type == hipMemoryTypeHost;
}

// frames per chunk of the host-buffer pipeline: about 1 GiB of LLRs, never so few frames that the
// decode of a chunk falls into the small-batch regime, whole groups of every kernel shape
static size_t pipeline_chunk(size_t n)
{
	if (const char *e = getenv("POLAR_HIP_HOST_CHUNK")) { // frames; tests and experiments only
		size_t c = (size_t)atoll(e);
		return c < 512 ? 512 : c - c % 512;
	}
	size_t c = ((size_t)1 << 30) / n;
	if (c < 16384)
		c = 16384;
	return c - c % 512;
}
