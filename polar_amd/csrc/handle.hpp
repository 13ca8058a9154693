// handle.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): the handle (device buffers, resident programs, code cache), host-side construction entry points, create / destroy / configure, kernel-shape choice.
#pragma once

// Scratch of polar_hip_decode_static(): the wavefront arenas of callers that may not allocate.
// One copy per GPU, resident from code-object load until the process ends. Host code never uses
// the symbol's name as an address: g_static_scratch[device] holds what hipGetSymbolAddress
// returned with that device current (polar_hip_create).
enum : size_t { STATIC_SCRATCH_BYTES = (size_t)POLAR_STATIC_SCRATCH_BYTES };
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];
enum { MAX_DEVICES = 64, STATIC_Q = 8, STATIC_RL = 8, STATIC_TEAM = 4 }; // the kernel shape of the static entry point: 32 frames per wavefront, in teams of four when the batch is whole teams
static void *g_static_scratch[MAX_DEVICES];

// ---------------------------------------------------------------------------------------
// handle
// ---------------------------------------------------------------------------------------

struct DeviceBuffer {
	void *p = nullptr;
	size_t bytes = 0;
	// chunked form (reserve_chunked): one virtual range backed by separately allocated pieces of device
	// memory, mapped in a shuffled order (hipMemCreate / hipMemMap)
	std::vector<hipMemGenericAllocationHandle_t> pieces;
	size_t piece_bytes = 0;
	int reserve(size_t want)
	{
		if (want <= bytes)
			return POLAR_OK;
		release();
		HIP_TRY(hipMalloc(&p, want));
		bytes = want;
		return POLAR_OK;
	}
	// `want` bytes as pieces of `piece` bytes each, piece i of the range backed by allocation perm(i):
	// a fixed pseudo-random order when `shuffle`, else the order of allocation
	int reserve_chunked(size_t want, size_t piece, bool shuffle, int device, bool align = false)
	{
		if (want <= bytes)
			return POLAR_OK;
		release();
		hipMemAllocationProp prop = {};
		prop.type = hipMemAllocationTypePinned;
		prop.location.type = hipMemLocationTypeDevice;
		prop.location.id = device;
		size_t gran = 0;
		HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
		if (!gran)
			gran = (size_t)2 << 20;
		piece = (piece + gran - 1) / gran * gran;
		const size_t count = (want + piece - 1) / piece, total = count * piece;
		void *base = nullptr;
		// `align`: the range starts on a multiple of the piece size (a power of two then), so that a piece the driver
		// hands out aligned to its size can be mapped with one large translation
		HIP_TRY(hipMemAddressReserve(&base, total, align && !(piece & (piece - 1)) ? piece : 0, nullptr, 0));
		p = base;
		bytes = total;
		piece_bytes = piece;
		pieces.reserve(count);
		for (size_t i = 0; i < count; ++i) {
			hipMemGenericAllocationHandle_t hnd;
			hipError_t e = hipMemCreate(&hnd, piece, &prop, 0);
			if (e != hipSuccess) {
				release();
				HIP_TRY(e);
			}
			pieces.push_back(hnd);
		}
		std::vector<size_t> perm(count);
		for (size_t i = 0; i < count; ++i)
			perm[i] = i;
		if (shuffle) { // Fisher-Yates with a fixed LCG: the same order every time
			uint64_t x = 0x9E3779B97F4A7C15ull;
			for (size_t i = count - 1; i > 0; --i) {
				x = x * 6364136223846793005ull + 1442695040888963407ull;
				std::swap(perm[i], perm[(size_t)((x >> 33) % (i + 1))]);
			}
		}
		mapped = 0;
		for (size_t i = 0; i < count; ++i) {
			hipError_t e = hipMemMap((char *)base + i * piece, piece, 0, pieces[perm[i]], 0);
			if (e != hipSuccess) {
				release();
				HIP_TRY(e);
			}
			mapped = i + 1;
		}
		hipMemAccessDesc acc = {};
		acc.location = prop.location;
		acc.flags = hipMemAccessFlagsProtReadWrite;
		hipError_t e = hipMemSetAccess(base, total, &acc, 1);
		if (e != hipSuccess) {
			release();
			HIP_TRY(e);
		}
		return POLAR_OK;
	}
	// The wavefront arenas. From 2 GiB on: pieces of 1 GiB, each allocated on its own and mapped at a multiple
	// of 1 GiB of one virtual range. A plain hipMalloc of the same size lands, about every other time, on
	// memory that costs the decode kernel 5 % (231 instead of 219 ms for 2^20 frames of N = 2^16, the same
	// virtual address both times); with pieces that the driver hands out aligned to their size and that are
	// mapped equally aligned, four allocations in four gave 218-222 ms (profiles/r3_placement_pieces.txt;
	// pieces of 2-8 MiB in a shuffled order: always the slow 232 ms, which points at the size of the address
	// translations). Falls back to hipMalloc where the virtual-memory calls are refused.
	// POLAR_HIP_ARENA_PIECES=0 forces the plain allocation.
	int reserve_arenas(size_t want, int device)
	{
		if (want <= bytes)
			return POLAR_OK;
		static const bool plain = getenv("POLAR_HIP_ARENA_PIECES") && atoi(getenv("POLAR_HIP_ARENA_PIECES")) == 0;
		if (!plain && want >= ((size_t)2 << 30)) {
			if (reserve_chunked(want, (size_t)1 << 30, false, device, true) == POLAR_OK)
				return POLAR_OK;
			(void)hipGetLastError();
			release();
		}
		return reserve(want);
	}
	size_t mapped = 0;
	void release()
	{
		if (piece_bytes) {
			(void)hipDeviceSynchronize(); // hipFree waits for the device by itself, unmapping does not
			for (size_t i = 0; i < mapped; ++i)
				(void)hipMemUnmap((char *)p + i * piece_bytes, piece_bytes);
			for (auto hnd : pieces)
				(void)hipMemRelease(hnd);
			if (p)
				(void)hipMemAddressFree(p, bytes);
			pieces.clear();
			piece_bytes = 0;
			mapped = 0;
		} else if (p) {
			(void)hipFree(p);
		}
		p = nullptr;
		bytes = 0;
	}
};

struct LoadedProgram {
	HostProgram host;                // unfused translation: level, rows, validation
	std::vector<uint8_t> bytes;      // the reference byte-code as loaded
	std::map<int, DevOp *> dev;      // device instruction stream per frame-quads-per-wave (fusion depends on it)
	uint8_t *dev_bytes = nullptr;    // the byte-code itself on the device (float decoder)
};

struct polar_hip_handle {
	int device = 0;
	int cus = 0;
	hipStream_t own_stream = nullptr; // created with the handle
	hipStream_t stream = nullptr;     // where the synchronous entry points enqueue: own_stream or the caller's (polar_hip_set_stream)
	hipEvent_t ev0 = nullptr, ev1 = nullptr;
	// relay launches (decode_launch.hpp): the second stream and the events that fork it from and join it to the caller's
	hipStream_t relay = nullptr;
	hipEvent_t relay_fork = nullptr, relay_join = nullptr;
	hipStream_t relay_partner = nullptr; // the stream `relay` was last checked against (relay_ready): kernels on the two run side by side ...
	bool relay_partner_known = false, relay_usable = false; // ... or, if no second stream was found that does, relay launches are off for it
	// host-buffer pipeline (polar_hip_decode with pinned POLAR_HOST buffers): copy-in and copy-out
	// streams beside `stream`, and per staging slot the events "copied in", "decoded", "copied out"
	hipStream_t pipe_in = nullptr, pipe_out = nullptr;
	hipEvent_t pipe_ev[2][3] = {};
	void *bounce_in = nullptr, *bounce_out = nullptr; // page-locked staging for pageable caller buffers, two slots each
	size_t bounce_in_bytes = 0, bounce_out_bytes = 0;
	std::map<const void *, int> occupancy; // resident workgroups per CU, per kernel variant
	bool timed = false;
	std::vector<LoadedProgram> programs;
	struct CachedCode {
		std::vector<uint8_t> frozen, program;
		int program_id, ambiguous;
	};
	struct CodeKey {
		int level, K;
		long double p;
		bool operator<(const CodeKey &o) const { return level != o.level ? level < o.level : K != o.K ? K < o.K : p < o.p; }
	};
	std::map<CodeKey, CachedCode> codes; // polar_hip_code: (level, K, p) -> frozen set, program, resident id
	DeviceBuffer scratch, scratch_spare, hot, stage_in, stage_out, enc_tmp, tables, counters, pad_in, pad_out;
	DeviceBuffer sim_cw, sim_llr, sim_msg, sim_dec; // polar_hip_simulate's chunk
	DeviceBuffer f32_a, f32_b;                      // float path: LLRs / decisions of a chunk, symbol conversions
	std::vector<uint8_t> frozen_cache; // frozen set the device tables were built from
	int cfg_frames_per_wave = 0, cfg_lds_level = 0, cfg_team = 1, cfg_max_waves = 0;
	// Placement of the scratch arenas (decode_launch.hpp, placement_step): the workload being watched,
	// the kernel times of the placements tried so far, and whether the last launch is still to be read
	struct Placement {
		int program = -1, tries = 0;
		size_t frames = 0, bytes = 0;
		const void *fn = nullptr;
		float best_ms = 0;
		bool pending = false, done = false;
		// relay launches for this workload (decode_launch.hpp): decided by the clock once the placement is chosen:
		// one call as relay launches against the best single launch
		enum { RELAY_UNTESTED, RELAY_TRIAL, RELAY_ON, RELAY_OFF };
		int relay = RELAY_UNTESTED;
	} placement;
	DeviceBuffer set_table; // polar_hip_decode_set: the argument table of the common launch
	int last_shape[3] = {0, 0, 0}; // frames per wavefront, LDS level, team of the last decode launch
	int last_launches = 1;         // kernel launches it went out as (relay launches, decode_launch.hpp)
};

static int use_device(polar_hip_handle *h)
{
	HIP_TRY(hipSetDevice(h->device));
	return POLAR_OK;
}

// Device instruction stream of a loaded program for one wavefront shape (fusion depends on the
// frame quads per wavefront and the variant's chain depth): fused and uploaded on first use.
static int program_shape(const KernelVariant *v) { return v->q * 8 + v->chain_depth; }
static int upload_shape(LoadedProgram &lp, const KernelVariant *v)
{
	const int shape = program_shape(v);
	if (lp.dev.count(shape))
		return POLAR_OK;
	HostProgram hp;
	if (translate_program(lp.bytes.data(), (int)lp.bytes.size(), hp, v->q, v->chain_depth))
		return fail(POLAR_E_PROGRAM, "program translation failed");
	DevOp *d = nullptr;
	size_t bytes = hp.ops.size() * sizeof(DevOp);
	HIP_TRY(hipMalloc((void **)&d, bytes));
	hipError_t e = hipMemcpy(d, hp.ops.data(), bytes, hipMemcpyHostToDevice);
	if (e != hipSuccess) {
		(void)hipFree(d);
		HIP_TRY(e);
	}
	lp.dev[shape] = d;
	return POLAR_OK;
}

extern "C" const char *polar_hip_last_error(void)
{
	return g_error.c_str();
}

// ---- host-side construction ----

extern "C" int polar_code_const0_ex(uint8_t *frozen, int level, int K, long double erasure_probability, int *ambiguous)
{
	if (!frozen || level < 0 || level > MAX_LEVEL || K < 0 || K > (1 << level))
		return fail(POLAR_E_ARG, "polar_code_const0: bad argument");
	int amb = code_const0(frozen, level, K, erasure_probability);
	if (ambiguous)
		*ambiguous = amb;
	return POLAR_OK;
}

extern "C" int polar_code_const0(uint8_t *frozen, int level, int K, long double erasure_probability)
{
	return polar_code_const0_ex(frozen, level, K, erasure_probability, nullptr);
}

extern "C" int polar_code_const0_batch(uint8_t *frozen, int *ambiguous, int level, int K, const long double *erasure_probability, int count)
{
	if (!frozen || !erasure_probability || count < 0 || level < 0 || level > MAX_LEVEL || K < 0 || K > (1 << level))
		return fail(POLAR_E_ARG, "polar_code_const0_batch: bad argument");
	const size_t n = (size_t)1 << level;
	unsigned hw = std::thread::hardware_concurrency();
	int workers = (int)(hw ? (hw > 16 ? 16 : hw) : 4);
	if (workers > count)
		workers = count;
	auto work = [&](int w) {
		for (int i = w; i < count; i += workers) {
			int amb = code_const0(frozen + (size_t)i * n, level, K, erasure_probability[i]);
			if (ambiguous)
				ambiguous[i] = amb;
		}
	};
	std::vector<std::thread> pool;
	for (int w = 1; w < workers; ++w)
		pool.emplace_back(work, w);
	if (workers > 0)
		work(0);
	for (auto &t : pool)
		t.join();
	return POLAR_OK;
}

extern "C" int polar_freezer(uint8_t *frozen, int level, long double erasure_probability, long double freezing_threshold)
{
	if (!frozen || level < 0 || level > MAX_LEVEL)
		return fail(POLAR_E_ARG, "polar_freezer: bad argument");
	return freezer(frozen, level, erasure_probability, freezing_threshold);
}

extern "C" int polar_compile(uint8_t *program, const uint8_t *frozen, int level)
{
	if (!program || !frozen || level < 1 || level > MAX_LEVEL)
		return fail(POLAR_E_ARG, "polar_compile: bad argument");
	ProgramBuilder b;
	int n = b.run(program, frozen, level);
	if (n < 0)
		return fail(POLAR_E_PROGRAM, "polar_compile: an information bit directly left of a frozen bit cannot be compiled (reference asserts level > 0)");
	return n;
}

extern "C" int polar_program_rows(const uint8_t *program, int length)
{
	if (!program)
		return fail(POLAR_E_ARG, "polar_program_rows: null program");
	HostProgram hp;
	int err = translate_program(program, length, hp);
	if (err)
		return fail(POLAR_E_PROGRAM, "malformed program (rule " + std::to_string(err) + ")");
	return (int)hp.rows;
}

// ---- device objects ----

extern "C" int polar_hip_create(polar_hip_handle **out, int device)
{
	if (!out)
		return fail(POLAR_E_ARG, "polar_hip_create: null out pointer");
	*out = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
		return fail(POLAR_E_NODEVICE, "polar_hip_create: no HIP device");
	if (device < 0 || device >= count)
		return fail(POLAR_E_ARG, "polar_hip_create: device index out of range");
	hipDeviceProp_t prop;
	HIP_TRY(hipGetDeviceProperties(&prop, device));
	if (!strstr(prop.gcnArchName, "gfx950"))
		return fail(POLAR_E_NODEVICE, std::string("polar_hip_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
	polar_hip_handle *h = new polar_hip_handle;
	h->device = device;
	h->cus = prop.multiProcessorCount;
	auto init = [&]() -> int {
		HIP_TRY(hipSetDevice(device));
		HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
		h->stream = h->own_stream;
		HIP_TRY(hipEventCreate(&h->ev0));
		HIP_TRY(hipEventCreate(&h->ev1));
		if (device < MAX_DEVICES && !g_static_scratch[device]) { // once per device, with that device current
			void *ptr = nullptr;
			HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
			g_static_scratch[device] = ptr;
		}
		for (int team : {1, (int)STATIC_TEAM}) { // the static entry point must not meet a first-use query: resident workgroups of its kernel shapes, now
			const KernelVariant *sv = find_variant(STATIC_Q, STATIC_RL, team);
			if (!sv)
				return fail(POLAR_E_HIP, "polar_hip_create: the kernel shape of polar_hip_decode_static is not built into this library");
			int per_cu = 0;
			HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, sv->fn, 64 * team, 0));
			h->occupancy[(const void *)sv->fn] = per_cu;
		}
		return POLAR_OK;
	};
	int rc = init();
	if (rc) { // nothing of a half-made handle is left behind (the error text is kept)
		std::string keep = g_error;
		polar_hip_destroy(h);
		g_error = keep;
		return rc;
	}
	*out = h;
	return POLAR_OK;
}

extern "C" void polar_hip_destroy(polar_hip_handle *h)
{
	if (!h)
		return;
	(void)hipSetDevice(h->device);
	(void)hipStreamSynchronize(h->stream);
	if (h->stream != h->own_stream)
		(void)hipStreamSynchronize(h->own_stream);
	h->set_table.release();
	for (auto &p : h->programs) {
		for (auto &d : p.dev)
			(void)hipFree(d.second);
		if (p.dev_bytes)
			(void)hipFree(p.dev_bytes);
	}
	h->scratch.release();
	h->stage_in.release();
	h->stage_out.release();
	h->enc_tmp.release();
	h->tables.release();
	h->counters.release();
	h->pad_in.release();
	h->hot.release();
	h->scratch_spare.release();
	h->pad_out.release();
	h->f32_a.release();
	h->f32_b.release();
	h->sim_cw.release();
	h->sim_llr.release();
	h->sim_msg.release();
	h->sim_dec.release();
	(void)hipEventDestroy(h->ev0);
	(void)hipEventDestroy(h->ev1);
	if (h->relay)
		(void)hipStreamDestroy(h->relay);
	if (h->relay_fork)
		(void)hipEventDestroy(h->relay_fork);
	if (h->relay_join)
		(void)hipEventDestroy(h->relay_join);
	if (h->bounce_in)
		(void)hipHostFree(h->bounce_in);
	if (h->bounce_out)
		(void)hipHostFree(h->bounce_out);
	if (h->pipe_in) {
		(void)hipStreamDestroy(h->pipe_in);
		(void)hipStreamDestroy(h->pipe_out);
		for (auto &slot : h->pipe_ev)
			for (hipEvent_t e : slot)
				(void)hipEventDestroy(e);
	}
	(void)hipStreamDestroy(h->own_stream);
	delete h;
}

extern "C" int polar_hip_configure_team(polar_hip_handle *h, int frames_per_wave, int lds_level, int team, int max_waves)
{
	if (!h)
		return fail(POLAR_E_ARG, "polar_hip_configure: null handle");
	if (frames_per_wave || lds_level) {
		if (frames_per_wave % 4 || !find_variant(frames_per_wave / 4, lds_level, team))
			return fail(POLAR_E_ARG, "polar_hip_configure: no kernel built for this (frames_per_wave, lds_level, team)");
	}
	if (max_waves < 0)
		return fail(POLAR_E_ARG, "polar_hip_configure: negative max_waves");
	h->cfg_frames_per_wave = frames_per_wave;
	h->cfg_lds_level = lds_level;
	h->cfg_team = team;
	h->cfg_max_waves = max_waves;
	return POLAR_OK;
}

extern "C" int polar_hip_configure(polar_hip_handle *h, int frames_per_wave, int lds_level, int max_waves)
{
	return polar_hip_configure_team(h, frames_per_wave, lds_level, 1, max_waves);
}

extern "C" int polar_hip_set_stream(polar_hip_handle *h, void *stream, int use_callers)
{
	if (!h)
		return fail(POLAR_E_ARG, "polar_hip_set_stream: null handle");
	int rc = use_device(h);
	if (rc)
		return rc;
	hipStream_t next = use_callers ? (hipStream_t)stream : h->own_stream;
	if (next != h->stream) // work already enqueued through the handle shares its scratch: let it drain
		HIP_TRY(hipStreamSynchronize(h->stream));
	h->stream = next;
	return POLAR_OK;
}

extern "C" int polar_hip_load_program(polar_hip_handle *h, const uint8_t *program, int length)
{
	if (!h || !program)
		return fail(POLAR_E_ARG, "polar_hip_load_program: null argument");
	for (size_t i = 0; i < h->programs.size(); ++i) // the same byte-code again: the resident copy
		if ((int)h->programs[i].bytes.size() == length && !memcmp(h->programs[i].bytes.data(), program, (size_t)length))
			return (int)i;
	LoadedProgram lp;
	int err = translate_program(program, length, lp.host);
	if (err)
		return fail(POLAR_E_PROGRAM, "polar_hip_load_program: malformed program (rule " + std::to_string(err) + ")");
	lp.bytes.assign(program, program + length);
	// the shape polar_hip_decode_static() runs is made resident now: that entry point may not allocate
	int rc = use_device(h);
	if (rc || (rc = upload_shape(lp, find_variant(STATIC_Q, STATIC_RL))))
		return rc;
	h->programs.push_back(lp);
	return (int)h->programs.size() - 1;
}

// Bytes of scratch one wavefront needs: LLR levels above the LDS levels plus two partial-sum planes.
static size_t scratch_bytes_per_wave(int q, int top)
{
	size_t soft = (size_t)q << top;
	return (soft + 2 * (soft / 8)) * sizeof(u32);
}

// Default shape: the 20 KiB-of-LDS family (8 wavefronts per CU), widest first. 128 frames per
// wavefront measured best at N = 2^16; narrower groups when the batch is too small
// to give every resident wavefront a group, or when the scratch arenas would not fit.
// Scratch budget: what the handle holds already plus 5/8 of the free device memory (so that
// N = 2^20 with 2^16 frames gets its 80 GiB of (32, 8) arenas beside 128 GiB of caller data). The query
// is slow (milliseconds), so it is made only when a request exceeds what is held.
struct ScratchBudget {
	const polar_hip_handle *h;
	size_t cap = 0;
	bool queried = false;
	bool fits(size_t bytes)
	{
		if (bytes <= h->scratch.bytes)
			return true;
		return bytes <= limit();
	}
	size_t limit()
	{
		if (!queried) {
			size_t free_b = 0, total_b = 0;
			if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
				free_b = 0;
			cap = h->scratch.bytes + free_b / 8 * 5;
			queried = true;
		}
		return cap;
	}
};

extern "C" int polar_hip_load_program(polar_hip_handle *h, const uint8_t *program, int length);

extern "C" int polar_hip_code(polar_hip_handle *h, int level, int K, long double erasure_probability, const uint8_t **frozen, const uint8_t **program, int *program_length, int *ambiguous)
{
	if (!h || level < 1 || level > MAX_LEVEL || K < 0 || K > (1 << level) || !(erasure_probability >= 0) || !(erasure_probability <= 1))
		return fail(POLAR_E_ARG, "polar_hip_code: bad argument");
	polar_hip_handle::CodeKey key = {level, K, erasure_probability};
	auto it = h->codes.find(key);
	if (it == h->codes.end()) {
		polar_hip_handle::CachedCode c;
		const size_t n = (size_t)1 << level;
		c.frozen.resize(n);
		c.ambiguous = code_const0(c.frozen.data(), level, K, erasure_probability);
		c.program.resize(2 * n + 1);
		ProgramBuilder b;
		int len = b.run(c.program.data(), c.frozen.data(), level);
		if (len < 0)
			return fail(POLAR_E_PROGRAM, "polar_hip_code: the frozen set of this (level, K, p) cannot be compiled (reference asserts level > 0)");
		c.program.resize((size_t)len);
		c.program_id = polar_hip_load_program(h, c.program.data(), len);
		if (c.program_id < 0)
			return c.program_id;
		it = h->codes.emplace(key, std::move(c)).first;
	}
	if (frozen)
		*frozen = it->second.frozen.data();
	if (program)
		*program = it->second.program.data();
	if (program_length)
		*program_length = (int)it->second.program.size();
	if (ambiguous)
		*ambiguous = it->second.ambiguous;
	return it->second.program_id;
}

// Highest scratch level that keeps the default cache policy (sc_core.hpp, GVec::hot): the levels just
// above the LDS levels, as many as the Infinity Cache holds for all resident wavefronts together
// (wavefronts * frames per wavefront * 2^level bytes per level).
// Their vectors are read back and rewritten in place before the streamed (non-temporal) levels above
// can displace them, so they never travel to HBM (tools/ubench/hbm_ceiling.hip).
static int hot_top_for(const KernelVariant *v, int top, size_t waves)
{
	if (v->team > 1 && !POLAR_TEAM_HOT) // the team kernels are built without hot levels (sc_core.hpp, Group::HOT)
		return v->rl;
	static const char *e = getenv("POLAR_HIP_HOT_LEVELS"); // experiments: number of hot levels
	if (e)
		return v->rl + atoi(e);
	static const char *b = getenv("POLAR_HIP_HOT_BYTES"); // experiments: cache budget in MiB
	const size_t budget = (b ? (size_t)atoi(b) : (size_t)480) << 20;
	size_t used = 0;
	int lvl = v->rl;
	while (lvl + 1 < top) {
		size_t bytes = waves * 4 * (size_t)v->q << (lvl + 1);
		if (used + bytes > budget)
			break;
		used += bytes;
		++lvl;
	}
	return lvl;
}

static const KernelVariant *pick_variant(const polar_hip_handle *h, int top, size_t frames, ScratchBudget &budget)
{
	if (h->cfg_frames_per_wave)
		return find_variant(h->cfg_frames_per_wave / 4, h->cfg_lds_level, h->cfg_team);
	// A wavefront walks the program in about the same time whatever its width, a little longer the busier the
	// device is. Per shape: its walk time on a lightly and on a fully loaded device, relative, measured at
	// N = 2^16 (profiles/r3_pick_sweep.txt; the ratios hold for the other code sizes). A batch of `units` groups
	// (teams) on `cap` resident ones takes ceil(units / cap) sweeps of one walk each at load units / (sweeps * cap).
	// The shape with the shortest time by this model wins; a narrower one must win by 3 % (steady multi-sweep
	// batches favour the wide shape's hot levels a little more than the one-sweep figures say).
	// Groups of 32 and 64 frames go in teams of four (whole 128-byte input lines).
	// A handle limited to a share of the device (polar_hip_configure: max_waves; several handles decoding
	// side by side on their own streams) picks its shape for that share.
	size_t resident = (size_t)8 * h->cus;
	if (h->cfg_max_waves && (size_t)h->cfg_max_waves < resident)
		resident = (size_t)h->cfg_max_waves;
	// walk = light + (full - light) * load^bend: the wide shapes slow down in proportion to the load (bandwidth),
	// the narrow ones stay near their latency floor until the device is nearly full
	static const struct { int q, rl, team; double light, full; int bend; } family[] = {
		{32, 6, 1, 27.0, 57.0, 1}, {16, 7, 4, 17.3, 29.1, 1}, {8, 8, 4, 10.8, 16.9, 2}, {4, 9, 1, 7.3, 14.8, 2}, {2, 10, 1, 7.0, 10.8, 2}, {1, 11, 1, 7.0, 9.4, 2}};
	const KernelVariant *best = nullptr;
	double best_t = 0;
	for (const auto &f : family) {
		const KernelVariant *v = find_variant(f.q, f.rl, f.team);
		if (!v)
			continue;
		if (f.q > 1 && top > f.rl && !budget.fits(resident * scratch_bytes_per_wave(f.q, top)))
			continue; // the arenas of a full device would not fit: a narrower shape (fewer resident wavefronts as the last resort)
		const size_t unit = 4 * (size_t)f.q * f.team, units = (frames + unit - 1) / unit;
		const size_t cap = resident / f.team ? resident / f.team : 1;
		const size_t sweeps = (units + cap - 1) / cap;
		const double load = (double)units / (double)(sweeps * cap);
		const double t = (double)sweeps * (f.light + (f.full - f.light) * (f.bend == 2 ? load * load : load));
		if (!best || t < best_t * 0.97) {
			best = v;
			best_t = t;
		}
	}
	return best;
}
