// host_pipeline.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): host-buffer paths of polar_hip_decode: page-locked allocation, chunked copy / decode / copy pipelines.
#pragma once

extern "C" int polar_hip_where(const void *ptr)
{
	hipPointerAttribute_t attr;
	if (!ptr || hipPointerGetAttributes(&attr, ptr) != hipSuccess) {
		(void)hipGetLastError(); // pageable host memory is "invalid value" to this query
		return POLAR_HOST;
	}
	return attr.type == hipMemoryTypeDevice ? POLAR_DEVICE : POLAR_HOST;
}

extern "C" void *polar_hip_host_alloc(size_t bytes)
{
	void *p = nullptr;
	if (!bytes || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
		(void)hipGetLastError();
		fail(POLAR_E_NOMEM, "polar_hip_host_alloc: no page-locked memory of that size");
		return nullptr;
	}
	return p;
}

extern "C" void polar_hip_host_free(void *ptr)
{
	if (ptr)
		(void)hipHostFree(ptr);
}

enum { COPIED_IN = 0, DECODED = 1, COPIED_OUT = 2 }; // pipe_ev[slot][...]

static int pipeline_setup(polar_hip_handle *h, size_t n, size_t rows, size_t chunk)
{
	if (!h->pipe_in) {
		HIP_TRY(hipStreamCreateWithFlags(&h->pipe_in, hipStreamNonBlocking));
		HIP_TRY(hipStreamCreateWithFlags(&h->pipe_out, hipStreamNonBlocking));
		for (auto &slot : h->pipe_ev)
			for (hipEvent_t &e : slot)
				HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	}
	int rc;
	if ((rc = h->stage_in.reserve(2 * n * chunk)) || (rc = h->stage_out.reserve(2 * (rows ? rows : 1) * chunk)))
		return rc;
	return POLAR_OK;
}

// rows [0, count) split over a few host threads
template <class F>
static void parallel_rows(size_t count, F fn)
{
	static const unsigned want = [] {
		const char *e = getenv("POLAR_HIP_HOST_THREADS");
		unsigned t = e ? (unsigned)atoi(e) : std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
		return t < 1 ? 1u : t;
	}();
	unsigned threads = (unsigned)std::min<size_t>(want, count ? count : 1);
	std::vector<std::thread> pool;
	for (unsigned t = 1; t < threads; ++t)
		pool.emplace_back([=] { fn(count * t / threads, count * (t + 1) / threads); });
	fn(0, count / threads);
	for (auto &th : pool)
		th.join();
}

// Pageable host buffers, several chunks' worth of frames: the column block of a chunk is gathered
// by host threads into a page-locked slot, copied in, decoded and copied out asynchronously, and
// scattered back into the caller's rows while the next chunk is on its way.
static int decode_host_staged_body(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, size_t chunk)
{
	const HostProgram &hp = h->programs[program_id].host;
	const size_t n = (size_t)1 << hp.top, rows = hp.rows;
	int rc;
	if ((rc = pipeline_setup(h, n, rows, chunk)))
		return rc;
	if (h->bounce_in_bytes < 2 * n * chunk) {
		if (h->bounce_in)
			(void)hipHostFree(h->bounce_in);
		h->bounce_in = nullptr;
		h->bounce_in_bytes = 0;
		HIP_TRY(hipHostMalloc(&h->bounce_in, 2 * n * chunk, hipHostMallocDefault));
		h->bounce_in_bytes = 2 * n * chunk;
	}
	if (rows && h->bounce_out_bytes < 2 * rows * chunk) {
		if (h->bounce_out)
			(void)hipHostFree(h->bounce_out);
		h->bounce_out = nullptr;
		h->bounce_out_bytes = 0;
		HIP_TRY(hipHostMalloc(&h->bounce_out, 2 * rows * chunk, hipHostMallocDefault));
		h->bounce_out_bytes = 2 * rows * chunk;
	}
	const size_t chunks = (frames + chunk - 1) / chunk;
	for (size_t i = 0; i <= chunks; ++i) {
		const int b = (int)(i & 1);
		if (i < chunks) {
			const size_t f0 = i * chunk, c = frames - f0 < chunk ? frames - f0 : chunk;
			int8_t *hin = (int8_t *)h->bounce_in + b * n * chunk, *hout = (int8_t *)h->bounce_out + b * rows * chunk;
			int8_t *din = (int8_t *)h->stage_in.p + b * n * chunk, *dout = (int8_t *)h->stage_out.p + b * rows * chunk;
			if (i >= 2) // the slot's previous chunk has left the page-locked buffer
				HIP_TRY(hipEventSynchronize(h->pipe_ev[b][COPIED_IN]));
			parallel_rows(n, [=](size_t r0, size_t r1) {
				for (size_t r = r0; r < r1; ++r)
					memcpy(hin + r * c, codeword + r * frames + f0, c);
			});
			if (i >= 2)
				HIP_TRY(hipStreamWaitEvent(h->pipe_in, h->pipe_ev[b][DECODED], 0));
			HIP_TRY(hipMemcpyAsync(din, hin, n * c, hipMemcpyHostToDevice, h->pipe_in));
			HIP_TRY(hipEventRecord(h->pipe_ev[b][COPIED_IN], h->pipe_in));
			HIP_TRY(hipStreamWaitEvent(h->stream, h->pipe_ev[b][COPIED_IN], 0));
			if (i >= 2 && rows)
				HIP_TRY(hipStreamWaitEvent(h->stream, h->pipe_ev[b][COPIED_OUT], 0));
			if ((rc = launch_decode(h, program_id, dout, din, c, h->stream, true)))
				return rc;
			HIP_TRY(hipEventRecord(h->pipe_ev[b][DECODED], h->stream));
			if (rows) { // the slot's previous messages were scattered in the iteration before this one
				HIP_TRY(hipStreamWaitEvent(h->pipe_out, h->pipe_ev[b][DECODED], 0));
				HIP_TRY(hipMemcpyAsync(hout, dout, rows * c, hipMemcpyDeviceToHost, h->pipe_out));
				HIP_TRY(hipEventRecord(h->pipe_ev[b][COPIED_OUT], h->pipe_out));
			}
		}
		if (i >= 1 && rows) {
			const int pb = (int)((i - 1) & 1);
			const size_t f0 = (i - 1) * chunk, c = frames - f0 < chunk ? frames - f0 : chunk;
			const int8_t *hout = (const int8_t *)h->bounce_out + pb * rows * chunk;
			HIP_TRY(hipEventSynchronize(h->pipe_ev[pb][COPIED_OUT]));
			parallel_rows(rows, [=](size_t r0, size_t r1) {
				for (size_t r = r0; r < r1; ++r)
					memcpy(message + r * frames + f0, hout + r * c, c);
			});
		}
	}
	HIP_TRY(hipStreamSynchronize(h->pipe_in));
	HIP_TRY(hipStreamSynchronize(h->stream));
	HIP_TRY(hipStreamSynchronize(h->pipe_out));
	return POLAR_OK;
}

// Page-locked host buffers, several chunks' worth of frames: chunk i+1 is copied in and chunk i-1
// copied out while chunk i decodes (three streams, two staging slots, column blocks of the
// [rows][frames] matrices as 2D copies).
static int decode_host_pipelined_body(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, size_t chunk)
{
	const HostProgram &hp = h->programs[program_id].host;
	const size_t n = (size_t)1 << hp.top, rows = hp.rows;
	int rc;
	if ((rc = pipeline_setup(h, n, rows, chunk)))
		return rc;
	size_t i = 0;
	for (size_t f0 = 0; f0 < frames; f0 += chunk, ++i) {
		const size_t c = frames - f0 < chunk ? frames - f0 : chunk;
		const int b = (int)(i & 1);
		int8_t *din = (int8_t *)h->stage_in.p + b * n * chunk, *dout = (int8_t *)h->stage_out.p + b * rows * chunk;
		if (i >= 2) // the slot's previous chunk has been decoded
			HIP_TRY(hipStreamWaitEvent(h->pipe_in, h->pipe_ev[b][DECODED], 0));
		HIP_TRY(hipMemcpy2DAsync(din, c, codeword + f0, frames, c, n, hipMemcpyHostToDevice, h->pipe_in));
		HIP_TRY(hipEventRecord(h->pipe_ev[b][COPIED_IN], h->pipe_in));
		HIP_TRY(hipStreamWaitEvent(h->stream, h->pipe_ev[b][COPIED_IN], 0));
		if (i >= 2 && rows) // ... and its messages have left the slot
			HIP_TRY(hipStreamWaitEvent(h->stream, h->pipe_ev[b][COPIED_OUT], 0));
		if ((rc = launch_decode(h, program_id, dout, din, c, h->stream, true))) // timed: polar_hip_last_decode_ms reports the last chunk
			return rc;
		HIP_TRY(hipEventRecord(h->pipe_ev[b][DECODED], h->stream));
		if (rows) {
			HIP_TRY(hipStreamWaitEvent(h->pipe_out, h->pipe_ev[b][DECODED], 0));
			HIP_TRY(hipMemcpy2DAsync(message + f0, frames, dout, c, c, rows, hipMemcpyDeviceToHost, h->pipe_out));
			HIP_TRY(hipEventRecord(h->pipe_ev[b][COPIED_OUT], h->pipe_out));
		}
	}
	HIP_TRY(hipStreamSynchronize(h->pipe_in));
	HIP_TRY(hipStreamSynchronize(h->stream));
	HIP_TRY(hipStreamSynchronize(h->pipe_out));
	return POLAR_OK;
}

// a failure in the middle of a pipeline leaves copies in flight that still name the caller's
// buffers and the page-locked slots: let them finish before the error goes back
static int drained(polar_hip_handle *h, int rc)
{
	if (rc && h->pipe_in) {
		const std::string why = g_error;
		(void)hipStreamSynchronize(h->pipe_in);
		(void)hipStreamSynchronize(h->stream);
		(void)hipStreamSynchronize(h->pipe_out);
		(void)hipGetLastError();
		g_error = why;
	}
	return rc;
}
static int decode_host_staged(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, size_t chunk)
{
	return drained(h, decode_host_staged_body(h, program_id, message, codeword, frames, chunk));
}
static int decode_host_pipelined(polar_hip_handle *h, int program_id, int8_t *message, const int8_t *codeword, size_t frames, size_t chunk)
{
	return drained(h, decode_host_pipelined_body(h, program_id, message, codeword, frames, chunk));
}
