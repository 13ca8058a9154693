// abi_shard.hpp -- part of polar_hip.hip (one translation unit; included there, in this order): counter all-reduce for callers that shard over GPUs (RCCL).
#pragma once

// ---------------------------------------------------------------------------------------
// Counter exchange for callers that shard the frames over GPUs themselves (SURVEY 8 e): the one
// collective of the path, an all-reduce (sum) of the int64 counters of testbench.cc:119-125, 185-192.
// RCCL is not linked: the library that made the caller's communicator is already in the process (or
// librccl.so is loaded here), and its ncclAllReduce is looked up once.
// ---------------------------------------------------------------------------------------
#include <dlfcn.h>

typedef int (*nccl_allreduce_t)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*nccl_errstr_t)(int);

extern "C" int polar_hip_allreduce_counters(polar_hip_handle *h, int64_t *counters, size_t count, void *nccl_comm)
{
	if (!h || !counters || !count || !nccl_comm)
		return fail(POLAR_E_ARG, "polar_hip_allreduce_counters: bad argument");
	if (polar_hip_where(counters) != POLAR_DEVICE)
		return fail(POLAR_E_ARG, "polar_hip_allreduce_counters: counters must be device memory (polar_hip_simulate / polar_hip_count with POLAR_DEVICE)");
	static nccl_allreduce_t allreduce = nullptr;
	static nccl_errstr_t errstr = nullptr;
	if (!allreduce) {
		void *lib = nullptr;
		if (dlsym(RTLD_DEFAULT, "ncclAllReduce")) // the caller linked RCCL: its symbols are global
			lib = RTLD_DEFAULT;
		const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
		for (int k = 0; !lib && k < 3; ++k)
			lib = dlopen(names[k], RTLD_NOW | RTLD_NOLOAD); // loaded by the caller without global symbols (dlopen, Python)
		for (int k = 0; !lib && k < 3; ++k)
			lib = dlopen(names[k], RTLD_NOW);
		if (!lib)
			return fail(POLAR_E_ARG, "polar_hip_allreduce_counters: RCCL (librccl.so) is not in this process and cannot be loaded");
		allreduce = (nccl_allreduce_t)dlsym(lib, "ncclAllReduce");
		errstr = (nccl_errstr_t)dlsym(lib, "ncclGetErrorString");
		if (!allreduce)
			return fail(POLAR_E_ARG, "polar_hip_allreduce_counters: ncclAllReduce not found in RCCL");
	}
	int rc = use_device(h);
	if (rc)
		return rc;
	const int nccl_int64 = 4, nccl_sum = 0; // ncclInt64, ncclSum (rccl.h)
	int st = allreduce(counters, counters, count, nccl_int64, nccl_sum, nccl_comm, h->stream);
	if (st) {
		static thread_local char msg[160];
		snprintf(msg, sizeof msg, "polar_hip_allreduce_counters: ncclAllReduce failed (%d): %s", st, errstr ? errstr(st) : "?");
		return fail(POLAR_E_HIP, msg);
	}
	HIP_TRY(hipStreamSynchronize(h->stream));
	return POLAR_OK;
}
