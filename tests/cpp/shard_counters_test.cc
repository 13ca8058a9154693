// tests/cpp/shard_counters_test.cc -- a C++ caller that shards the bench loop over GPUs the way
// INTEGRATION.md shows: one process per GPU, polar_hip_simulate on its own frame range into device
// counters, polar_hip_allreduce_counters over its RCCL communicator. Run here as a world of one
// (the GPU box has one GPU): the all-reduce must leave the counters as they are, two shards must add up
// to the whole batch, and misuse must be refused. Exit code 0 and "shard counters ok" on success.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "polar_amd/polar.hh"

#define CHECK(x) \
	do { \
		if (!(x)) { \
			printf("failed: %s (%s)\n", #x, polar_hip_last_error()); \
			return 1; \
		} \
	} while (0)

int main()
{
	const int M = 10, N = 1 << M, K = N / 2;
	std::vector<uint8_t> frozen(N), program(2 * N + 1);
	PolarCodeConst0<M>()(frozen.data(), M, K, std::exp(-1.L));
	int length = PolarCompiler()(program.data(), frozen.data(), M);
	polar_hip_handle *h = nullptr;
	CHECK(polar_hip_create(&h, 0) == POLAR_OK);
	int id = polar_hip_load_program(h, program.data(), length);
	CHECK(id >= 0);
	ncclUniqueId uid;
	ncclComm_t comm;
	CHECK(ncclGetUniqueId(&uid) == ncclSuccess);
	CHECK(ncclCommInitRank(&comm, 1, uid, 0) == ncclSuccess);
	int64_t *dev = nullptr, whole[6], parts[6], host[6] = {0, 0, 0, 0, 0, 0};
	CHECK(hipMalloc((void **)&dev, sizeof whole) == hipSuccess);
	// the whole batch at once
	CHECK(hipMemset(dev, 0, sizeof whole) == hipSuccess);
	CHECK(polar_hip_simulate(h, id, frozen.data(), M, 1, -2.0, 99, 1000, 700, dev, POLAR_DEVICE) == POLAR_OK);
	CHECK(polar_hip_allreduce_counters(h, dev, 6, comm) == POLAR_OK);
	CHECK(hipMemcpy(whole, dev, sizeof whole, hipMemcpyDeviceToHost) == hipSuccess);
	// the same frames as two shards (what two ranks would do), summed
	CHECK(hipMemset(dev, 0, sizeof whole) == hipSuccess);
	CHECK(polar_hip_simulate(h, id, frozen.data(), M, 1, -2.0, 99, 1000, 300, dev, POLAR_DEVICE) == POLAR_OK);
	CHECK(polar_hip_simulate(h, id, frozen.data(), M, 1, -2.0, 99, 1300, 400, dev, POLAR_DEVICE) == POLAR_OK);
	CHECK(polar_hip_allreduce_counters(h, dev, 6, comm) == POLAR_OK);
	CHECK(hipMemcpy(parts, dev, sizeof parts, hipMemcpyDeviceToHost) == hipSuccess);
	CHECK(memcmp(whole, parts, sizeof whole) == 0);
	CHECK(whole[4] == 700 && whole[5] == 700 * (int64_t)K && whole[0] > 0 && whole[2] > 0);
	// misuse: host counters, no communicator
	CHECK(polar_hip_allreduce_counters(h, host, 6, comm) == POLAR_E_ARG);
	CHECK(polar_hip_allreduce_counters(h, dev, 6, nullptr) == POLAR_E_ARG);
	printf("awgn errors %lld, uncorrected %lld of %lld\n", (long long)whole[0], (long long)whole[2], (long long)whole[5]);
	ncclCommDestroy(comm);
	(void)hipFree(dev);
	polar_hip_destroy(h);
	printf("shard counters ok\n");
	return 0;
}
