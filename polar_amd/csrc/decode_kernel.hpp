// decode_kernel.hpp -- the decode kernel template and the table entry of one compiled variant.
// The variants are instantiated in several translation units (variants.hip, one group each)
// so that they compile in parallel; polar_hip.hip collects the groups.
#pragma once

#include <hip/hip_runtime.h>

#include "sc_core.hpp"

using namespace polar;

struct DecodeArgs {
	const DevOp *prog;
	const uint8_t *cw;
	uint8_t *msg;
	u32 *scratch;          // per-block arenas, block_dwords each
	size_t block_dwords;
	size_t stride;         // bytes per row of cw: a multiple of 16, groups * 4Q <= stride
	size_t msg_stride;     // bytes per row of msg, likewise
	u32 groups;
	int top;
	int hot_top;           // scratch levels up to this one keep the default cache policy (sc_core.hpp, GVec::hot) ...
	u32 *hot;              // ... and live here: one block of hot_dwords per wavefront, side by side
	size_t hot_dwords;     // (Q << (hot_top + 1)) - (Q << (RL + 1)), 0 when there is no hot level
	unsigned long long *prof; // diagnostic build (POLAR_PROFILE) only
	// a set of jobs in one launch (team kernels, polar_hip_decode_set): `set` = set_jobs + 1 entries of this
	// struct in device memory, entry j = the arguments of job j and set_first = the position of its first team in
	// the common queue; the last entry only carries the total; set_counter = the queue's head (zero at launch).
	// 0 jobs: an ordinary launch.
	const DecodeArgs *set;
	u32 *set_counter;
	u32 set_jobs;
	u32 set_first;
};

// Persistent wavefronts: block b walks the groups b, b + gridDim, ... Blocks are dealt to
// the 8 XCDs round-robin, so within one sweep the blocks that share an XCD (equal b % 8) take
// neighbouring groups: those groups share the 128-byte lines of every input row.
// Occupancy hint: 4 waves per SIMD, which the variants with more than 10 KiB of LDS cannot reach; the
// compiler then budgets their registers by what the LDS allows. The pipelined variants (Group::PIPELINE)
// state their two waves per SIMD outright: a hard 256-register budget, else the compiler takes 260.
template <int Q, int RL>
__global__ void __launch_bounds__(64, (Group<Q, RL>::PIPELINE ? 2 : 4)) sc_decode_kernel(DecodeArgs a)
{
	typedef Group<Q, RL> G;
	__shared__ __attribute__((aligned(16))) u32 lds[G::LDS_DWORDS];
	const u32 nb = gridDim.x, b = blockIdx.x;
	const u32 slot = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
	u32 *arena = a.scratch ? a.scratch + (size_t)b * a.block_dwords : nullptr;
	const size_t plane = ((size_t)Q << a.top) / 8;
	G g;
	g.sl = (lds_u32 *)lds;
	g.hls = g.sl + G::LDS_SOFT;
	g.hlz = g.hls + G::LDS_PLANE;
	g.sg = arena;
	g.hgs = arena ? arena + ((size_t)Q << a.top) : nullptr;
	g.hgz = arena ? g.hgs + plane : nullptr;
	g.top = a.top;
	g.hot_top = a.hot_top;
	g.sh = a.hot ? a.hot + (size_t)b * a.hot_dwords - ((size_t)Q << (RL + 1)) : nullptr;
#ifdef POLAR_PROFILE
	g.prof = a.prof;
#endif
#if defined(POLAR_PROFILE) || defined(POLAR_SPAN)
	const unsigned long long t_start_ = __builtin_amdgcn_s_memtime();
#endif
	for (u32 base = 0; base < a.groups; base += nb) {
		u32 grp = base + slot;
		if (grp >= a.groups)
			continue;
#ifdef POLAR_PRIO_ALTERNATE
		// the SIMD favours the older of its two wavefronts (those of the first half of the grid): in every other
		// walk the younger one asks for the higher priority instead
		if (b >= nb / 2) {
			if ((base / nb) & 1)
				__builtin_amdgcn_s_setprio(1);
			else
				__builtin_amdgcn_s_setprio(0);
		}
#endif
		size_t g0 = (size_t)grp * (4 * Q);
		g.in = typename G::In{a.cw + g0, a.stride};
		g.out = OutRows<Q>{a.msg + g0, a.msg_stride};
		g.run((prog_ptr)a.prog);
	}
#if defined(POLAR_PROFILE) || defined(POLAR_SPAN)
	if (a.prof && threadIdx.x == 0 && b < 4096) { // diagnostic: when this wavefront started and ended (constant-rate counter)
		a.prof[1024 + 2 * b] = t_start_;
		a.prof[1024 + 2 * b + 1] = __builtin_amdgcn_s_memtime();
	}
#endif
}

// Team form: a workgroup of TEAM_WAVES wavefronts, wavefront w of block b takes the groups
// TEAM_WAVES * slot + w (the host pads the batch to whole teams), arena b * TEAM_WAVES + w.
// Two wavefronts per SIMD are asked for outright: left alone the compiler budgets the registers
// for one (253 + 4 accumulation registers) and halves the occupancy.
template <int Q, int RL, int TEAM_WAVES>
__global__ void __launch_bounds__(64 * TEAM_WAVES, 2) sc_decode_team_kernel(DecodeArgs a)
{
	typedef Group<Q, RL, true> G;
	__shared__ __attribute__((aligned(16))) u32 lds[TEAM_WAVES * G::LDS_DWORDS];
	const u32 nb = gridDim.x, b = blockIdx.x, w = threadIdx.x >> 6;
	const u32 slot = (nb % 8 == 0) ? (b % 8) * (nb / 8) + b / 8 : b;
	u32 *arena = a.scratch ? a.scratch + ((size_t)b * TEAM_WAVES + w) * a.block_dwords : nullptr;
	G g;
	g.sl = (lds_u32 *)lds + w * G::LDS_DWORDS;
	g.hls = g.sl + G::LDS_SOFT;
	g.hlz = g.hls + G::LDS_PLANE;
	g.sg = arena;
	g.hgs = arena ? arena + ((size_t)Q << a.top) : nullptr;
	g.hgz = arena ? g.hgs + ((size_t)Q << a.top) / 8 : nullptr;
	g.top = a.top;
	g.hot_top = a.hot_top;
	g.sh = a.hot ? a.hot + ((size_t)b * TEAM_WAVES + w) * a.hot_dwords - ((size_t)Q << (RL + 1)) : nullptr;
#ifdef POLAR_PROFILE
	g.prof = a.prof;
#endif
	// One launch, several jobs (polar_hip_decode_set): the teams of all jobs form one queue, longest walks first,
	// and every workgroup takes the next team from it when it has finished one (a counter in device memory,
	// handed to the four wavefronts through the first LDS word, which is dead between two walks); a workgroup
	// leaves when the queue is empty. Job j's arguments come from a table (scalar loads, like the program);
	// all jobs use this workgroup's arenas, sized for the largest code of the set.
	typedef const __attribute__((address_space(4))) DecodeArgs *table_ptr;
	const table_ptr tab = (table_ptr)a.set;
	const u32 jobs = a.set_jobs;
	for (u32 base = 0;; base += TEAM_WAVES * nb) {
		u32 grp;
		if (jobs) {
			__syncthreads(); // every wavefront is done with the LDS of its last walk
			if (threadIdx.x == 0)
				lds[0] = atomicAdd(a.set_counter, 1u);
			__syncthreads();
			const u32 unit = (u32)__builtin_amdgcn_readfirstlane((int)lds[0]);
			__syncthreads(); // ... and has read the word before the walk overwrites it
			if (unit >= tab[jobs].set_first)
				break;
			u32 j = 0;
			for (u32 k = 1; k < jobs; ++k)
				j += unit >= tab[k].set_first;
			a.prog = tab[j].prog;
			a.cw = tab[j].cw;
			a.msg = tab[j].msg;
			a.stride = tab[j].stride;
			a.msg_stride = tab[j].msg_stride;
			a.top = tab[j].top;
			g.top = a.top;
			g.hgs = arena + ((size_t)Q << a.top);
			g.hgz = g.hgs + ((size_t)Q << a.top) / 8;
			grp = (unit - tab[j].set_first) * TEAM_WAVES + w;
		} else {
			if (base >= a.groups)
				break;
			if (base + TEAM_WAVES * slot >= a.groups) // whole teams only (a.groups is a multiple of TEAM_WAVES)
				continue;
			grp = base + TEAM_WAVES * slot + w;
		}
		size_t g0 = (size_t)grp * (4 * Q);
		g.in = typename G::In{a.cw + g0, a.stride};
		g.out = OutRows<Q>{a.msg + g0, a.msg_stride};
		g.run((prog_ptr)a.prog);
	}
}

typedef void (*decode_kernel_t)(DecodeArgs);

struct KernelVariant {
	int q, rl;
	decode_kernel_t fn;
	size_t lds_bytes;
	int chain_depth;
	int team; // wavefronts per workgroup: 1, or the team size of the team form
};

enum { VARIANT_GROUPS = 9 };
// the variants compiled into group g (variants.hip with -DVARIANT_GROUP=g)
#define DECLARE_GROUP(g) const KernelVariant *polar_variant_group_##g(int *count);
DECLARE_GROUP(0) DECLARE_GROUP(1) DECLARE_GROUP(2) DECLARE_GROUP(3) DECLARE_GROUP(4) DECLARE_GROUP(5) DECLARE_GROUP(6) DECLARE_GROUP(7) DECLARE_GROUP(8)
#undef DECLARE_GROUP
