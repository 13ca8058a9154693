// polar_hip.hip -- kernels and C ABI of libpolar_hip.so (gfx950 only).
//
//   sc_decode_kernel        : batched SC decoder, one wavefront per group of 4*Q frames (decode_kernel.hpp,
//                             sc_core.hpp, variants.hip; replaces PolarDecoder::operator(), polar_decoder.hh:131)
//   sc_decode_team_kernel   : the same with four wavefronts per workgroup that share the input lines (32- and 64-frame groups)
//   sc_decode_f32_kernel    : the reference's float helper per lane (polar_helper.hh:62-111)
//   packed_tile_kernel      : polar transform on bit-packed symbols through LDS tiles: the encoders (polar_encoder.hh:15-58)
//                             and the systematic extraction = re-encode + gather (testbench.cc:177-183)
//   packed_transform_kernel : the same in registers only (codes below N = 2^6)
//   butterfly_kernel        : transform stages on byte symbols (encoders on ragged batches)
//   channel_kernel, count_kernel : the bench's message source, AWGN channel and error counters (testbench.cc:112-192)
//
// One translation unit in parts, included at the end of this file in this order:
//   kernels_float.hpp, kernels_transform.hpp, kernels_channel.hpp   the kernels other than the decode kernel
//   handle.hpp          the handle, resident programs, code cache, create / destroy / configure, shape choice
//   decode_launch.hpp   launch of the decode kernel (arenas, hot levels, ragged batches)
//   host_pipeline.hpp   host-buffer paths (page-locked buffers, chunked copy / decode / copy)
//   abi_*.hpp           the extern "C" entry points of include/polar_hip.h by family
//
// Everything here fails loudly: there is no CPU fallback for any device entry point.
#include <hip/hip_runtime.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/polar_hip.h"
#include "construct.hpp"
#include "program.hpp"
#include "channel.hpp"
#include "decode_kernel.hpp"

using namespace polar;

static thread_local std::string g_error;

static int fail(int code, const std::string &msg)
{
	g_error = msg;
	return code;
}

#define HIP_TRY(expr) \
	do { \
		hipError_t e_ = (expr); \
		if (e_ != hipSuccess) \
			return fail(e_ == hipErrorOutOfMemory ? POLAR_E_NOMEM : POLAR_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

// ---------------------------------------------------------------------------------------
// decoder kernel
// ---------------------------------------------------------------------------------------

// all compiled decode-kernel variants (variants.hip, one group per translation unit)
struct VariantTable {
	std::vector<KernelVariant> v;
	VariantTable()
	{
		typedef const KernelVariant *(*group_fn)(int *);
		static const group_fn groups[VARIANT_GROUPS] = {polar_variant_group_0, polar_variant_group_1, polar_variant_group_2, polar_variant_group_3,
			polar_variant_group_4, polar_variant_group_5, polar_variant_group_6, polar_variant_group_7, polar_variant_group_8};
		for (group_fn g : groups) {
			int n = 0;
			const KernelVariant *t = g(&n);
			v.insert(v.end(), t, t + n);
		}
	}
};
static const std::vector<KernelVariant> &variants()
{
	static const VariantTable t;
	return t.v;
}

static const KernelVariant *find_variant(int q, int rl, int team = 1)
{
	for (const KernelVariant &v : variants())
		if (v.q == q && v.rl == rl && v.team == team)
			return &v;
	return nullptr;
}

#include "kernels_float.hpp"
#include "kernels_transform.hpp"
#include "kernels_channel.hpp"
#include "handle.hpp"
#include "decode_launch.hpp"
#include "host_pipeline.hpp"
#include "abi_decode.hpp"
#include "abi_set.hpp"
#include "abi_encode.hpp"
#include "abi_simulate.hpp"
#include "abi_float.hpp"
#include "abi_shard.hpp"
