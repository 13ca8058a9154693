// variants.hip -- one group of decode-kernel variants per translation unit (-DVARIANT_GROUP=0..7).
//   256 frames per wavefront: every node keeps all 64 lanes busy; ~10 KB of LDS per wavefront
//   (15 per CU): (64,6) (32,7); ~20 KB (7-8 per CU): the default family (128,6) ... (4,11);
//   ~40 KB (3 per CU): (16,10) (16,8). Team forms (four wavefronts per workgroup): (32,8) (64,7).
#include "decode_kernel.hpp"

#define VARIANT(q, rl) {q, rl, sc_decode_kernel<q, rl>, sizeof(u32) * (size_t)Group<q, rl>::LDS_DWORDS, Group<q, rl>::CHAIN_DEPTH, 1}
#define TEAM_VARIANT(q, rl, t) {q, rl, sc_decode_team_kernel<q, rl, t>, sizeof(u32) * (size_t)Group<q, rl>::LDS_DWORDS * t, Group<q, rl>::CHAIN_DEPTH, t}
static const KernelVariant table[] = {
#if VARIANT_GROUP == 0
	VARIANT(32, 6), VARIANT(1, 11),
#elif VARIANT_GROUP == 1
	VARIANT(64, 4), VARIANT(64, 5),
#elif VARIANT_GROUP == 2
	VARIANT(64, 6), VARIANT(2, 10),
#elif VARIANT_GROUP == 3
	VARIANT(32, 5), VARIANT(4, 9),
#elif VARIANT_GROUP == 4
	VARIANT(32, 7), VARIANT(4, 8),
#elif VARIANT_GROUP == 5
	VARIANT(16, 6), VARIANT(8, 8),
#elif VARIANT_GROUP == 6
	VARIANT(8, 7), VARIANT(16, 7),
#elif VARIANT_GROUP == 7
	VARIANT(4, 10),
#elif VARIANT_GROUP == 8
	TEAM_VARIANT(8, 8, 4), TEAM_VARIANT(16, 7, 4),
#endif
};
#undef VARIANT
#undef TEAM_VARIANT

#define GROUP_FN2(g) polar_variant_group_##g
#define GROUP_FN(g) GROUP_FN2(g)
const KernelVariant *GROUP_FN(VARIANT_GROUP)(int *count)
{
	*count = (int)(sizeof(table) / sizeof(table[0]));
	return table;
}
