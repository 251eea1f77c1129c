// One of the translation units libindelpost_hip.so is built from: the tier kernels (k_dp_skew_tier: several segLen classes of one
// occupancy in one launch), 16-bit forward passes (csrc/ipx_kernels.h, end of file).  Split only to compile in parallel; nothing else lives here.
#define IPX_DP_TEMPLATES_ONLY 1
#include "ipx_kernels.h"
IPX_TIER_WORD(IPX_TIER_DEFINE, false)
