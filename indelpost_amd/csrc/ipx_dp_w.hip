// One of the translation units libindelpost_hip.so is built from: k_dp_pass_tier (the stepped 8-bit passes of classes 1..16 in one
// launch), reverse pass (csrc/ipx_kernels.h, end of file).  Split only to compile in parallel; nothing else lives here.
#define IPX_DP_TEMPLATES_ONLY 1
#include "ipx_kernels.h"
IPX_PASS_TIER_DEFINE(true)
