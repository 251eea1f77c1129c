// One of the translation units libindelpost_hip.so is built from: k_dp_skew in the 8-bit dialect, the plain-first forward stage
// (IPX_SKEW_BH_FAMILY, csrc/ipx_kernels.h, end of file).  Split only to compile in parallel; nothing else lives here.
#define IPX_DP_TEMPLATES_ONLY 1
#include "ipx_kernels.h"
IPX_SKEW_BH_FAMILY(IPX_SKEW_BH_DEFINE, false, 2)
