// One of the translation units libindelpost_hip.so is built from: the k_dp_skew instantiations of IPX_DP_UNIT_L
// (csrc/ipx_kernels.h, end of file).  Split only to compile in parallel; nothing else lives here.
#define IPX_DP_TEMPLATES_ONLY 1
#include "ipx_kernels.h"
IPX_DP_UNIT_L(IPX_SKEW_DEFINE)
