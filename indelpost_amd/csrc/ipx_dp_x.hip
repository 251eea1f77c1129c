// One of the translation units libindelpost_hip.so is built from: the latency tier of k_dp_skew (32 lanes per read, small batches),
// 16-bit passes and the plain 8-bit recurrence, forward and reverse (IPX_LAT_FAMILY, csrc/ipx_kernels.h, end of file).  Split only to
// compile in parallel; nothing else lives here.
#define IPX_DP_TEMPLATES_ONLY 1
#include "ipx_kernels.h"
IPX_DP_UNIT_X(IPX_LAT_DEFINE)
