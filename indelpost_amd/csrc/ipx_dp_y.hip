// One of the translation units libindelpost_hip.so is built from: k_dp_wide, the 16-bit passes of reads beyond 504 bp as one wavefront per
// read (IPX_WIDE_FAMILY, csrc/ipx_kernels.h).  Split only to compile in parallel; nothing else lives here.
#define IPX_DP_TEMPLATES_ONLY 1
#include "ipx_kernels.h"
IPX_WIDE_FAMILY(IPX_WIDE_DEFINE)
