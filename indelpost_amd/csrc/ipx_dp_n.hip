// One of the translation units libindelpost_hip.so is built from: the k_dp_pass instantiations of IPX_VL2_FAMILY
// (csrc/ipx_kernels.h, end of file).  Split only to compile in parallel; nothing else lives here.
#define IPX_DP_TEMPLATES_ONLY 1
#include "ipx_kernels.h"
IPX_VL2_FAMILY(IPX_VL2_DEFINE)
