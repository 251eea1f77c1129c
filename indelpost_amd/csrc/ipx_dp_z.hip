// One of the translation units libindelpost_hip.so is built from: k_dp_band_rev, the 16-bit reverse pass as a band, one lane per pair of
// reads (IPX_BAND_FAMILY, csrc/ipx_kernels.h).  Split only to compile in parallel; nothing else lives here.
#define IPX_DP_TEMPLATES_ONLY 1
#include "ipx_kernels.h"
IPX_BAND_FAMILY(IPX_BAND_DEFINE)
IPX_BAND8_FAMILY(IPX_BAND8_DEFINE)
