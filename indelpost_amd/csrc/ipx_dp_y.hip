// One of the translation units libindelpost_hip.so is built from: k_dp_skew at 16 lanes per read for the long classes of a big batch
// (26..32 segments at 8 lanes = 13..16 here), 16-bit passes, forward and reverse (IPX_W16_FAMILY, csrc/ipx_kernels.h, end of file).
// Split only to compile in parallel; nothing else lives here.
#define IPX_DP_TEMPLATES_ONLY 1
#include "ipx_kernels.h"
IPX_DP_UNIT_Y(IPX_W16_DEFINE)
