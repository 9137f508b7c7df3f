/* inst_u64x_k3.hip -- instantiates every fused/column kernel for ArithU64X<3> (integer policy, headroom class 3:
 * 64 multiples of q below 2^64). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS(ArithU64X<3>, 3)
} /* namespace ntt */
