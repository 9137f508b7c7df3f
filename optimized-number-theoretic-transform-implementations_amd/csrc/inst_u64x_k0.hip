/* inst_u64x_k0.hip -- instantiates every fused/column kernel for ArithU64X<0> (integer policy, headroom class 0:
 * 8 multiples of q below 2^64). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS(ArithU64X<0>, 0)
} /* namespace ntt */
