/* inst_u64x_k1.hip -- instantiates every fused/column kernel for ArithU64X<1> (integer policy, headroom class 1:
 * 16 multiples of q below 2^64). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS(ArithU64X<1>, 1)
} /* namespace ntt */
