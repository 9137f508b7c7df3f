/* inst_mul_u64.hip -- instantiates the forward-transform-times-b^ kernels (fwd_mul_kernel) for (ArithU64, headroom class 0). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_FWD_MUL(ArithU64, 0)
} /* namespace ntt */
