/* inst_u64.hip -- instantiates every fused/column kernel for (ArithU64, headroom class 0). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS(ArithU64, 0)
} /* namespace ntt */
