/* inst_dot_u64.hip -- instantiates the NTT-domain product kernels (dot_inv_kernel) for (ArithU64, headroom class 0). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_DOT(ArithU64, 0)
} /* namespace ntt */
