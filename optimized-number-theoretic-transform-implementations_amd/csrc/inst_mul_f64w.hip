/* inst_mul_f64w.hip -- instantiates the forward-transform-times-b^ kernels (fwd_mul_kernel) for (ArithF64W, headroom class 0). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_FWD_MUL(ArithF64W, 0)
} /* namespace ntt */
