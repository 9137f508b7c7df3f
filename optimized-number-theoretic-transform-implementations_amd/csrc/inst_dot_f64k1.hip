/* inst_dot_f64k1.hip -- instantiates the NTT-domain product kernels (dot_inv_kernel) for (ArithF64, headroom class 1). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_DOT(ArithF64, 1)
} /* namespace ntt */
