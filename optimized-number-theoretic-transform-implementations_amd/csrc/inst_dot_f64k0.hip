/* inst_dot_f64k0.hip -- instantiates the NTT-domain product kernels (dot_inv_kernel) for (ArithF64, headroom class 0). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_DOT(ArithF64, 0)
} /* namespace ntt */
