/* inst_dot_f64w.hip -- instantiates the NTT-domain product kernels (dot_inv_kernel) for (ArithF64W, headroom class 0). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_DOT(ArithF64W, 0)
} /* namespace ntt */
