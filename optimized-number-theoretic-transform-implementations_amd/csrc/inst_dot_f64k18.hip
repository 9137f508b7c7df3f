/* inst_dot_f64k18.hip -- instantiates the NTT-domain product kernels (dot_inv_kernel) for (ArithF64, headroom class 18). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_DOT(ArithF64, 18)
} /* namespace ntt */
