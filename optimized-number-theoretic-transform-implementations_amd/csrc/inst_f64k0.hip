/* inst_f64k0.hip -- instantiates every fused/column kernel for (ArithF64, headroom class 0). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS(ArithF64, 0)
NTT_DEFINE_LAUNCH_PRODUCT(ArithF64, 0)
} /* namespace ntt */
