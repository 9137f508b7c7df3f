/* inst_f64k1.hip -- instantiates every fused/column kernel for (ArithF64, headroom class 1). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS(ArithF64, 1)
NTT_DEFINE_LAUNCH_PRODUCT(ArithF64, 1)
} /* namespace ntt */
