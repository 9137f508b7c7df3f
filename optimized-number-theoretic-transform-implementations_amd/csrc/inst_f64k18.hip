/* inst_f64k18.hip -- instantiates every fused/column kernel for (ArithF64, headroom class 18). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS(ArithF64, 18)
NTT_DEFINE_LAUNCH_PRODUCT(ArithF64, 18)
} /* namespace ntt */
