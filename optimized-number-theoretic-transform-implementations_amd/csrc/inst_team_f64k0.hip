/* inst_team_f64k0.hip -- instantiates the one-launch products of N = 2^15..2^17 (team_product_kernel) for (ArithF64, headroom class 0). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_TEAM_PRODUCT(ArithF64, 0)
} /* namespace ntt */
