/* inst_team_f64k18.hip -- instantiates the one-launch products of N = 2^15..2^17 (team_product_kernel) for (ArithF64, headroom class 18). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_TEAM_PRODUCT(ArithF64, 18)
} /* namespace ntt */
