/* inst_u64r4.hip -- instantiates the fused kernels for ArithU64R4 (the reference's radix-4 butterflies);
 * the policy has no column-pass form. */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS_FUSED_ONLY(ArithU64R4, 0)
} /* namespace ntt */
