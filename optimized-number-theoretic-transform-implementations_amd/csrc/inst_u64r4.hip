/* inst_u64r4.hip -- instantiates the kernels of ArithU64R4 (the reference's radix-4 butterflies): the fused block kernels
 * and the column passes of one or two radix-4 levels that precede (forward) / follow (inverse) them for N > 2^14. */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS_RADIX4(ArithU64R4, 0)
} /* namespace ntt */
