/* inst_dm_u64x_k3.hip -- instantiates the NTT-domain product kernels (dot_inv_kernel, fwd_mul_kernel) for ArithU64X<3>:
 * the products themselves are one Barrett reduction per 128-bit product (ArithU64X::dot_term / mul_out), the transform stages around them the wide policy's. */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_DOT(ArithU64X<3>, 3)
NTT_DEFINE_LAUNCH_FWD_MUL(ArithU64X<3>, 3)
} /* namespace ntt */
