/* inst_f64w.hip -- instantiates the kernels for ArithF64W, the FP64 policy for moduli between 2^51(1+2^-10) and 2^52
 * (every butterfly reduces both operands; the headroom-class parameter is unused: class 0). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS(ArithF64W, 0)
NTT_DEFINE_LAUNCH_PRODUCT(ArithF64W, 0)
} /* namespace ntt */
