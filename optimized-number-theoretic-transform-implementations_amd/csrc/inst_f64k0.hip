/* inst_f64k0.hip -- instantiates every fused/column kernel for (ArithF64, headroom class 0). */
#include "ntt_kernels.h"

namespace ntt {
NTT_DEFINE_LAUNCH_PASS(ArithF64, 0)
NTT_DEFINE_LAUNCH_PRODUCT(ArithF64, 0)
} /* namespace ntt */

#ifdef NTT_STAMPS
/* diagnostic build only: read back the stamps of THIS translation unit's kernels */
extern "C" __attribute__((visibility("default"))) int ntt_debug_stamps(unsigned long long *out, int reset)
{
  if(reset) {
    void *sym = nullptr;
    if(hipGetSymbolAddress(&sym, HIP_SYMBOL(ntt::g_stamps)) != hipSuccess) return -1;
    return hipMemset(sym, 0, sizeof(ntt::g_stamps)) == hipSuccess ? 0 : -1;
  }
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(ntt::g_stamps), sizeof(ntt::g_stamps)) == hipSuccess ? 0 : -1;
}
#endif

