/*
 * ntt_host.hip -- host side of libntt_mi355x.so: plans, pass dispatch, the C ABI
 * declared in include/ntt_mi355x.h and the reference-signature entry points of
 * include/ntt_reference.h, ntt_radix4.h, ntt_radix4x4.h and ntt_seal.h.
 *
 * No CPU compute path exists in this library: every transform is a HIP kernel
 * launch and every failure is reported (status code / abort for the void
 * reference signatures).
 */
#include <algorithm>
#include <climits>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "ntt_kernels.h"
#include "ntt_passplan.h"
#include "ntt_tables.h"

#include "ntt_mi355x.h"
#include "ntt_radix4.h"
#include "ntt_radix4x4.h"
#include "ntt_reference.h"
#include "ntt_seal.h"

using namespace ntt;

namespace ntt {
/* defined in inst_*.hip */
template <> hipError_t launch_pass<ArithU64, 0>(const PassArgs &);
template <> hipError_t launch_pass<ArithU64R4, 0>(const PassArgs &);
template <> hipError_t launch_pass<ArithU64X<0>, 0>(const PassArgs &);
template <> hipError_t launch_pass<ArithU64X<1>, 1>(const PassArgs &);
template <> hipError_t launch_pass<ArithU64X<3>, 3>(const PassArgs &);
template <> hipError_t launch_pass<ArithF64, 0>(const PassArgs &);
template <> hipError_t launch_pass<ArithF64, 1>(const PassArgs &);
template <> hipError_t launch_pass<ArithF64, 18>(const PassArgs &);
template <> hipError_t launch_pass<ArithF64W, 0>(const PassArgs &);
template <> hipError_t launch_product<ArithF64, 0>(const ProdArgs &);
template <> hipError_t launch_product<ArithF64, 1>(const ProdArgs &);
template <> hipError_t launch_product<ArithF64, 18>(const ProdArgs &);
template <> hipError_t launch_product<ArithF64W, 0>(const ProdArgs &);
template <> hipError_t launch_team_product<ArithF64, 0>(const ProdArgs &);
template <> hipError_t launch_team_product<ArithF64, 1>(const ProdArgs &);
template <> hipError_t launch_team_product<ArithF64, 18>(const ProdArgs &);
template <> hipError_t launch_team_product<ArithF64W, 0>(const ProdArgs &);
template <> hipError_t launch_dot<ArithU64, 0>(const DotArgs &);
template <> hipError_t launch_dot<ArithF64, 0>(const DotArgs &);
template <> hipError_t launch_dot<ArithF64, 1>(const DotArgs &);
template <> hipError_t launch_dot<ArithF64, 18>(const DotArgs &);
template <> hipError_t launch_dot<ArithF64W, 0>(const DotArgs &);
template <> hipError_t launch_fwd_mul<ArithU64, 0>(const MulArgs &);
template <> hipError_t launch_dot<ArithU64X<0>, 0>(const DotArgs &);
template <> hipError_t launch_dot<ArithU64X<1>, 1>(const DotArgs &);
template <> hipError_t launch_dot<ArithU64X<3>, 3>(const DotArgs &);
template <> hipError_t launch_fwd_mul<ArithU64X<0>, 0>(const MulArgs &);
template <> hipError_t launch_fwd_mul<ArithU64X<1>, 1>(const MulArgs &);
template <> hipError_t launch_fwd_mul<ArithU64X<3>, 3>(const MulArgs &);
template <> hipError_t launch_fwd_mul<ArithF64, 0>(const MulArgs &);
template <> hipError_t launch_fwd_mul<ArithF64, 1>(const MulArgs &);
template <> hipError_t launch_fwd_mul<ArithF64, 18>(const MulArgs &);
template <> hipError_t launch_fwd_mul<ArithF64W, 0>(const MulArgs &);
} // namespace ntt

/* ------------------------------------------------------------------ */
/* errors                                                              */
/* ------------------------------------------------------------------ */
static thread_local std::string g_err;

static int fail(int code, const std::string &msg)
{
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if(e_ != hipSuccess) {                                                                    \
      return fail(NTT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));            \
    }                                                                                         \
  } while(0)

/* The environment is read for ONE purpose: the reference-signature entry points (void functions with the reference's argument
 * lists: no plan, no option argument) take their device and arithmetic from NTT_DEVICE / NTT_COMPAT_ARITH -- once, at their first
 * call (compat_config below).  Everything else is a plan option (ntt_plan_set_option). */

/* Every entry point selects the plan's (or the named) device for its own HIP calls and puts the
 * caller's current device back on return, so the library can be mixed with torch / other HIP code
 * that relies on hipSetDevice state. */
struct DeviceGuard {
  int  prev = -1;
  bool ok   = false;
  explicit DeviceGuard(int device)
  {
    if(hipGetDevice(&prev) != hipSuccess) prev = -1;
    ok = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard()
  {
    if(prev >= 0) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard &) = delete;
  DeviceGuard &operator=(const DeviceGuard &) = delete;
};
#define USE_DEVICE(device)                                             \
  DeviceGuard guard_(device);                                          \
  if(!guard_.ok) return fail(NTT_ERR_HIP, "hipSetDevice failed")

static int check_device(int device)
{
  int        n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if(e != hipSuccess || n <= 0) {
    return fail(NTT_ERR_NO_DEVICE, std::string("no usable HIP device (") +
                                     (e != hipSuccess ? hipGetErrorString(e) : "device count 0") +
                                     "); libntt_mi355x has no CPU fallback");
  }
  if(device < 0 || device >= n) return fail(NTT_ERR_ARG, "device index out of range");
  return NTT_OK;
}

/* the host layer by concern (round 6: one 3,200-line file before); ONE translation unit, sections included in dependency order */
#include "host/host_plan.inc"
#include "host/host_transforms.inc"
#include "host/host_products.inc"
#include "host/host_ntt_domain.inc"
#include "host/host_runtime.inc"
#include "host/host_compat.inc"
