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

/* ------------------------------------------------------------------ */
/* on-device table construction                             */
/* ------------------------------------------------------------------ */
/*
 * The reference builds its tables on the host with % and / on 128-bit integers (tests/test_cases.h:212-311,
 * include/internal/pre_compute.h:38-105: "we don't care about the performance").  A plan for N = 2^17 needs
 * 2 x 2^17 modular powers plus one 128-by-64-bit division (integer policies) or one FP64 division (FP64 policy)
 * per entry -- milliseconds on one host core per plan, times primes, times GPUs.  Here the host only squares
 * the root log2 N times; every table entry is produced by one GPU thread:
 *   w[k] = root^bitrev(k) = product of root^(2^j) over the set bits j of bitrev(k)   (reference layout, :38-51)
 *   con  = floor(w * 2^64 / q) by 64 steps of shift-and-subtract (exact)                 (:68-77)
 *   FP64 : balanced w and its correctly rounded quotient by q (v_div: IEEE division)
 *   radix-4 expanded table e[2k] = w[k], e[4k+1] = w[k] w[2k], e[4k+3] = q - w[k] w[2k+1]  (:85-105)
 * Caller-supplied tables (ntt_plan_create_from_tables, the reference-signature entry points) still take the host
 * route: their entries are data, not something to regenerate.
 */
struct PowBasis {
  uint64_t p[32]; /* base^(2^j) mod q */
};

__device__ __forceinline__ uint64_t dev_precon64(uint64_t w, uint64_t q)
{
  uint64_t r = w, con = 0; /* w < q < 2^61: 2r never overflows */
  for(int i = 0; i < 64; i++) {
    r <<= 1;
    const uint64_t ge = r >= q;
    r -= ge ? q : 0;
    con = (con << 1) | ge;
  }
  return con;
}

__device__ __forceinline__ uint64_t dev_brev(uint64_t v, unsigned bits) { return bits ? (__brevll(v) >> (64 - bits)) : 0; }

__global__ void __launch_bounds__(256) power_table_kernel(uint64_t *w, uint64_t N, unsigned m, PowBasis basis, ArithU64::consts c)
{
  for(uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < N; k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t e = dev_brev(k, m);
    uint64_t       acc = 1 % c.q;
    for(unsigned j = 0; j < m; j++) {
      if((e >> j) & 1) acc = ArithU64::mulmod_full(acc, basis.p[j], c);
    }
    w[k] = acc;
  }
}

/* records N .. N+15 behind an inverse table: N^-1 * winv[k] (run_group0_folded) */
__device__ __forceinline__ uint64_t folded_word(const uint64_t *w, uint64_t N, uint64_t k, uint64_t ninv, const ArithU64::consts &c)
{
  if(k < N) return w[k];
  const uint64_t j = k - N;
  return j < N ? ArithU64::mulmod_full(ninv, w[j], c) : ninv;
}

__global__ void __launch_bounds__(256) records_u64_kernel(TwU64 *out, const uint64_t *w, uint64_t N, uint64_t total, uint64_t ninv,
                                                          ArithU64::consts c)
{
  for(uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t v = folded_word(w, N, k, ninv, c);
    out[k]           = TwU64{v, dev_precon64(v, c.q)};
  }
}

__global__ void __launch_bounds__(256) records_f64_kernel(TwF64 *out, double *out8, const uint64_t *w, uint64_t N, uint64_t total,
                                                          uint64_t ninv, ArithU64::consts c)
{
  const double qd = (double)c.q; /* q < 2^52: exact */
  for(uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t v  = folded_word(w, N, k, ninv, c);
    const double   wb = v > c.q / 2 ? -(double)(c.q - v) : (double)v;
    out[k]            = TwF64{wb, wb / qd};
    if(k < N) out8[k] = wb;
  }
}

__global__ void __launch_bounds__(256) records_r4_kernel(TwU64 *out, const uint64_t *w, uint64_t N, ArithU64::consts c)
{
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * N; i += (uint64_t)gridDim.x * blockDim.x) {
    uint64_t v;
    if((i & 1) == 0) {
      v = w[i >> 1];
    } else {
      const uint64_t k = i >> 2;
      if(k == 0) {
        v = 0; /* slots 1 and 3 stay 0 as in the reference (pre_compute.h:90-93) */
      } else if((i & 3) == 1) {
        v = ArithU64::mulmod_full(w[k], w[2 * k], c);
      } else {
        v = c.q - ArithU64::mulmod_full(w[k], w[2 * k + 1], c);
      }
    }
    out[i] = TwU64{v, dev_precon64(v, c.q)};
  }
}

/* ------------------------------------------------------------------ */
/* plan                                                                */
/* ------------------------------------------------------------------ */
constexpr int kWideClass = -1; /* ntt_plan::kcls of the FP64 policy for 2^51(1+2^-10) < q < 2^52 (ArithF64W) */

struct ntt_plan {
  int      device  = 0;
  uint64_t N       = 0, q = 0, root = 0;
  int      m       = 0;
  int      arith   = NTT_ARITH_U64; /* resolved: U64, F64 or U64_R4 */
  int      kcls    = 0;             /* instantiated FP64 headroom class */
  int      int_cls = -1;            /* integer policy: -1 = the reference's butterflies (ArithU64), else the headroom class K of
                                     * ArithU64X<K> the transforms run with (same tables, same canonical results) */
  bool     generic = false;
  bool     has_fwd = false, has_inv = false;
  void *   d_fwd   = nullptr; /* U64/F64: N records (+16 folded N^-1 records, inverse); U64_R4: 2N expanded records */
  void *   d_inv   = nullptr;
  void *   d_fwd8  = nullptr; /* compact forward twiddles (FP64 policy) */
  void *   d_inv8  = nullptr; /* compact inverse twiddles (FP64 policy) */
  ArithU64::consts cu{};
  F64Consts        cf{};
  std::vector<unsigned char> limbrec; /* this plan's LimbRec<A> (table pointers + constants): copied into the kernel arguments of every launch */
  std::vector<unsigned char> limbrec_mid; /* radix-4 policy: the same with the record of 1 in place of N^-1 -- what the block pass of a
                                           * two-pass INVERSE multiplies by (the reference's N^-1 pass, src/ntt_radix4.c:111-113, comes
                                           * once, after the last level: in the column pass's store) */
  /* XCD-local two-pass launches (team_kernel, N = 2^15..2^17): queue heads and per-polynomial counters in device memory,
   * one buffer per stream the plan is used on (launches on one stream are ordered; two streams must not share counters) */
  int        xcd_local = -1; /* 1 on, 0 off, -1 automatic */
  int        team_lag = 0, team_wpc = 0; /* 0 = the kernel's defaults */
  struct TeamBuf {
    void * stream;
    void * d;       /* block of the direct launches on this stream */
    size_t bytes;
    void * g;       /* block of the launches CAPTURED on this stream: its address is baked into graph nodes, so it is never
                     * freed, regrown or shared with direct launches before the plan is destroyed */
    size_t gbytes;
  };
  /* The control blocks are the ONE piece of plan state a batched call (const ntt_plan *) may touch: mutable, every access
   * under team_mu.  ntt_plan_reserve / NTT_OPT_MAX_BATCH_HINT size them ahead of time so that no batched call allocates. */
  mutable std::vector<TeamBuf> team_bufs;
  mutable std::vector<void *>  team_retired; /* outgrown direct blocks: launches still queued may use them, freed with the plan */
  mutable std::mutex           team_mu;
  uint64_t                     batch_hint = 0; /* NTT_OPT_MAX_BATCH_HINT: polynomials x limbs of the largest call; blocks of new streams start at this size */
  mutable uint64_t             ctl_allocs = 0; /* hipMalloc calls team_buffer has made for this plan (NTT_OPT_CTL_ALLOCATIONS, read-only) */
  /* pointer batches handed over as HOST arrays (ntt_transform_ptrs): the sorted addresses travel to the device through a pinned
   * staging buffer and live in a device table, one pair per stream the plan is used on (stream order makes the reuse safe: the next
   * call's copy queues behind the previous call's kernels); grown by retiring, like the control blocks; under team_mu */
  struct PtrBuf {
    void *     stream;
    uint64_t * d;      /* device table */
    uint64_t * h;      /* pinned host staging */
    size_t     words;
    hipEvent_t copied; /* the last upload from h has completed (h may be rewritten) */
  };
  mutable std::vector<PtrBuf> ptr_bufs;
  mutable std::vector<std::pair<void *, void *>> ptr_retired; /* {device, pinned host} */
  hipStream_t      own_stream = nullptr; /* used by ntt_batch_multi */
  int              max_grid   = 0;
  int              rns_launch = -1; /* ntt_rns_*: 0 = one launch over a run of limbs wherever it is built, 1 = one launch chain per limb,
                                     * -1 = where it pays (rns_one_launch_pays); read from the run's first plan */
  int              dot_fused  = 1;  /* the NTT-domain product kernels (dot_inv_kernel, fwd_mul_kernel): 0 = pointwise launches + transform */
  int              block_oversub = 0; /* persistent block kernels: workgroups per resident slot (0 = the kernels' defaults) */
  int              num_cus    = 256;
  int              chunk_mib  = 256; /* bytes of one multi-pass chunk (Infinity Cache residency) */
  int              block_log  = 0;     /* multi-pass transforms: block size below the column passes (0 = multi_pass_block's choice) */
  int              fused_product = 1; /* N = 2^8..2^17, FP64: ntt_negacyclic_mul_batch through the fused product kernels (0: four-launch
                                       * chain; 2: as 1, but a's forward transform always as a launch of its own) */
  int              one_pass   = -1;    /* 2^15, FP64 policies: the transform in ONE pass, the polynomial in the registers of one workgroup
                                        * (onepass_kernel): 1 on, 0 off, -1 = batches that give every CU a polynomial */
  int              two_phase  = -1;    /* 2^16, 2^17: both passes of a polynomial inside one workgroup (twophase_kernel):
                                        * 1 on, 0 off, -1 where it measured faster (forward 2^16, scheduled FP64 policy: +3 %) */
};

static bool is_pow2(uint64_t n) { return n && !(n & (n - 1)); }
static bool h_is_prime(uint64_t n); /* deterministic Miller-Rabin (below) */

/* headroom class of ArithU64X for q (ntt_arith.h): B = 8 * 2^K multiples of q below 2^64; -1 = not served (reduce_any
 * wants q >= 2^40 -- smaller moduli belong to the FP64 policies anyway -- and 8q < 2^64) */
static int int_wide_class(uint64_t q)
{
  if(q < (1ull << 40) || q >= (1ull << 61)) return -1;
  return q < (1ull << 58) ? 3 : (q < (1ull << 60) ? 1 : 0);
}

static int resolve_arith(int requested, uint64_t q, int m, int *out)
{
  /* FP64: the scheduled policy up to 2^51(1+2^-10), the reduce-both-operands policy (ArithF64W) up to 2^52 */
  if(requested == NTT_ARITH_AUTO) requested = (h_f64_eligible(q) || h_f64w_eligible(q)) ? NTT_ARITH_F64 : NTT_ARITH_U64;
  if(requested == NTT_ARITH_F64 && !h_f64_eligible(q) && !h_f64w_eligible(q)) {
    return fail(NTT_ERR_UNSUPPORTED, "FP64 arithmetic needs q < 2^52");
  }
  if(requested == NTT_ARITH_U64_R4 && (m < kFusedMin || m > kRadix4Max)) {
    return fail(NTT_ERR_UNSUPPORTED, "the radix-4 policy covers 2^6..2^18");
  }
  if(requested == NTT_ARITH_U64_R4 && q >= (1ull << 60)) return fail(NTT_ERR_UNSUPPORTED, "radix-4 lazy range needs 16q < 2^64");
  if(requested != NTT_ARITH_F64 && requested != NTT_ARITH_U64 && requested != NTT_ARITH_U64_R4) return fail(NTT_ERR_ARG, "bad arith");
  *out = requested;
  return NTT_OK;
}

template <class TW> static int upload_records(void **d_out, const std::vector<TW> &host)
{
  HIP_TRY(hipMalloc(d_out, host.size() * sizeof(TW)));
  HIP_TRY(hipMemcpy(*d_out, host.data(), host.size() * sizeof(TW), hipMemcpyHostToDevice));
  return NTT_OK;
}

template <class TW, class MK>
static int upload_table(void **d_out, const std::vector<uint64_t> &w, uint64_t q, MK mk)
{
  std::vector<TW> host(w.size());
  for(size_t i = 0; i < w.size(); i++) host[i] = mk(w[i], q);
  return upload_records(d_out, host);
}

/* integer records {w, con}: con from the caller's own precomputation when given (the reference passes
 * w_con next to every table, tests/test_cases.h:226-251), else floor(w * 2^64 / q) */
static int upload_u64(void **d_out, const std::vector<uint64_t> &w, const std::vector<uint64_t> &con, uint64_t q)
{
  std::vector<TwU64> host(w.size());
  for(size_t i = 0; i < w.size(); i++) host[i] = i < con.size() ? TwU64{w[i], con[i]} : h_tw_u64(w[i], q);
  return upload_records(d_out, host);
}

/* What a plan is built from: radix-2 power tables in bit-reversed order (either may be empty), optionally
 * the caller's Shoup precomputation for them, and -- radix-4 policy built from caller tables -- the 2N-entry
 * expanded tables as they are (pre_compute.h:85-105). */
struct TableSet {
  std::vector<uint64_t> fwd, inv, fwd_con, inv_con;
  std::vector<uint64_t> efwd, einv, efwd_con, einv_con;
  uint64_t gen_root = 0, gen_root_inv = 0; /* both non-zero: generate every table on the device from these roots */
};

/* one direction's tables of plan p from `base` (root or its inverse), on the device */
static int device_build_direction(ntt_plan *p, uint64_t base, bool inverse, uint64_t ninv)
{
  const uint64_t N = p->N, q = p->q;
  PowBasis       basis{};
  uint64_t       sq = base % q;
  for(int j = 0; j < p->m && j < 32; j++) {
    basis.p[j] = sq;
    sq         = h_mulmod(sq, sq, q);
  }
  /* a private non-blocking stream: the build neither waits for nor stalls the caller's streams (the legacy NULL stream
   * + hipDeviceSynchronize of round 2 did both).  Plans must still be created outside stream capture: hipMalloc and
   * the synchronisation below are not capturable. */
  hipStream_t bs = nullptr;
  HIP_TRY(hipStreamCreateWithFlags(&bs, hipStreamNonBlocking));
  uint64_t *d_w = nullptr;
  {
    hipError_t em = hipMalloc((void **)&d_w, N * sizeof(uint64_t));
    if(em != hipSuccess) {
      (void)hipStreamDestroy(bs);
      return fail(em == hipErrorOutOfMemory ? NTT_ERR_NOMEM : NTT_ERR_HIP, std::string("device table build: ") + hipGetErrorString(em));
    }
  }
  const unsigned g = (unsigned)((N + 255) / 256 > 4096 ? 4096 : (N + 255) / 256);
  hipLaunchKernelGGL(power_table_kernel, dim3(g), dim3(256), 0, bs, d_w, N, (unsigned)p->m, basis, p->cu);
  int         rc    = NTT_OK;
  void **     d_rec = inverse ? &p->d_inv : &p->d_fwd;
  void **     d_cmp = inverse ? &p->d_inv8 : &p->d_fwd8;
  hipError_t  e     = hipSuccess;
  if(p->arith == NTT_ARITH_U64_R4) {
    e = hipMalloc(d_rec, 2 * N * sizeof(TwU64));
    if(e == hipSuccess) hipLaunchKernelGGL(records_r4_kernel, dim3(g), dim3(256), 0, bs, (TwU64 *)*d_rec, d_w, N, p->cu);
  } else {
    const uint64_t total = inverse ? N + 16 : N;
    if(p->arith == NTT_ARITH_F64) {
      e = hipMalloc(d_rec, total * sizeof(TwF64));
      if(e == hipSuccess) e = hipMalloc(d_cmp, N * sizeof(double));
      if(e == hipSuccess)
        hipLaunchKernelGGL(records_f64_kernel, dim3(g), dim3(256), 0, bs, (TwF64 *)*d_rec, (double *)*d_cmp, d_w, N, total, ninv, p->cu);
    } else {
      e = hipMalloc(d_rec, total * sizeof(TwU64));
      if(e == hipSuccess) hipLaunchKernelGGL(records_u64_kernel, dim3(g), dim3(256), 0, bs, (TwU64 *)*d_rec, d_w, N, total, ninv, p->cu);
    }
  }
  if(e == hipSuccess) e = hipGetLastError();
  {
    const hipError_t es = hipStreamSynchronize(bs); /* always: d_w is freed below */
    if(e == hipSuccess) e = es;
  }
  (void)hipStreamDestroy(bs);
  (void)hipFree(d_w);
  if(e != hipSuccess) rc = fail(e == hipErrorOutOfMemory ? NTT_ERR_NOMEM : NTT_ERR_HIP, std::string("device table build: ") + hipGetErrorString(e));
  return rc;
}

/* the plan's LimbRec as raw bytes (host copy): the layout the kernels read (ntt_kernels.h) */
static std::vector<unsigned char> limbrec_bytes(const ntt_plan *p)
{
  std::vector<unsigned char> b;
  if(p->arith == NTT_ARITH_F64) {
    LimbRec<ArithF64> r{};
    r.tw_f  = static_cast<const TwF64 *>(p->d_fwd);
    r.tw8_f = static_cast<const double *>(p->d_fwd8);
    r.tw_i  = static_cast<const TwF64 *>(p->d_inv);
    r.tw8_i = static_cast<const double *>(p->d_inv8);
    r.c     = p->cf;
    b.resize(sizeof r);
    memcpy(b.data(), &r, sizeof r);
  } else {
    LimbRec<ArithU64> r{};
    r.tw_f = static_cast<const TwU64 *>(p->d_fwd);
    r.tw_i = static_cast<const TwU64 *>(p->d_inv);
    r.c    = p->cu;
    b.resize(sizeof r);
    memcpy(b.data(), &r, sizeof r);
  }
  return b;
}
static_assert(sizeof(LimbRec<ArithU64>) == sizeof(LimbRec<ArithU64R4>) && sizeof(LimbRec<ArithF64>) == sizeof(LimbRec<ArithF64W>),
              "the policies of one family share a record layout");


/* ninv_override: 0 = derive N^-1 */
static int plan_build(ntt_plan **out, int device, uint64_t N, uint64_t q, uint64_t root, const TableSet &ts, int arith,
                      uint64_t ninv_override)
{
  if(!out) return fail(NTT_ERR_ARG, "null plan pointer");
  *out = nullptr;
  if(!is_pow2(N) || N < 2 || N > (1ull << 28)) return fail(NTT_ERR_ARG, "N must be a power of two in [2,2^28]");
  if(q < 3 || !(q & 1) || q >= (1ull << 61)) return fail(NTT_ERR_ARG, "q must be odd, 3 <= q < 2^61");
  if((q - 1) % (2 * N) != 0) return fail(NTT_ERR_ARG, "2N must divide q-1");
  int ar = 0;
  int rc = resolve_arith(arith, q, (int)h_log2(N), &ar);
  if(rc) return rc;
  rc = check_device(device);
  if(rc) return rc;
  USE_DEVICE(device);
  ntt_plan *p = new ntt_plan();
  p->device   = device;
  p->N        = N;
  p->q        = q;
  p->root     = root;
  p->m        = (int)h_log2(N);
  p->arith    = ar;
  /* NTT_ARITH_AUTO on a modulus the FP64 policies cannot serve: the throughput form of the integer arithmetic; an
   * explicit NTT_ARITH_U64 keeps the reference's butterflies and lazy words (NTT_OPT_INT_WIDE switches either way) */
  p->int_cls  = (arith == NTT_ARITH_AUTO && ar == NTT_ARITH_U64) ? int_wide_class(q) : -1;
  {
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
      p->num_cus = prop.multiProcessorCount;
    }
  }
  const bool r4  = ar == NTT_ARITH_U64_R4;
  const bool gen = ts.gen_root != 0 && ts.gen_root_inv != 0;
  p->has_fwd     = gen || (r4 ? !(ts.efwd.empty() && ts.fwd.empty()) : !ts.fwd.empty());
  p->has_inv     = gen || (r4 ? !(ts.einv.empty() && ts.inv.empty()) : !ts.inv.empty());
  /* inverse power table slot 1 (w^-N/2) is all the constants need */
  std::vector<uint64_t> inv_for_consts(2, 1);
  if(gen) inv_for_consts[1] = h_powmod(ts.gen_root_inv, N / 2, q);
  else if(!ts.inv.empty()) inv_for_consts.assign(ts.inv.begin(), ts.inv.begin() + 2);
  else if(ts.einv.size() >= 4) inv_for_consts[1] = ts.einv[2]; /* e[2k] = w[k] */
  p->cu = h_consts_u64(q, N, inv_for_consts);
  if(ninv_override) {
    p->cu.ninv  = h_tw_u64(ninv_override % q, q);
    p->cu.wninv = h_tw_u64(h_mulmod(ninv_override % q, inv_for_consts[1], q), q);
  }
  if(ar == NTT_ARITH_F64) {
    p->cf = h_consts_f64(q, N, inv_for_consts);
    if(ninv_override) {
      p->cf.ninv  = h_tw_f64(ninv_override % q, q);
      p->cf.wninv = h_tw_f64(h_mulmod(ninv_override % q, inv_for_consts[1], q), q);
    }
    const int k = h_f64_ksh(q);
    p->kcls     = !h_f64_eligible(q) ? kWideClass : (k >= 18 ? 18 : (k >= 1 ? 1 : 0));
  }
  rc = NTT_OK;
  if(gen) {
    const uint64_t ninv = ninv_override ? ninv_override % q : h_powmod(N % q, q - 2, q);
    rc                  = device_build_direction(p, ts.gen_root, false, ninv);
    if(!rc) rc = device_build_direction(p, ts.gen_root_inv, true, ninv);
  } else if(r4) {
    /* expanded tables: the caller's, or derived from the power tables */
    if(p->has_fwd) {
      const std::vector<uint64_t> e = ts.efwd.empty() ? h_expand_radix4(ts.fwd, q) : ts.efwd;
      rc                            = upload_u64(&p->d_fwd, e, ts.efwd_con, q);
    }
    if(!rc && p->has_inv) {
      const std::vector<uint64_t> e = ts.einv.empty() ? h_expand_radix4(ts.inv, q) : ts.einv;
      rc                            = upload_u64(&p->d_inv, e, ts.einv_con, q);
    }
  } else {
    /* the full inverse table carries 16 extra records behind its N slots: N^-1 * winv[k], k < 16,
     * the twiddles of the last inverse group with the scaling folded in (run_group0_folded) */
    const std::vector<uint64_t> inv_ext =
      p->has_inv ? h_with_folded_ninv(ts.inv, ninv_override ? ninv_override % q : h_powmod(N % q, q - 2, q), q) : ts.inv;
    if(ar == NTT_ARITH_F64) {
      if(p->has_fwd) rc = upload_table<TwF64>(&p->d_fwd, ts.fwd, q, h_tw_f64);
      if(!rc && p->has_fwd) rc = upload_table<double>(&p->d_fwd8, ts.fwd, q, [](uint64_t w, uint64_t qq) { return h_tw_f64(w, qq).w; });
      if(!rc && p->has_inv) rc = upload_table<TwF64>(&p->d_inv, inv_ext, q, h_tw_f64);
      if(!rc && p->has_inv) rc = upload_table<double>(&p->d_inv8, ts.inv, q, [](uint64_t w, uint64_t qq) { return h_tw_f64(w, qq).w; });
    } else {
      if(p->has_fwd) rc = upload_u64(&p->d_fwd, ts.fwd, ts.fwd_con, q);
      if(!rc && p->has_inv) rc = upload_u64(&p->d_inv, inv_ext, ts.inv_con, q); /* caller precons cover the first N records */
    }
  }
  if(!rc) p->limbrec = limbrec_bytes(p);
  if(!rc && p->arith == NTT_ARITH_U64_R4) {
    LimbRec<ArithU64> r{};
    memcpy(&r, p->limbrec.data(), sizeof r);
    r.c.ninv = h_tw_u64(1, p->q);
    p->limbrec_mid.resize(sizeof r);
    memcpy(p->limbrec_mid.data(), &r, sizeof r);
  }
  if(rc) {
    ntt_plan_destroy(p);
    return rc;
  }
  *out = p;
  return NTT_OK;
}

extern "C" int ntt_plan_create(ntt_plan **out, int device, uint64_t N, uint64_t q, uint64_t root, int arith)
{
  if(!is_pow2(N) || q < 3 || root == 0 || root >= q) return fail(NTT_ERR_ARG, "bad N, q or root");
  if(h_powmod(root, N, q) != q - 1) return fail(NTT_ERR_ARG, "root is not a primitive 2N-th root of unity mod q");
  /* root^-1 and N^-1 come from Fermat's little theorem, which holds only for a prime q (a composite q can pass
   * 2N | q-1 and root^N == -1, and Carmichael-type composites even pass a Fermat test): deterministic Miller-Rabin
   * for 64-bit integers, then the two inverses are checked for what they are */
  if(!h_is_prime(q)) return fail(NTT_ERR_ARG, "q is not prime");
  const uint64_t rinv = h_powmod(root, q - 2, q);
  if(h_mulmod(root, rinv, q) != 1 || h_mulmod(N % q, h_powmod(N % q, q - 2, q), q) != 1) {
    return fail(NTT_ERR_ARG, "q is not prime (root^(q-2) is not the inverse of root)");
  }
  TableSet ts;
  ts.gen_root     = root;
  ts.gen_root_inv = rinv;
  return plan_build(out, device, N, q, root, ts, arith, 0);
}

extern "C" int ntt_plan_create_from_tables(ntt_plan **out, int device, uint64_t N, uint64_t q,
                                           const uint64_t *w_powers, const uint64_t *w_inv_powers, int arith)
{
  if(!w_powers && !w_inv_powers) return fail(NTT_ERR_ARG, "no table given");
  if(!is_pow2(N)) return fail(NTT_ERR_ARG, "N must be a power of two");
  if(N < 2 || q < 3) return fail(NTT_ERR_ARG, "bad N or q");
  /* cheap consistency checks of the caller's tables (reference layout, pre_compute.h:38-66): slot 0 is
   * root^0, slot 1 is root^(N/2) -- a square root of -1 -- and the two tables are inverses of each other */
  for(const uint64_t *t : {w_powers, w_inv_powers}) {
    if(!t) continue;
    if(t[0] != 1 || t[1] >= q || h_mulmod(t[1], t[1], q) != q - 1) {
      return fail(NTT_ERR_ARG, "table is not a bit-reversed power table of a primitive 2N-th root of unity mod q");
    }
  }
  if(w_powers && w_inv_powers) {
    const uint64_t last = N - 1; /* root^(bitrev(N-1)) = root^(N-1) */
    if(h_mulmod(w_powers[1], w_inv_powers[1], q) != 1 || h_mulmod(w_powers[last], w_inv_powers[last], q) != 1) {
      return fail(NTT_ERR_ARG, "w_inv_powers is not the inverse of w_powers");
    }
  }
  TableSet ts;
  if(w_powers) ts.fwd.assign(w_powers, w_powers + N);
  if(w_inv_powers) ts.inv.assign(w_inv_powers, w_inv_powers + N);
  return plan_build(out, device, N, q, 0, ts, arith, 0);
}

extern "C" void ntt_plan_destroy(ntt_plan *p)
{
  if(!p) return;
  DeviceGuard guard_(p->device);
  if(p->d_fwd) (void)hipFree(p->d_fwd);
  if(p->d_inv) (void)hipFree(p->d_inv);
  if(p->d_fwd8) (void)hipFree(p->d_fwd8);
  if(p->d_inv8) (void)hipFree(p->d_inv8);
  for(const ntt_plan::TeamBuf &tb : p->team_bufs) {
    if(tb.d) (void)hipFree(tb.d);
    if(tb.g) (void)hipFree(tb.g);
  }
  for(void *d : p->team_retired) (void)hipFree(d);
  for(const ntt_plan::PtrBuf &pb : p->ptr_bufs) {
    if(pb.d) (void)hipFree(pb.d);
    if(pb.h) (void)hipHostFree(pb.h);
    if(pb.copied) (void)hipEventDestroy(pb.copied);
  }
  for(const std::pair<void *, void *> &r : p->ptr_retired) {
    (void)hipFree(r.first);
    (void)hipHostFree(r.second);
  }
  if(p->own_stream) (void)hipStreamDestroy(p->own_stream);
  delete p;
}

extern "C" int ntt_plan_info(const ntt_plan *p, uint64_t info[8])
{
  if(!p || !info) return fail(NTT_ERR_ARG, "null argument");
  info[0] = p->N;
  info[1] = p->q;
  info[2] = (uint64_t)p->m;
  info[3] = (uint64_t)p->arith;
  info[4] = p->kcls == kWideClass ? 52u : (uint64_t)p->kcls; /* 52: the reduce-both-operands policy for q up to 2^52 */
  if(p->arith == NTT_ARITH_U64) info[4] = p->int_cls < 0 ? 0u : 100u + (uint64_t)p->int_cls; /* 100 + K: ArithU64X<K> */
  /* launches (= passes over the data) of one forward transform of a large batch: 1 where one launch carries both passes --
   * the XCD-local kernel (FP64 policies, N = 2^15..2^17, unless switched off: its automatic choice takes forward
   * transforms of 512 polynomials or more) or the two-phase kernel where it is forced -- else the pass list's length */
  {
    const bool f64big = !p->generic && p->arith == NTT_ARITH_F64;
    const bool team   = (f64big || (!p->generic && p->arith == NTT_ARITH_U64 && p->int_cls >= 0)) && p->m >= kTeamBlock + 3 &&
                      p->m <= kTeamBlock + 5 && p->xcd_local != 0;
    const bool tp     = f64big && p->m >= kFusedMax + 2 && p->m <= kFusedMax + 3 &&
                    (p->two_phase == 1 || (p->two_phase < 0 && p->m == kFusedMax + 2 && p->kcls != kWideClass));
    const bool op     = f64big && p->m == kFusedMax + 1 && p->one_pass != 0; /* 2^15: one pass, the polynomial in registers */
    info[5] = (team || tp || op) ? 1u : (uint64_t)make_passes(p->m, p->generic).n;
  }
  info[6] = (uint64_t)p->device;
  info[7] = p->root;
  return NTT_OK;
}

extern "C" int ntt_plan_export_table(const ntt_plan *p, int which, void *h_dst, size_t bytes)
{
  if(!p || !h_dst) return fail(NTT_ERR_ARG, "null argument");
  const void *src = which == 0 ? p->d_fwd : which == 1 ? p->d_inv : which == 2 ? p->d_fwd8 : which == 3 ? p->d_inv8 : nullptr;
  if(!src) return fail(NTT_ERR_ARG, "the plan has no such table");
  const size_t rec  = p->arith == NTT_ARITH_F64 ? sizeof(TwF64) : sizeof(TwU64);
  const size_t have = which >= 2 ? p->N * sizeof(double)
                                 : (p->arith == NTT_ARITH_U64_R4 ? 2 * p->N : p->N + (which == 1 ? 16 : 0)) * rec;
  if(bytes > have) return fail(NTT_ERR_ARG, "table is smaller than the request");
  USE_DEVICE(p->device);
  HIP_TRY(hipMemcpy(h_dst, src, bytes, hipMemcpyDeviceToHost));
  return NTT_OK;
}

extern "C" int ntt_plan_set_generic(ntt_plan *p, int on)
{
  if(!p) return fail(NTT_ERR_ARG, "null plan");
  if(on && p->arith == NTT_ARITH_U64_R4) return fail(NTT_ERR_UNSUPPORTED, "the radix-4 policy has no column-pass form");
  p->generic = on != 0;
  return NTT_OK;
}

extern "C" int ntt_plan_set_option(ntt_plan *p, int option, int64_t value)
{
  if(!p) return fail(NTT_ERR_ARG, "null plan");
  switch(option) {
    case NTT_OPT_MAX_GRID:
      if(value < 0) return fail(NTT_ERR_ARG, "max grid must be >= 0");
      p->max_grid = (int)value;
      return NTT_OK;
    case NTT_OPT_RNS_LAUNCH:
      p->rns_launch = value < 0 ? -1 : (value != 0);
      return NTT_OK;
    case NTT_OPT_DOT_FUSED:
      p->dot_fused = value != 0;
      return NTT_OK;
    case NTT_OPT_MAX_BATCH_HINT:
      if(value < 0) return fail(NTT_ERR_ARG, "batch hint must be >= 0");
      {
        std::lock_guard<std::mutex> lock(p->team_mu); /* team_buffer reads it under this mutex */
        p->batch_hint = (uint64_t)value;
      }
      return value ? ntt_plan_reserve(p, nullptr, (uint64_t)value) : NTT_OK;
    case NTT_OPT_BLOCK_OVERSUB:
      if(value < 0 || value > 256) return fail(NTT_ERR_ARG, "workgroups per resident slot: 0 (default) .. 256");
      p->block_oversub = (int)value;
      return NTT_OK;
    case NTT_OPT_CHUNK_MIB:
      if(value < 1) return fail(NTT_ERR_ARG, "chunk must be >= 1 MiB");
      p->chunk_mib = (int)value;
      return NTT_OK;
    case NTT_OPT_TWO_PHASE:
      p->two_phase = value < 0 ? -1 : (value != 0);
      return NTT_OK;
    case NTT_OPT_ONE_PASS:
      p->one_pass = value < 0 ? -1 : (value != 0);
      return NTT_OK;
    case NTT_OPT_FUSED_PRODUCT:
      if(value < 0 || value > 2) return fail(NTT_ERR_ARG, "fused product: 0, 1 or 2");
      p->fused_product = (int)value;
      return NTT_OK;
    case NTT_OPT_XCD_LOCAL:
      p->xcd_local = value < 0 ? -1 : (value != 0);
      return NTT_OK;
    case NTT_OPT_XCD_LOCAL_LAG:
      if(value < 0 || value > 64) return fail(NTT_ERR_ARG, "lag must be 0 (default) .. 64 polynomials");
      p->team_lag = (int)value;
      return NTT_OK;
    case NTT_OPT_XCD_LOCAL_WGS_PER_CU:
      if(value < 0 || value > 4) return fail(NTT_ERR_ARG, "workgroups per CU: 0 (default) .. 4");
      p->team_wpc = (int)value;
      return NTT_OK;
    case NTT_OPT_BLOCK_LOG:
      if(value != 0 && value != kFusedSmallBlock && value != kFusedLarge) return fail(NTT_ERR_ARG, "block size must be 0 (automatic), 12 or 14");
      if(value == kFusedSmallBlock && p->m > kFusedSmallBlock + 4) return fail(NTT_ERR_ARG, "2^12-point blocks need at most 4 leading stages");
      p->block_log = (int)value;
      return NTT_OK;
    case NTT_OPT_F64_CLASS: {
      /* a coarser (smaller) headroom class than the modulus allows is always valid: it only reduces more often */
      if(p->arith != NTT_ARITH_F64 || p->kcls == kWideClass) return fail(NTT_ERR_ARG, "not a plan of the scheduled FP64 policy");
      const int k = h_f64_ksh(p->q);
      if(value == 0 || (value == 1 && k >= 1) || (value == 18 && k >= 18)) {
        p->kcls = (int)value;
        return NTT_OK;
      }
      return fail(NTT_ERR_ARG, "headroom class not instantiated or not valid for this modulus");
    }
    case NTT_OPT_INT_WIDE: {
      /* 0: the reference's butterflies; 1: the widest headroom class q permits; 10 + K: class K (tests: a narrower class
       * than q permits is always valid, it only folds more often) */
      if(p->arith != NTT_ARITH_U64) return fail(NTT_ERR_ARG, "not a plan of the integer policy");
      const int best = int_wide_class(p->q);
      if(value == 0) {
        p->int_cls = -1;
        return NTT_OK;
      }
      if(best < 0) return fail(NTT_ERR_UNSUPPORTED, "the wide integer policy serves 2^40 <= q < 2^61");
      if(value == 1) {
        p->int_cls = best;
        return NTT_OK;
      }
      if((value == 10 || value == 11 || value == 13) && value - 10 <= best) {
        p->int_cls = (int)value - 10;
        return NTT_OK;
      }
      return fail(NTT_ERR_ARG, "integer headroom class not instantiated or not valid for this modulus");
    }
    default: return fail(NTT_ERR_ARG, "unknown option");
  }
}

extern "C" int ntt_plan_get_option(const ntt_plan *p, int option, int64_t *value)
{
  if(!p || !value) return fail(NTT_ERR_ARG, "null argument");
  switch(option) {
    case NTT_OPT_MAX_GRID: *value = p->max_grid; return NTT_OK;
    case NTT_OPT_CHUNK_MIB: *value = p->chunk_mib; return NTT_OK;
    case NTT_OPT_F64_CLASS: *value = p->arith == NTT_ARITH_F64 ? p->kcls : -1; return NTT_OK;
    case NTT_OPT_TWO_PHASE: *value = p->two_phase; return NTT_OK;
    case NTT_OPT_ONE_PASS: *value = p->one_pass; return NTT_OK;
    case NTT_OPT_FUSED_PRODUCT: *value = p->fused_product; return NTT_OK;
    case NTT_OPT_BLOCK_LOG: *value = p->block_log; return NTT_OK;
    case NTT_OPT_XCD_LOCAL: *value = p->xcd_local; return NTT_OK;
    case NTT_OPT_XCD_LOCAL_LAG: *value = p->team_lag; return NTT_OK;
    case NTT_OPT_XCD_LOCAL_WGS_PER_CU: *value = p->team_wpc; return NTT_OK;
    case NTT_OPT_INT_WIDE: *value = p->arith == NTT_ARITH_U64 ? (p->int_cls < 0 ? 0 : 10 + p->int_cls) : -1; return NTT_OK;
    case NTT_OPT_BLOCK_OVERSUB: *value = p->block_oversub; return NTT_OK;
    case NTT_OPT_RNS_LAUNCH: *value = p->rns_launch; return NTT_OK;
    case NTT_OPT_DOT_FUSED: *value = p->dot_fused; return NTT_OK;
    case NTT_OPT_MAX_BATCH_HINT: *value = (int64_t)p->batch_hint; return NTT_OK;
    case NTT_OPT_CTL_ALLOCATIONS: {
      std::lock_guard<std::mutex> lock(p->team_mu);
      *value = (int64_t)p->ctl_allocs;
      return NTT_OK;
    }
    default: return fail(NTT_ERR_ARG, "unknown option");
  }
}

/* ------------------------------------------------------------------ */
/* transforms                                                          */
/* ------------------------------------------------------------------ */
/* A launch over several limbs runs ONE kernel instantiation: the coarsest headroom class among the limbs of the run (a coarser
 * class is valid for every modulus a finer one serves: it only reduces / folds more often).  rns_for_runs sets the override
 * around the run's call; every class dispatch below goes through these two. */
static thread_local int t_run_kcls = INT_MIN, t_run_int_cls = INT_MIN;
static int eff_kcls(const ntt_plan *p) { return (t_run_kcls != INT_MIN && p->kcls != kWideClass) ? t_run_kcls : p->kcls; }
static int eff_int_cls(const ntt_plan *p) { return (t_run_int_cls != INT_MIN && p->int_cls >= 0) ? t_run_int_cls : p->int_cls; }

static hipError_t dispatch_pass(const ntt_plan *p, const PassArgs &pa)
{
  if(p->arith == NTT_ARITH_U64) {
    switch(p->generic ? -1 : eff_int_cls(p)) { /* (generic = the column-pass cross-check path: the reference's butterflies) */
      case 3: return launch_pass<ArithU64X<3>, 3>(pa);
      case 1: return launch_pass<ArithU64X<1>, 1>(pa);
      case 0: return launch_pass<ArithU64X<0>, 0>(pa);
      default: return launch_pass<ArithU64, 0>(pa);
    }
  }
  if(p->arith == NTT_ARITH_U64_R4) return launch_pass<ArithU64R4, 0>(pa);
  switch(eff_kcls(p)) {
    case kWideClass: return launch_pass<ArithF64W, 0>(pa);
    case 18: return launch_pass<ArithF64, 18>(pa);
    case 1: return launch_pass<ArithF64, 1>(pa);
    default: return launch_pass<ArithF64, 0>(pa);
  }
}

/* XCD-local two-pass launches: FP64 policies and the wide integer policy, N = 2^15..2^17, plain calls (canonical in,
 * canonical out).  Automatic choice (-1), as measured (profiles/r03/sweep_xcd_local.txt, profiles/r04/ab_xcd_*.txt): the
 * FORWARD transform of a batch that keeps all eight queues busy for several lags (+8..14 % over one launch per pass; wide
 * integer policy +18..23 %); the inverse only at 2^17 (+3..5 %, integer +15 %) -- at 2^15 and 2^16 its first pass is the
 * heavy one and the column items wait longer than the L2 can hold their polynomials: per-pass launches there.
 * NTT_OPT_XCD_LOCAL 1 forces the path wherever it is built (batch >= 64), 0 disables it. */
/* nlimbs > 1: the limbs of an RNS set in ONE launch (the queues run over all limbs' polynomials): `batch` is per limb */
static bool team_applies(const ntt_plan *p, uint64_t batch, bool inverse, bool wide, bool lazy, int nlimbs, bool product = false)
{
  if(nlimbs < 1 || nlimbs > kMaxLimbs) return false;
  batch *= (uint64_t)nlimbs; /* polynomials of the launch */
  /* the FP64 policies; transforms (not the product launch) also for the wide integer policy */
  const bool int_wide = p->arith == NTT_ARITH_U64 && p->int_cls >= 0 && !product;
  if((p->arith != NTT_ARITH_F64 && !int_wide) || p->generic || p->m < kTeamBlock + 3 || p->m > kTeamBlock + 5 || wide || lazy ||
     batch < 64 || batch >= (1ull << 29)) {
    return false;
  }
  const int on = p->xcd_local;
  if(on >= 0) return on == 1;
  /* the wide integer policy (profiles/r04/ab_xcd_int.txt): forward +18..23 % at all three sizes; inverse +15 % at 2^17, none at
   * 2^16, -14 % at 2^15 */
  if(int_wide) return batch >= 512 && (!inverse || p->m == kTeamBlock + 5);
  /* FP64 inverse (profiles/r04/ab_xcd_f64_inv.txt): slower at 2^15 and 2^16 (-21 %, -12 %), +2.6..5.5 % at 2^17 */
  if(inverse) return p->m == kTeamBlock + 5 && batch >= 512;
  /* the product launch (all transforms of a product as items of one launch) pays from 2^23 coefficients per operand on:
   * 64 / 128 / 256 polynomials at 2^17 / 2^16 / 2^15 (measured against the per-chunk launches, single limb, batches
   * 64..384: profiles/r03/ablations.txt (h)) */
  if(product) return batch >= 64 && (batch << p->m) >= (1ull << 23);
  return !inverse && batch >= 512;
}

/* The stream's control block, at least sizeof(TeamCtl) + batch counters.
 * A (plan, stream) pair owns TWO blocks, both allocated by the first direct (uncaptured) call: one for direct launches
 * and one for launches captured into HIP graphs.  A captured launch bakes the block's address into its clearing-kernel and kernel
 * nodes, so the graph block is never freed or regrown while the plan lives, and direct launches never touch it: a graph
 * replayed on another stream cannot collide with direct calls on the capture stream.  (Two graphs captured on the same
 * plan and stream share the graph block: replay them one after the other, not concurrently -- INTEGRATION.md.)
 * Nothing here synchronises the device: an outgrown direct block is retired, not freed (launches still queued may be
 * using it), and released with the plan.  While capturing nothing is allocated (not capturable): without a graph block of
 * sufficient size *out stays null and the caller takes the per-pass launches. */
static int team_buffer(const ntt_plan *p, void *stream, uint64_t batch, void **out)
{
  std::lock_guard<std::mutex> lock(p->team_mu);
  /* (batch_hint is written by ntt_plan_set_option under the same mutex) */
  const size_t need = sizeof(TeamCtl) + (size_t)(batch > 2 * p->batch_hint ? batch : 2 * p->batch_hint) * sizeof(unsigned);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  /* a query that fails (the legacy stream asked during a global-mode capture) counts as capturing: allocating would invalidate it */
  const bool capturing = hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone;
  *out = nullptr;
  ntt_plan::TeamBuf *tb = nullptr;
  for(ntt_plan::TeamBuf &t : p->team_bufs) {
    if(t.stream == stream) tb = &t;
  }
  if(capturing) {
    if(tb && tb->g && tb->gbytes >= need) *out = tb->g;
    return NTT_OK;
  }
  if(!tb) {
    p->team_bufs.push_back(ntt_plan::TeamBuf{stream, nullptr, 0, nullptr, 0});
    tb = &p->team_bufs.back();
  }
  if(tb->bytes < need) {
    /* a reserved size is taken as given, an outgrown one doubled */
    const size_t grow = p->batch_hint && !tb->d ? need : need * 2;
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, grow));
    p->ctl_allocs++;
    if(tb->d) p->team_retired.push_back(tb->d);
    tb->d     = d;
    tb->bytes = grow;
  }
  if(tb->gbytes < tb->bytes) {
    /* the graph block follows the direct block's size (first direct call, a larger direct call, ntt_plan_reserve): graphs
     * captured so far keep the address baked into their nodes -- the old block is retired, not freed -- and captures from now on
     * get the larger one.  (Rounds 3-5 sized it once: a larger ntt_plan_reserve before a capture silently left the captured call on
     * the per-pass launches.) */
    void *g = nullptr;
    HIP_TRY(hipMalloc(&g, tb->bytes));
    p->ctl_allocs++;
    if(tb->g) p->team_retired.push_back(tb->g);
    tb->g      = g;
    tb->gbytes = tb->bytes;
  }
  *out = tb->d;
  return NTT_OK;
}

/* Sizes the (plan, stream) control blocks for calls of up to `polys` polynomials x limbs (products count their operands:
 * twice that many entries) -- the allocation an XCD-local launch would otherwise make on its first call, an implicit device
 * synchronisation inside a call documented as asynchronous.  Not while the stream is being captured. */
extern "C" int ntt_plan_reserve(const ntt_plan *p, void *stream, uint64_t polys)
{
  if(!p) return fail(NTT_ERR_ARG, "null plan");
  if(polys == 0) return NTT_OK;
  USE_DEVICE(p->device);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if(hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
    return fail(NTT_ERR_ARG, "reserve before the capture begins");
  void *ctl = nullptr;
  return team_buffer(p, stream, 2 * polys, &ctl);
}

/* The limbs one launch serves: the plan's own record (every ordinary call), or an RNS set's records (host array, copied
 * into the kernel arguments) with the word distance between the limbs' slabs ([limb][batch][N]). */
struct LimbSet {
  const void *d;
  int         n;
  uint64_t    stride;  /* words between consecutive limbs of one polynomial */
  uint64_t    pstride; /* words between consecutive polynomials of one limb; 0 = dense (N) */
  const uint64_t *ptab = nullptr; /* pointer batch: DEVICE table, entry i = address of polynomial i's limb 0 (the data pointer of the call is
                                   * then the limb offset alone: null + limb * stride words); pstride is then only a hint of the typical
                                   * spacing (queue numbering of the XCD-local launches) */
};
/* (pointer + words) that is also defined for the null base of a pointer batch */
static uint64_t *advance(uint64_t *base, uint64_t words) { return reinterpret_cast<uint64_t *>(reinterpret_cast<uintptr_t>(base) + 8u * (uintptr_t)words); }
/* the words between consecutive polynomials of a limb */
static uint64_t poly_words(const ntt_plan *p, const LimbSet &ls) { return ls.pstride ? ls.pstride : p->N; }

static int run_transform(const ntt_plan *p, uint64_t *d_a, uint64_t batch, bool inverse, bool wide, void *stream,
                         bool lazy = false, const LimbSet *set = nullptr)
{
  if(!p || (!d_a && batch && !(set && set->ptab))) return fail(NTT_ERR_ARG, "null argument");
  if(batch == 0) return NTT_OK;
  if(inverse ? !p->has_inv : !p->has_fwd) return fail(NTT_ERR_ARG, "plan lacks the table for this direction");
  USE_DEVICE(p->device);
  const LimbSet ls = set ? *set : LimbSet{p->limbrec.data(), 1, 0, 0};
  const PassList L = p->arith == NTT_ARITH_U64_R4
                         ? make_passes_r4(p->m)
                         : make_passes(p->m, p->generic, p->block_log ? p->block_log : multi_pass_block(p->m, inverse, p->arith == NTT_ARITH_F64));
  /* N = 2^15, FP64 policies: ONE pass -- a workgroup holds the whole polynomial in its registers, every coefficient crosses HBM
   * twice (ntt_kernels.h: onepass_kernel; measured against the two-pass forms in profiles/r06/onepass_2p15.txt).  One workgroup
   * per CU and polynomial: the automatic choice wants a polynomial for every CU; smaller batches spread over the chip as 2^12-point
   * blocks through the per-pass launches.  Lazy outputs keep the two-pass forms (the FP64 lazy store is a kernel variant of the
   * block kernels). */
  if(p->m == kFusedMax + 1 && p->arith == NTT_ARITH_F64 && !p->generic && !lazy && ls.n <= kMaxLimbs &&
     (p->one_pass == 1 || (p->one_pass < 0 && batch * (uint64_t)ls.n >= (uint64_t)p->num_cus))) {
    PassArgs pa{};
    pa.a           = d_a;
    pa.limbs       = ls.d;
    pa.nlimbs      = ls.n;
    pa.limb_stride = ls.stride;
    pa.poly_stride = ls.pstride;
    pa.ptab        = ls.ptab;
    pa.batch       = batch;
    pa.logn        = (uint32_t)p->m;
    pa.fused       = 4;
    pa.r           = 1;
    pa.s           = 0;
    pa.inverse     = inverse;
    pa.wide        = wide;
    pa.lastinv     = inverse;
    pa.ends        = 1;
    pa.max_grid    = p->max_grid;
    pa.num_cus     = p->num_cus;
    pa.stream      = (hipStream_t)stream;
    hipError_t e   = dispatch_pass(p, pa);
    if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return NTT_OK;
  }
  /* both passes as items of ONE launch with the intermediate kept in each XCD's L2 (ntt_kernels.h: team_kernel) */
  void *ctl = nullptr;
  if(team_applies(p, batch, inverse, wide, lazy, ls.n)) {
    /* the queue heads and counters live in a buffer the plan keeps per stream; it is allocated on first use -- except
     * while the stream is being captured into a HIP graph (allocation is not capturable): the call then takes the
     * per-pass launches, which need no memory of their own */
    int rc = team_buffer(p, stream, batch * (uint64_t)ls.n, &ctl);
    if(rc) return rc;
  }
  if(ctl) {
    PassArgs pa{};
    pa.a           = d_a;
    pa.limbs       = ls.d;
    pa.nlimbs      = ls.n;
    pa.limb_stride = ls.stride;
    pa.poly_stride = ls.pstride;
    pa.ptab        = ls.ptab;
    pa.batch       = batch;
    pa.logn        = (uint32_t)p->m;
    pa.fused       = 3;
    pa.r           = p->m - kTeamBlock;
    pa.s           = 0;
    pa.inverse     = inverse;
    pa.lastinv     = inverse;
    pa.ends        = 1;
    pa.max_grid    = p->max_grid;
    pa.num_cus     = p->num_cus;
    pa.oversub     = p->block_oversub;
    pa.team_ctl    = ctl;
    /* polynomials between the two passes of a queue.  The L2 keeps the intermediate while lag x polynomial size stays
     * below about 2.5 MiB (measured, FETCH_SIZE 1.0x the data: 2^15 up to lag 10, 2^16 up to 5, 2^17 not even at 2;
     * profiles/r03/team_kernel_l2_retention.txt), but with four workgroups per CU a lag below 8 makes second-pass items
     * wait: 2^15 gets both (lag 10: 0.43 of the roofline against 0.40-0.41 beyond), 2^16 and 2^17 run fastest at 8-10
     * with the second pass served by the Infinity Cache (profiles/r03/sweep_xcd_local_lag.txt) */
    pa.team_lag    = p->team_lag ? p->team_lag : (p->m == kTeamBlock + 4 ? 8 : 10);
    pa.team_wpc    = p->team_wpc;
    pa.stream      = (hipStream_t)stream;
    hipError_t e   = dispatch_pass(p, pa);
    if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return NTT_OK;
  }
  /* the one-launch two-phase kernel owns a CU per polynomial: it only pays when the batch fills the chip (measured +3 % at
   * batch ~30k; a batch below the CU count would leave CUs idle where the per-pass launches spread one polynomial
   * over 16 workgroups) -- the automatic choice is gated on that, an explicit NTT_OPT_TWO_PHASE 1 is honoured as given */
  const bool tp_auto = p->two_phase < 0 && !inverse && p->m == kFusedMax + 2 && p->kcls != kWideClass &&
                       batch >= 2ull * (uint64_t)p->num_cus;
  if((p->two_phase == 1 || tp_auto) && ls.n == 1 && !p->generic && p->arith == NTT_ARITH_F64 && p->m >= kFusedMax + 2 &&
     p->m <= kFusedMax + 3) {
    /* one launch, one workgroup per polynomial, both passes back to back (ntt_kernels.h: twophase_kernel) */
    PassArgs pa{};
    pa.a        = d_a;
    pa.limbs    = ls.d;
    pa.nlimbs   = ls.n;
    pa.limb_stride = ls.stride;
    pa.poly_stride = ls.pstride;
    pa.ptab     = ls.ptab;
    pa.batch    = batch;
    pa.logn     = (uint32_t)p->m;
    pa.fused    = 2;
    pa.r        = p->m - kFusedMax;
    pa.s        = 0;
    pa.inverse  = inverse;
    pa.wide     = wide;
    pa.lastinv  = inverse;
    pa.lazy     = lazy;
    pa.ends     = 1;
    pa.max_grid = p->max_grid;
    pa.num_cus  = p->num_cus;
    pa.oversub  = p->block_oversub;
    pa.stream   = (hipStream_t)stream;
    hipError_t e = dispatch_pass(p, pa);
    if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    return NTT_OK;
  }
  /* Multi-pass transforms (N > 2^14) are run chunk by chunk so that what one pass
   * writes is still in the 256 MiB Infinity Cache when the next pass reads it:
   * only the first read and the last write of a chunk have to reach HBM. */
  uint64_t chunk = batch;
  if(L.n > 1) {
    const uint64_t budget = (uint64_t)p->chunk_mib << 20;
    chunk                 = budget / (p->N * sizeof(uint64_t) * (uint64_t)ls.n); /* a chunk holds this many polynomials of EVERY limb */
    if(chunk < 1) chunk = 1;
    if(chunk > batch) chunk = batch;
  }
  for(uint64_t first = 0; first < batch; first += chunk) {
    const uint64_t nb = batch - first < chunk ? batch - first : chunk;
    for(int k = 0; k < L.n; k++) {
      const Pass &ps = L.p[inverse ? L.n - 1 - k : k];
      PassArgs    pa{};
      /* a chunk starts `first` polynomials in: that many strides -- or table entries -- further */
      pa.a        = ls.ptab ? d_a : d_a + first * poly_words(p, ls);
      pa.ptab     = ls.ptab ? ls.ptab + first : nullptr;
      /* (radix-4 inverse: a pass that does not end the transform multiplies by 1, not by N^-1) */
      pa.limbs    = p->arith == NTT_ARITH_U64_R4 && inverse && ps.s != 0 ? (const void *)p->limbrec_mid.data() : ls.d;
      pa.nlimbs   = ls.n;
      pa.limb_stride = ls.stride;
      pa.poly_stride = ls.pstride;
      pa.lazy     = lazy;
      pa.ends     = k == L.n - 1;
      pa.batch    = nb;
      pa.logn     = (uint32_t)p->m;
      pa.fused    = ps.fused;
      pa.r        = ps.r;
      pa.s        = ps.s;
      pa.inverse  = inverse;
      pa.wide     = wide && k == 0;
      pa.lastinv  = inverse && ps.s == 0;
      pa.max_grid = p->max_grid;
      pa.num_cus  = p->num_cus;
      pa.oversub  = p->block_oversub;
      pa.stream   = (hipStream_t)stream;
      hipError_t e = dispatch_pass(p, pa);
      if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    }
  }
  return NTT_OK;
}

extern "C" int ntt_fwd_batch(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream)
{
  return run_transform(p, d_a, batch, false, false, stream);
}
extern "C" int ntt_inv_batch(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream)
{
  return run_transform(p, d_a, batch, true, false, stream);
}
extern "C" int ntt_transform_batch(const ntt_plan *p, uint64_t *d_a, uint64_t batch, unsigned flags, void *stream)
{
  if(flags & ~(unsigned)(NTT_FLAG_INVERSE | NTT_FLAG_WIDE_IN | NTT_FLAG_LAZY_OUT)) return fail(NTT_ERR_ARG, "unknown flag");
  return run_transform(p, d_a, batch, (flags & NTT_FLAG_INVERSE) != 0, (flags & NTT_FLAG_WIDE_IN) != 0, stream,
                       (flags & NTT_FLAG_LAZY_OUT) != 0);
}
extern "C" int ntt_fwd_batch_lazy(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream)
{
  return run_transform(p, d_a, batch, false, false, stream, true);
}
extern "C" int ntt_inv_batch_lazy(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream)
{
  return run_transform(p, d_a, batch, true, false, stream, true);
}
extern "C" int ntt_fwd_batch_wide(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream)
{
  return run_transform(p, d_a, batch, false, true, stream);
}
extern "C" int ntt_inv_batch_wide(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream)
{
  return run_transform(p, d_a, batch, true, true, stream);
}

/* ------------------------------------------------------------------ */
/* utility kernels                                                     */
/* ------------------------------------------------------------------ */
__device__ __forceinline__ uint64_t splitmix64_dev(uint64_t x)
{
  x += 0x9e3779b97f4a7c15ULL;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}

__global__ void __launch_bounds__(256) fill_uniform_kernel(uint64_t *a, uint64_t n, uint64_t q, uint64_t seed,
                                                           uint64_t offset)
{
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    a[i] = splitmix64_dev(seed ^ (offset + i)) % q;
  }
}

/* one workgroup per polynomial: sum_i splitmix64(i) * a[i] mod 2^64 */
__global__ void __launch_bounds__(256) checksum_kernel(uint64_t *out, const uint64_t *a, uint64_t N, uint64_t batch)
{
  __shared__ uint64_t part[256];
  for(uint64_t p = blockIdx.x; p < batch; p += gridDim.x) {
    const uint64_t *src = a + p * N;
    uint64_t        acc = 0;
    for(uint64_t i = threadIdx.x; i < N; i += blockDim.x) acc += splitmix64_dev(i) * src[i];
    part[threadIdx.x] = acc;
    __syncthreads();
    for(int s = 128; s > 0; s >>= 1) {
      if((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
      __syncthreads();
    }
    if(threadIdx.x == 0) out[p] = part[0];
    __syncthreads();
  }
}

/* Element i of a batch of polynomials of 2^logn words: polynomial i >> logn starts (i >> logn) * stride words into its operand
 * (ntt_core.h block_offset with whole polynomials as blocks).  pstride: a, c and every operand laid out like them; bstride: the b
 * operand -- 0 when it is ONE polynomial shared by the batch (a broadcast key). */
struct PwLayout {
  uint32_t logn;
  uint64_t pstride, bstride;
};
__device__ __forceinline__ uint64_t pw_index(uint64_t i, uint32_t logn, uint64_t stride)
{
  return (i >> logn) * stride + (i & ((1ull << logn) - 1ull));
}

/* LAZYIN: operands in [0,4q) as ntt_fwd_batch_lazy leaves them; the product is always fully reduced */
template <class A, bool LAZYIN>
__global__ void __launch_bounds__(256) pointwise_kernel(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n, const PwLayout lay,
                                                        const typename A::consts k)
{
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t j = pw_index(i, lay.logn, lay.pstride);
    c[j]             = LAZYIN ? A::mulmod_full_lazy4(a[j], b[j], k) : A::mulmod_full(a[j], b[j], k);
  }
}

/* c = (ACC ? c : 0) + a * b: the unfused form of the inner product in the NTT domain (plans the fused kernel is not built
 * for: column-pass-only plans, the radix-4 formulation, N < 2^6).  lay.bstride = 0 when b is ONE polynomial shared by the batch. */
template <class A, bool LAZYIN, bool ACC>
__global__ void __launch_bounds__(256) pointwise_acc_kernel(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n, const PwLayout lay,
                                                            uint64_t q, const typename A::consts k)
{
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t j = pw_index(i, lay.logn, lay.pstride), jb = pw_index(i, lay.logn, lay.bstride);
    const uint64_t t = LAZYIN ? A::mulmod_full_lazy4(a[j], b[jb], k) : A::mulmod_full(a[j], b[jb], k);
    if constexpr(ACC) {
      const uint64_t v = c[j] + t; /* both canonical: < 2q < 2^64 */
      c[j]             = v < q ? v : v - q;
    } else {
      c[j] = t;
    }
  }
}

static unsigned grid_for(uint64_t n, unsigned cap = 256 * 32)
{
  uint64_t g = (n + 255) / 256;
  if(g > cap) g = cap;
  /* a launch's threads per dimension must stay below 2^32 (hipErrorInvalidConfiguration otherwise: the 128 GiB slab of
   * `bench.py --scaling strong --gpus 1` asked the one-word-per-thread copy probe for exactly 2^32); the kernels that use this are
   * grid-stride loops, so a smaller grid only means a second iteration */
  if(g > (1u << 24) - 1u) g = (1u << 24) - 1u;
  if(g == 0) g = 1;
  return (unsigned)g;
}
/* The grid of the pointwise kernels (element-wise, memory-bound, grid-stride loop): about FOUR iterations per workgroup.  Workgroups
 * that loop over a slab in step produce their traffic in bursts: with the 8192 workgroups of rounds 1-4 (dozens of iterations
 * each) these kernels reached 0.66-0.68 of the roofline, with four iterations each 0.73-0.75 at 0.25, 1 and 4 GB per operand, and
 * one-shot workgroups (no loop, what the dispatcher staggers best for a plain copy: 6.3 TB/s, tools/copy_variants.hip) 0.65-0.70
 * (profiles/r05/pointwise_grid.txt).  max_grid: NTT_OPT_MAX_GRID (0 = this rule). */
static unsigned grid_pw(uint64_t n, int max_grid)
{
  const uint64_t total = (n + 255) / 256;
  if(max_grid > 0) return (unsigned)(total < (uint64_t)max_grid ? (total ? total : 1) : (uint64_t)max_grid);
  uint64_t g = (total + 3) / 4;
  if(g < 2048) g = total < 2048 ? total : 2048;
  if(g > (1u << 22)) g = 1u << 22;
  return (unsigned)(g ? g : 1);
}

/* pstride: words between consecutive polynomials of all three operands (0 = dense: N) */
static int pointwise_launch(const ntt_plan *p, uint64_t *d_c, const uint64_t *d_a, const uint64_t *d_b, uint64_t batch,
                            void *stream, bool lazy_in, uint64_t pstride = 0)
{
  if(!p || !d_c || !d_a || !d_b) return fail(NTT_ERR_ARG, "null argument");
  if(batch == 0) return NTT_OK;
  USE_DEVICE(p->device);
  const uint64_t n = batch * p->N;
  const dim3     g(grid_pw(n, p->max_grid)), t(256);
  hipStream_t    st = (hipStream_t)stream;
  const PwLayout lay{(uint32_t)p->m, pstride ? pstride : p->N, pstride ? pstride : p->N};
  if(p->arith == NTT_ARITH_F64) {
    if(lazy_in) hipLaunchKernelGGL((pointwise_kernel<ArithF64, true>), g, t, 0, st, d_c, d_a, d_b, n, lay, p->cf);
    else hipLaunchKernelGGL((pointwise_kernel<ArithF64, false>), g, t, 0, st, d_c, d_a, d_b, n, lay, p->cf);
  } else {
    if(lazy_in) hipLaunchKernelGGL((pointwise_kernel<ArithU64, true>), g, t, 0, st, d_c, d_a, d_b, n, lay, p->cu);
    else hipLaunchKernelGGL((pointwise_kernel<ArithU64, false>), g, t, 0, st, d_c, d_a, d_b, n, lay, p->cu);
  }
  HIP_TRY(hipGetLastError());
  return NTT_OK;
}

extern "C" int ntt_pointwise_mul_batch(const ntt_plan *p, uint64_t *d_c, const uint64_t *d_a, const uint64_t *d_b,
                                       uint64_t batch, void *stream)
{
  return pointwise_launch(p, d_c, d_a, d_b, batch, stream, false);
}
extern "C" int ntt_pointwise_mul_batch_lazy(const ntt_plan *p, uint64_t *d_c, const uint64_t *d_a, const uint64_t *d_b,
                                            uint64_t batch, void *stream)
{
  return pointwise_launch(p, d_c, d_a, d_b, batch, stream, true);
}

/* one pass of a transform on nb polynomials starting at d (shared by run_transform's loop and the fused product) */
static int launch_one_pass(const ntt_plan *p, const Pass &ps, uint64_t *d, uint64_t nb, bool inverse, bool wide, bool lazy,
                           bool ends, void *stream, const LimbSet &ls)
{
  PassArgs pa{};
  pa.a        = d;
  pa.limbs    = ls.d;
  pa.nlimbs   = ls.n;
  pa.limb_stride = ls.stride;
  pa.poly_stride = ls.pstride;
  pa.batch    = nb;
  pa.logn     = (uint32_t)p->m;
  pa.fused    = ps.fused;
  pa.r        = ps.r;
  pa.s        = ps.s;
  pa.inverse  = inverse;
  pa.wide     = wide;
  pa.lastinv  = inverse && ps.s == 0;
  pa.lazy     = lazy;
  pa.ends     = ends;
  pa.max_grid = p->max_grid;
  pa.num_cus  = p->num_cus;
  pa.oversub  = p->block_oversub;
  pa.stream   = (hipStream_t)stream;
  hipError_t e = dispatch_pass(p, pa);
  if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  return NTT_OK;
}

/* c = a * b with the fused product kernels (FP64, N = 2^8 .. 2^17).
 *   N <= 2^14: ONE launch: a -> fwd, b -> fwd, product in registers, -> inv -> c.            24N bytes, 1 launch
 *              (NTT_OPT_FUSED_PRODUCT 2: a^ = fwd(a) (lazy words) by a launch of its own first: 40N bytes, 2 launches).
 *   N > 2^14 : a^ = fwd(a); per chunk: column stages on b, ONE launch over its 2^14- or 2^12-point blocks (fwd block * a^ block
 *              -> inverse block: the product is element-wise, so it fuses block by block), column stages of the
 *              inverse on c.                                                                   88N bytes, 5 launches
 *              (120N and 7 launches for fwd, fwd, pointwise, inv).
 *   N > 2^14, 2^23 coefficients per operand or more: everything, both forward transforms included, as the items of
 *              ONE launch (team_product_kernel<..., FOUR>): a^ never exists in memory.        48N bytes, 1 launch. */
/* ahat_given: d_a already holds a^ = fwd(a) (canonical or lazy words below 2^53): c = inv(fwd(b) (.) a^), the three-pass
 * forms of the kernels (ntt_mul_transformed_batch) */
static int fused_product(const ntt_plan *p, uint64_t *d_c, uint64_t *d_a, uint64_t *d_b, uint64_t batch, void *stream,
                         const LimbSet *set = nullptr, bool ahat_given = false)
{
  const LimbSet ls = set ? *set : LimbSet{p->limbrec.data(), 1, 0, 0};
  /* a^ = fwd(a).  The product kernels take a^ as lazy words v + 2q, v in (-2q, 2q); a canonical word c is the lazy word of
   * v = c - 2q, so the reduced forward transform is a valid producer too -- used where it is the faster launch (the
   * XCD-local kernel, N >= 2^15, large batches: +13..20 % over the per-pass lazy transform) */
  const bool canonical_a = team_applies(p, batch, false, false, false, ls.n, true);
  /* N >= 2^15, large batches: the chain as the item kinds of ONE launch (ntt_kernels.h: team_product_kernel) -- column
   * stages of b AND a, block products (both blocks through their twelve stages, product, inverse stages), inverse column
   * stages of c: a^ never exists in memory (48N instead of 64N bytes across the fabric).  NTT_OPT_FUSED_PRODUCT 2 keeps a's
   * forward transform as a launch of its own in front of the three-pass form (measurements, tests). */
  void *ctl  = nullptr;
  int   rc   = NTT_OK;
  if(canonical_a && !p->block_log) {
    rc = team_buffer(p, stream, 2 * batch * (uint64_t)ls.n, &ctl);
    if(rc) return rc;
  }
  const bool four = !ahat_given && ctl && p->fused_product != 2;
  /* N <= 2^14: a's coefficients go straight into the fused kernel, which takes both operands through the forward
   * stages (24N instead of 40N bytes, one launch; a is left as it was) */
  /* N > 2^14 below the one-launch form's batch: the same inside the block launch of every chunk -- a gets b's column
   * passes and the blocks of both operands meet in registers (72N instead of 88N bytes, 6 launches per chunk, no
   * transform of a in front) */
  const bool both = !ahat_given && !four && p->fused_product != 2;
  if(!four && !both && !ahat_given) {
    rc = run_transform(p, d_a, batch, false, false, stream, !canonical_a, &ls);
    if(rc) return rc;
  }
  USE_DEVICE(p->device);
  if(canonical_a && !p->block_log) {
    if(ctl) {
      ProdArgs pa{};
      pa.four        = four;
      pa.b           = d_b;
      pa.ahat        = d_a;
      pa.out         = d_c;
      pa.limbs       = ls.d;
      pa.nlimbs      = ls.n;
      pa.limb_stride = ls.stride;
      pa.poly_stride = ls.pstride;
      pa.batch       = batch;
      pa.logn        = (uint32_t)p->m;
      pa.a_lazy      = 1;
      pa.max_grid    = p->max_grid;
      pa.num_cus     = p->num_cus;
      pa.oversub     = p->block_oversub;
      pa.team_ctl    = ctl;
      /* three passes, four workgroups per CU: the lag that keeps second- and third-pass items from waiting is larger than
       * the transform's (measured, profiles/r03/sweep_product_lag.txt: flat optimum 12-14 at 2^17, 12-20 at 2^16, 20-24 at 2^15;
       * with both operands' column tiles in the first pass: 8 at 2^17, 12-16 at 2^16, 20-24 at 2^15) */
      pa.team_lag    = p->team_lag ? p->team_lag : (p->m == kTeamBlock + 3 ? 20 : (p->m == kTeamBlock + 4 ? 14 : (four ? 8 : 12)));
      pa.team_wpc    = p->team_wpc;
      pa.stream      = (hipStream_t)stream;
      const int  kc = eff_kcls(p);
      hipError_t e = kc == kWideClass ? launch_team_product<ArithF64W, 0>(pa)
                     : kc == 18       ? launch_team_product<ArithF64, 18>(pa)
                     : kc == 1        ? launch_team_product<ArithF64, 1>(pa)
                                           : launch_team_product<ArithF64, 0>(pa);
      if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
      return NTT_OK;
    }
  }
  /* blocks of the fused launch for N > 2^14: as for the transforms, stages are cheaper in the memory-bound column
   * passes than in the FP64-bound fused launch -- 2^12-point blocks where 4 column stages reach (measured +3..5 % at
   * 2^15 and 2^16; 2^13-point blocks at 2^17: -1 %, not used) */
  const int      pblk   = p->m > kFusedMax ? (p->block_log ? p->block_log : (p->m <= kFusedSmallBlock + 4 ? kFusedSmallBlock : kFusedLarge)) : p->m;
  const PassList L     = make_passes(p->m, false, pblk);
  uint64_t       chunk = batch;
  if(L.n > 1) {
    chunk = ((uint64_t)p->chunk_mib << 20) / (p->N * sizeof(uint64_t) * (uint64_t)ls.n);
    if(chunk < 1) chunk = 1;
    if(chunk > batch) chunk = batch;
  }
  for(uint64_t first = 0; first < batch; first += chunk) {
    const uint64_t nb  = batch - first < chunk ? batch - first : chunk;
    const uint64_t off = first * poly_words(p, ls);
    for(int k = 0; k + 1 < L.n; k++) { /* forward column passes of b (every pass but the last, which is the block pass) */
      rc = launch_one_pass(p, L.p[k], d_b + off, nb, false, false, false, false, stream, ls);
      if(!rc && both) rc = launch_one_pass(p, L.p[k], d_a + off, nb, false, false, false, false, stream, ls); /* ... and of a */
      if(rc) return rc;
    }
    ProdArgs pa{};
    pa.b        = d_b + off;
    pa.ahat     = d_a + off;
    pa.out      = d_c + off;
    pa.limbs    = ls.d;
    pa.nlimbs   = ls.n;
    pa.limb_stride = ls.stride;
    pa.poly_stride = ls.pstride;
    pa.batch    = nb;
    pa.logn     = (uint32_t)p->m;
    pa.block_log = (uint32_t)pblk;
    pa.a_lazy   = 1;
    pa.both     = both;
    pa.max_grid = p->max_grid;
    pa.num_cus  = p->num_cus;
    pa.oversub  = p->block_oversub;
    pa.stream   = (hipStream_t)stream;
    const int  kc = eff_kcls(p);
    hipError_t e = kc == kWideClass ? launch_product<ArithF64W, 0>(pa)
                   : kc == 18       ? launch_product<ArithF64, 18>(pa)
                   : kc == 1        ? launch_product<ArithF64, 1>(pa)
                                         : launch_product<ArithF64, 0>(pa);
    if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    for(int k = L.n - 2; k >= 0; k--) { /* inverse column passes of c, the last one ends the transform (N^-1) */
      rc = launch_one_pass(p, L.p[k], d_c + off, nb, true, false, false, k == 0, stream, ls);
      if(rc) return rc;
    }
  }
  return NTT_OK;
}

static bool fused_product_applies(const ntt_plan *p, const uint64_t *d_c, const uint64_t *d_a, const uint64_t *d_b, uint64_t batch)
{
  return p && p->fused_product && p->arith == NTT_ARITH_F64 && p->m >= 8 && p->m <= kFusedMax + 3 && !p->generic && p->has_fwd &&
         p->has_inv && d_a != d_b && d_a && d_b && d_c && batch;
}

static bool dot_kernel_applies(const ntt_plan *p);
static int  inv_dot(const ntt_plan *p, uint64_t *d_c, int k, const uint64_t *const *a, const uint64_t *const *b, uint64_t batch,
                    unsigned flags, void *stream, const LimbSet *set, uint64_t b_limb_stride);

/* pstride: words between consecutive polynomials of all three operands (0 = dense) */
static int negacyclic_mul_one(const ntt_plan *p, uint64_t *d_c, uint64_t *d_a, uint64_t *d_b, uint64_t batch, void *stream, uint64_t pstride)
{
  /* the chain never leaves the lazy domain: both forward transforms skip their final reduction,
   * the pointwise product takes [0,4q) operands, only the inverse's output is reduced.
   * d_a == d_b is a squaring: the operand is transformed once (transforming the shared buffer twice
   * would multiply fwd(fwd(a)) with itself) */
  if(!p) return fail(NTT_ERR_ARG, "null argument");
  const LimbSet own{p->limbrec.data(), 1, 0, pstride};
  if(p->arith == NTT_ARITH_U64_R4) {
    /* the reference's radix-4 formulation end to end: fwd_ntt_radix4 on both operands (canonical outputs), the pointwise
     * product, inv_ntt_radix4 */
    int rc4 = run_transform(p, d_a, batch, false, false, stream, false, &own);
    if(!rc4 && d_b != d_a) rc4 = run_transform(p, d_b, batch, false, false, stream, false, &own);
    if(!rc4) rc4 = pointwise_launch(p, d_c, d_a, d_b, batch, stream, false, pstride);
    if(!rc4) rc4 = run_transform(p, d_c, batch, true, false, stream, false, &own);
    return rc4;
  }
  if(fused_product_applies(p, d_c, d_a, d_b, batch)) return fused_product(p, d_c, d_a, d_b, batch, stream, &own);
  int rc = run_transform(p, d_a, batch, false, false, stream, true, &own);
  if(!rc && d_b != d_a) rc = run_transform(p, d_b, batch, false, false, stream, true, &own);
  if(!rc && dot_kernel_applies(p) && p->has_inv && d_c) {
    /* the products inside the inverse transform's first pass (dot_inv_kernel): 24N bytes instead of 40N for the last two
     * steps -- the integer policies (no one-launch product kernel), squarings, plans with the fused product switched off */
    const uint64_t *pa = d_a, *pb = d_b;
    return inv_dot(p, d_c, 1, &pa, &pb, batch, NTT_MUL_LAZY_IN, stream, &own, 0);
  }
  if(!rc) rc = pointwise_launch(p, d_c, d_a, d_b, batch, stream, true, pstride);
  if(!rc) rc = run_transform(p, d_c, batch, true, false, stream, false, &own);
  return rc;
}

extern "C" int ntt_negacyclic_mul_batch(const ntt_plan *p, uint64_t *d_c, uint64_t *d_a, uint64_t *d_b,
                                        uint64_t batch, void *stream)
{
  return negacyclic_mul_one(p, d_c, d_a, d_b, batch, stream, 0);
}

static int rns_check(int nlimbs, ntt_plan *const *plans)
{
  if(nlimbs <= 0 || !plans) return fail(NTT_ERR_ARG, "bad limb list");
  for(int l = 0; l < nlimbs; l++) {
    if(!plans[l] || plans[l]->N != plans[0]->N || plans[l]->device != plans[0]->device) {
      return fail(NTT_ERR_ARG, "RNS limbs must share N and device");
    }
  }
  return NTT_OK;
}

/* One launch chain for all limbs (SURVEY 8e "split primes"): possible when the limbs' plans agree in
 * everything a launch is shaped by -- size, device, arithmetic policy and headroom class (one kernel instantiation serves
 * every limb), and the plan options.  Primes of the same bit size always do.  Otherwise the limbs are looped. */
/* plans whose kernels have MULTI variants (several limbs per launch): the FP64 policies and the wide integer policy */
static bool multi_limb_plan(const ntt_plan *p)
{
  return p->arith == NTT_ARITH_F64 || (p->arith == NTT_ARITH_U64 && p->int_cls >= 0 && !p->generic);
}

static std::vector<unsigned char> rns_records(ntt_plan *const *plans, int first, int n);
static bool rns_compatible(const ntt_plan *a, const ntt_plan *b)
{
  /* (the headroom class may differ inside a policy: the run takes the coarsest one -- but the reduce-as-scheduled FP64 policy
   * for 52-bit moduli and the reference's integer butterflies are policies of their own) */
  return multi_limb_plan(a) && b->arith == a->arith && (b->kcls == kWideClass) == (a->kcls == kWideClass) &&
         (b->int_cls >= 0) == (a->int_cls >= 0) && b->m == a->m &&
         b->generic == a->generic && b->block_log == a->block_log && b->chunk_mib == a->chunk_mib && b->two_phase == a->two_phase &&
         b->fused_product == a->fused_product && b->max_grid == a->max_grid && b->block_oversub == a->block_oversub && b->rns_launch == a->rns_launch && b->dot_fused == a->dot_fused && b->has_fwd == a->has_fwd && b->has_inv == a->has_inv &&
         b->xcd_local == a->xcd_local && b->team_lag == a->team_lag && b->team_wpc == a->team_wpc && b->one_pass == a->one_pass;
}

/* Where the limbs and polynomials of an RNS operand live (words): polynomial p of limb l starts l * limb + p * poly words in.
 * [limb][batch][N] (the layout of the plain ntt_rns_* entry points): {batch * N, N}; SURVEY 8(d)'s [batch][prime][N] -- what an
 * FHE library holds: a ciphertext polynomial = its limbs side by side --: {N, limbs * N}.  Padded forms of either are fine. */
struct Layout {
  uint64_t limb, poly;
  const uint64_t *ptab = nullptr; /* pointer batch: device table of the polynomials' limb-0 addresses (poly = typical spacing, a hint) */
};
/* the strides must keep the (limb, polynomial) ranges apart: limb-major (a limb's polynomials inside its slab) or
 * polynomial-major (a polynomial's limbs inside its record) */
static int layout_check(uint64_t N, int nlimbs, uint64_t batch, const Layout &lay)
{
  if(lay.poly < N || lay.limb < N) return fail(NTT_ERR_ARG, "layout: strides must be at least N words");
  if(lay.poly > (1ull << 40) || lay.limb > (1ull << 40)) return fail(NTT_ERR_ARG, "layout: stride too large");
  const bool limb_major = batch <= 1 || lay.limb >= (batch - 1) * lay.poly + N;
  const bool poly_major = nlimbs <= 1 || lay.poly >= (uint64_t)(nlimbs - 1) * lay.limb + N;
  if(!limb_major && !poly_major) return fail(NTT_ERR_ARG, "layout: limbs and polynomials overlap");
  return NTT_OK;
}

/* The limb list as maximal RUNS of consecutive compatible limbs, at most kMaxLimbs each (the records one launch carries): a
 * modulus chain with one prime of another size -- a 60-bit first prime in front of 50-bit ones -- is served as that limb by
 * itself plus one launch per pass for the others, not limb by limb altogether.  {first, count}. */
static std::vector<std::pair<int, int>> rns_runs(int nlimbs, ntt_plan *const *plans)
{
  std::vector<std::pair<int, int>> runs;
  for(int i = 0; i < nlimbs;) {
    int j = i + 1;
    while(j < nlimbs && j - i < kMaxLimbs && rns_compatible(plans[i], plans[j])) j++;
    runs.emplace_back(i, j - i);
    i = j;
  }
  return runs;
}

/* calls set_fn(first, LimbSet) for every run that `pays(plan of the run's first limb, run length)` says one launch should
 * serve, one_fn(limb) for every other limb; stops at the first error */
template <class Pays, class SetFn, class OneFn>
static int rns_for_runs(int nlimbs, ntt_plan *const *plans, const Layout &lay, Pays pays, SetFn set_fn, OneFn one_fn)
{
  int rc = NTT_OK;
  for(const std::pair<int, int> &run : rns_runs(nlimbs, plans)) {
    const int first = run.first, n = run.second;
    if(n > 1 && pays(plans[first], n)) {
      const std::vector<unsigned char> recs = rns_records(plans, first, n);
      const LimbSet                    ls{recs.data(), n, lay.limb, lay.poly, lay.ptab};
      int kc = plans[first]->kcls, ic = plans[first]->int_cls;
      for(int l = first + 1; l < first + n; l++) {
        kc = plans[l]->kcls < kc ? plans[l]->kcls : kc;
        ic = plans[l]->int_cls < ic ? plans[l]->int_cls : ic;
      }
      t_run_kcls    = kc;
      t_run_int_cls = ic;
      rc            = set_fn(first, ls);
      t_run_kcls = t_run_int_cls = INT_MIN;
    } else {
      for(int l = first; !rc && l < first + n; l++) rc = one_fn(l);
    }
    if(rc) break;
  }
  return rc;
}

/* One launch for all limbs pays when a single limb's share cannot fill the chip by itself (a ciphertext: a few
 * polynomials x tens of primes); with thousands of polynomials per limb every per-limb launch fills it, and the
 * single-set kernels are the faster ones (no run-time limb index: ntt_kernels.h, MULTI).  NTT_OPT_RNS_LAUNCH 0 / 1 on the
 * run's first plan forces the one-launch / the per-limb form (tests, measurements). */
static bool rns_one_launch_pays(const ntt_plan *p, uint64_t batch)
{
  if(p->rns_launch >= 0) return p->rns_launch == 0;
  const uint64_t wg_equivalents = (batch * p->N) >> 12; /* 256-thread workgroups' worth of coefficients per limb */
  return wg_equivalents < 8ull * (uint64_t)p->num_cus;
}

/* Large per-limb batches at N = 2^15..2^17: the XCD-local launches (team_kernel, team_product_kernel) take the limb as part
 * of the queue entry, so a whole RNS set is ONE launch there too -- no launch tails between the limbs (measured 3.5 % of a
 * config-5 step, profiles/r03/ablations.txt (f)).  NTT_OPT_RNS_LAUNCH 1 keeps the per-limb launches. */
static bool rns_team_launch(const ntt_plan *p, int nlimbs, uint64_t batch, bool inverse, bool product)
{
  if(p->rns_launch == 1) return false;
  return team_applies(p, batch, inverse, false, false, nlimbs < kMaxLimbs ? nlimbs : kMaxLimbs, product) &&
         team_applies(p, batch, inverse, false, false, 1, product); /* (a limb's own share qualifies: same lag tuning) */
}

/* the records of limbs [first, first + n) as one host array */
static std::vector<unsigned char> rns_records(ntt_plan *const *plans, int first, int n)
{
  std::vector<unsigned char> all;
  for(int l = first; l < first + n; l++) all.insert(all.end(), plans[l]->limbrec.begin(), plans[l]->limbrec.end());
  return all;
}

/* the plan's own record as a one-limb set whose polynomials are lay.poly words apart */
static LimbSet own_set(const ntt_plan *p, const Layout &lay) { return LimbSet{p->limbrec.data(), 1, 0, lay.poly, lay.ptab}; }

static int rns_transform(int nlimbs, ntt_plan *const *plans, uint64_t *d_a, uint64_t batch, bool inverse, void *stream, const Layout &lay)
{
  int rc = rns_check(nlimbs, plans);
  if(rc || batch == 0) return rc;
  if(!d_a && !lay.ptab) return fail(NTT_ERR_ARG, "null argument");
  /* (a pointer batch: the images were checked one by one where the pointers were visible -- ptr_runs) */
  rc = lay.ptab ? NTT_OK : layout_check(plans[0]->N, nlimbs, batch, lay);
  if(rc) return rc;
  return rns_for_runs(
    nlimbs, plans, lay,
    [&](const ntt_plan *p, int n) { return rns_one_launch_pays(p, batch) || rns_team_launch(p, n, batch, inverse, false); },
    [&](int first, const LimbSet &ls) { return run_transform(plans[first], advance(d_a, (uint64_t)first * lay.limb), batch, inverse, false, stream, false, &ls); },
    [&](int l) {
      const LimbSet own = own_set(plans[l], lay);
      return run_transform(plans[l], advance(d_a, (uint64_t)l * lay.limb), batch, inverse, false, stream, false, &own);
    });
}

/* [limb][batch][N]: what the plain ntt_rns_* entry points take */
static Layout limb_major(ntt_plan *const *plans, int nlimbs, uint64_t batch)
{
  const uint64_t N = (nlimbs > 0 && plans && plans[0]) ? plans[0]->N : 0;
  return Layout{batch * N, N};
}

extern "C" int ntt_rns_fwd_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_a, uint64_t batch, void *stream)
{
  return rns_transform(nlimbs, plans, d_a, batch, false, stream, limb_major(plans, nlimbs, batch));
}

extern "C" int ntt_rns_inv_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_a, uint64_t batch, void *stream)
{
  return rns_transform(nlimbs, plans, d_a, batch, true, stream, limb_major(plans, nlimbs, batch));
}

extern "C" int ntt_rns_fwd_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_a, uint64_t limb_stride, uint64_t poly_stride,
                                         uint64_t batch, void *stream)
{
  return rns_transform(nlimbs, plans, d_a, batch, false, stream, Layout{limb_stride, poly_stride});
}

extern "C" int ntt_rns_inv_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_a, uint64_t limb_stride, uint64_t poly_stride,
                                         uint64_t batch, void *stream)
{
  return rns_transform(nlimbs, plans, d_a, batch, true, stream, Layout{limb_stride, poly_stride});
}

static int rns_negacyclic_mul(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, uint64_t *d_b, uint64_t batch, void *stream,
                              const Layout &lay)
{
  int rc = rns_check(nlimbs, plans);
  if(rc || batch == 0) return rc;
  if(!d_c || !d_a || !d_b) return fail(NTT_ERR_ARG, "null argument");
  rc = layout_check(plans[0]->N, nlimbs, batch, lay);
  if(rc) return rc;
  /* a run of the wide integer policy: both forward transforms (lazy words, in place -- the operands are scratch on this path as
   * they are for a single plan) and the products inside the inverse transform's first pass, each ONE launch over the run */
  auto int_run = [&](const ntt_plan *p) {
    return p->arith == NTT_ARITH_U64 && dot_kernel_applies(p) && p->has_fwd && p->has_inv;
  };
  return rns_for_runs(
    nlimbs, plans, lay,
    [&](const ntt_plan *p, int n) {
      if(fused_product_applies(p, d_c, d_a, d_b, batch)) {
        return rns_one_launch_pays(p, batch) || (rns_team_launch(p, n, batch, false, true) && !p->block_log);
      }
      return int_run(p) && rns_one_launch_pays(p, batch);
    },
    [&](int first, const LimbSet &ls) {
      const ntt_plan *p   = plans[first];
      const uint64_t  off = (uint64_t)first * lay.limb;
      if(fused_product_applies(p, d_c, d_a, d_b, batch)) return fused_product(p, d_c + off, d_a + off, d_b + off, batch, stream, &ls);
      int r = run_transform(p, d_a + off, batch, false, false, stream, true, &ls);
      if(!r && d_b != d_a) r = run_transform(p, d_b + off, batch, false, false, stream, true, &ls);
      const uint64_t *pa = d_a + off, *pb = d_b + off;
      if(!r) r = inv_dot(p, d_c + off, 1, &pa, &pb, batch, NTT_MUL_LAZY_IN, stream, &ls, ls.stride);
      return r;
    },
    [&](int l) {
      const uint64_t off = (uint64_t)l * lay.limb;
      return negacyclic_mul_one(plans[l], d_c + off, d_a + off, d_b + off, batch, stream, lay.poly);
    });
}

extern "C" int ntt_rns_negacyclic_mul_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a,
                                            uint64_t *d_b, uint64_t batch, void *stream)
{
  return rns_negacyclic_mul(nlimbs, plans, d_c, d_a, d_b, batch, stream, limb_major(plans, nlimbs, batch));
}

extern "C" int ntt_rns_negacyclic_mul_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, uint64_t *d_b,
                                                    uint64_t limb_stride, uint64_t poly_stride, uint64_t batch, void *stream)
{
  return rns_negacyclic_mul(nlimbs, plans, d_c, d_a, d_b, batch, stream, Layout{limb_stride, poly_stride});
}

/* ------------------------------------------------------------------ */
/* products of operands that are in the NTT domain      */
/* ------------------------------------------------------------------ */
static hipError_t dispatch_dot(const ntt_plan *p, const DotArgs &da)
{
  if(p->arith == NTT_ARITH_U64) {
    switch(eff_int_cls(p)) { /* (same tables; the wide policy's stages around its Barrett products) */
      case 3: return launch_dot<ArithU64X<3>, 3>(da);
      case 1: return launch_dot<ArithU64X<1>, 1>(da);
      case 0: return launch_dot<ArithU64X<0>, 0>(da);
      default: return launch_dot<ArithU64, 0>(da);
    }
  }
  switch(eff_kcls(p)) {
    case kWideClass: return launch_dot<ArithF64W, 0>(da);
    case 18: return launch_dot<ArithF64, 18>(da);
    case 1: return launch_dot<ArithF64, 1>(da);
    default: return launch_dot<ArithF64, 0>(da);
  }
}

/* plans the fused kernel serves: the radix-2 policies on blocks of 2^6 points and more */
static bool dot_kernel_applies(const ntt_plan *p)
{
  return (p->arith == NTT_ARITH_F64 || p->arith == NTT_ARITH_U64) && !p->generic && p->m >= kFusedMin && p->dot_fused != 0;
}

/* the automatic choice between the one-launch form of the NTT-domain products at N = 2^15..2^17 and the per-chunk launches, as
 * measured (profiles/r05/domain_bench_xcd_local.txt) */
static bool dot_team_pays(const ntt_plan *p, uint64_t polys, int k, bool bcast)
{
  /* (profiles/r05/domain_bench_xcd_local.txt, _batch.txt: +11..26 % from 2^25 coefficients per operand on -- 1024 / 512 / 512
   * polynomials at 2^15 / 2^16 / 2^17; half of that is even or a loss, the queues then being as long as the lag) */
  if(polys < 512 || (polys << p->m) < (1ull << 25)) return false;
  /* a key shared by the batch: its words come from the L2 and the row items get short -- the gain ends at three pairs (2^17: -3 %,
   * eight pairs -18 %) and between four and eight pairs at 2^16 (+12 % at three, -12 % at eight); 2^15 keeps +6 % at eight */
  if(bcast) return p->m == kTeamBlock + 3 || k <= (p->m == kTeamBlock + 4 ? 4 : 2);
  return true;
}

/* Polynomials between the two passes of a queue in team_dot_kernel.  Measured optimum: lag x polynomial size = 5-6 MiB per
 * queue at all three sizes (24 / 12 / 6 polynomials), a little more for one or two pairs, a shared key and the integer policy,
 * whose row items are short -- too short a lag and second-pass items wait for rows still in flight, too long and c has left the
 * Infinity Cache slice when it is read back.  Fine sweep, k = 1..4, three operand kinds: profiles/r05/teamdot_lag_fine.txt,
 * teamdot_lag_other.txt (a lag two steps off the optimum costs 5-15 %; the plain transform's 8-10 polynomials lose 22 % at 2^15). */
static int dot_team_lag(const ntt_plan *p, int k, bool bcast)
{
  const bool longer = k <= 2 || bcast || p->arith == NTT_ARITH_U64;
  switch(p->m - kTeamBlock) {
    case 3: return longer ? 24 : 20;
    case 4: return bcast ? 14 : longer ? 12 : 10;
    default: return (k == 1 || bcast) ? 8 : 6;
  }
}

/* c = inv(sum_i a_i^ (.) b_i^).  One launch up to N = 2^14: the products are formed where the inverse transform would
 * convert its input words (dot_inv_kernel).  Above: per 256 MiB chunk of c that kernel over the blocks (the product rides in
 * the inverse's first pass), then the inverse's column passes on c -- 16kN + 24N bytes instead of 24kN + 32N.  Plans the
 * kernel is not built for (column-pass-only, radix-4 formulation, N < 2^6) accumulate the products with pointwise
 * launches and run their own inverse transform.  ls: the limbs one launch serves; b_limb_stride: words between the limbs
 * of a b operand. */
static int inv_dot(const ntt_plan *p, uint64_t *d_c, int k, const uint64_t *const *a, const uint64_t *const *b, uint64_t batch,
                   unsigned flags, void *stream, const LimbSet *set = nullptr, uint64_t b_limb_stride = 0)
{
  if(!p || !d_c || !a || !b) return fail(NTT_ERR_ARG, "null argument");
  if(k < 1 || k > kMaxDot) return fail(NTT_ERR_ARG, "number of operand pairs must be 1 .. 32");
  if(flags & ~(unsigned)(NTT_MUL_LAZY_IN | NTT_MUL_B_BROADCAST)) return fail(NTT_ERR_ARG, "unknown flag");
  for(int i = 0; i < k; i++) {
    if(!a[i] || !b[i]) return fail(NTT_ERR_ARG, "null operand");
  }
  if(batch == 0) return NTT_OK;
  if(!p->has_inv) return fail(NTT_ERR_ARG, "plan lacks the inverse table");
  const bool lazy = (flags & NTT_MUL_LAZY_IN) != 0, bcast = (flags & NTT_MUL_B_BROADCAST) != 0;
  const LimbSet ls = set ? *set : LimbSet{p->limbrec.data(), 1, 0, 0};
  USE_DEVICE(p->device);
  if(!dot_kernel_applies(p) || (ls.n > 1 && !multi_limb_plan(p))) {
    if(ls.n != 1) return fail(NTT_ERR_UNSUPPORTED, "one launch over several limbs needs the FP64 policies or the wide integer policy");
    const uint64_t n  = batch * p->N;
    const dim3     g(grid_pw(n, p->max_grid)), t(256);
    hipStream_t    st = (hipStream_t)stream;
    const PwLayout lay{(uint32_t)p->m, poly_words(p, ls), bcast ? 0 : poly_words(p, ls)};
    for(int i = 0; i < k; i++) {
#define NTT_PW_ACC(A, CONSTS)                                                                                              \
  do {                                                                                                                     \
    if(lazy) {                                                                                                             \
      if(i) hipLaunchKernelGGL((pointwise_acc_kernel<A, true, true>), g, t, 0, st, d_c, a[i], b[i], n, lay, p->q, CONSTS);  \
      else hipLaunchKernelGGL((pointwise_acc_kernel<A, true, false>), g, t, 0, st, d_c, a[i], b[i], n, lay, p->q, CONSTS);  \
    } else {                                                                                                               \
      if(i) hipLaunchKernelGGL((pointwise_acc_kernel<A, false, true>), g, t, 0, st, d_c, a[i], b[i], n, lay, p->q, CONSTS); \
      else hipLaunchKernelGGL((pointwise_acc_kernel<A, false, false>), g, t, 0, st, d_c, a[i], b[i], n, lay, p->q, CONSTS); \
    }                                                                                                                      \
  } while(0)
      if(p->arith == NTT_ARITH_F64) NTT_PW_ACC(ArithF64, p->cf);
      else NTT_PW_ACC(ArithU64, p->cu);
#undef NTT_PW_ACC
      HIP_TRY(hipGetLastError());
    }
    return run_transform(p, d_c, batch, true, false, stream, false, &ls);
  }
  /* N = 2^15..2^17, batches that keep the eight queues busy: the blocks with the products AND the inverse's column stages as
   * the items of ONE launch (team_dot_kernel) instead of two launches per 128 / 256 MiB chunk.  NTT_OPT_XCD_LOCAL 1 / 0 forces
   * either form; while a stream is being captured without a control block of its own the per-chunk launches serve (team_buffer). */
  {
    const bool int_wide = p->arith == NTT_ARITH_U64 && p->int_cls >= 0;
    const uint64_t polys = batch * (uint64_t)ls.n;
    bool team = (p->arith == NTT_ARITH_F64 || int_wide) && p->m >= kTeamBlock + 3 && p->m <= kTeamBlock + 5 && !p->block_log && polys >= 64 &&
                polys < (1ull << 29) && ls.n <= kMaxLimbs;
    if(team && p->xcd_local >= 0) team = p->xcd_local == 1;
    else if(team) team = dot_team_pays(p, polys, k, bcast);
    void *ctl = nullptr;
    if(team) {
      int rc = team_buffer(p, stream, polys, &ctl);
      if(rc) return rc;
    }
    if(ctl) {
      DotArgs da{};
      da.out           = d_c;
      da.a             = a;
      da.b             = b;
      da.npairs        = k;
      da.lazy_in       = lazy;
      da.b_bcast       = bcast;
      da.limbs         = ls.d;
      da.nlimbs        = ls.n;
      da.limb_stride   = ls.stride;
      da.poly_stride   = ls.pstride;
      da.b_limb_stride = ls.n > 1 ? b_limb_stride : 0;
      da.batch         = batch;
      da.logn          = (uint32_t)p->m;
      da.block_log     = (uint32_t)kTeamBlock;
      da.max_grid      = p->max_grid;
      da.num_cus       = p->num_cus;
      da.team_ctl      = ctl;
      da.team_lag      = p->team_lag ? p->team_lag : dot_team_lag(p, k, bcast);
      da.team_wpc      = p->team_wpc;
      da.stream        = (hipStream_t)stream;
      hipError_t e = dispatch_dot(p, da);
      if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
      return NTT_OK;
    }
  }
  const int      pblk  = p->m > kFusedMax ? (p->block_log ? p->block_log : multi_pass_block(p->m, true, p->arith == NTT_ARITH_F64)) : p->m;
  const PassList L     = make_passes(p->m, false, pblk);
  uint64_t       chunk = batch;
  if(L.n > 1) {
    /* the chunk is what must still be in the Infinity Cache when the column pass reads c back; here the operands stream
     * through that cache as well (two to 2k times the chunk), so a smaller chunk than the transforms' 256 MiB pays at
     * N = 2^15 and 2^16 (measured: 128 MiB +4..7 %, 64 MiB the same for k = 3 and slower for k = 1, none at 2^17;
     * profiles/r04/dot_chunk.txt).  An explicit NTT_OPT_CHUNK_MIB is honoured as given. */
    const uint64_t mib = p->chunk_mib == 256 && p->m <= kFusedMax + 2 ? 128 : (uint64_t)p->chunk_mib;
    chunk              = (mib << 20) / (p->N * sizeof(uint64_t) * (uint64_t)ls.n);
    if(chunk < 1) chunk = 1;
    if(chunk > batch) chunk = batch;
  }
  const uint64_t *ca[kMaxDot], *cb[kMaxDot];
  for(uint64_t first = 0; first < batch; first += chunk) {
    const uint64_t nb  = batch - first < chunk ? batch - first : chunk;
    const uint64_t off = first * poly_words(p, ls);
    for(int i = 0; i < k; i++) {
      ca[i] = a[i] + off;
      cb[i] = bcast ? b[i] : b[i] + off;
    }
    DotArgs da{};
    da.out           = d_c + off;
    da.a             = ca;
    da.b             = cb;
    da.npairs        = k;
    da.lazy_in       = lazy;
    da.b_bcast       = bcast;
    da.limbs         = ls.d;
    da.nlimbs        = ls.n;
    da.limb_stride   = ls.stride;
    da.poly_stride   = ls.pstride;
    da.b_limb_stride = ls.n > 1 ? b_limb_stride : 0;
    da.batch         = nb;
    da.logn          = (uint32_t)p->m;
    da.block_log     = (uint32_t)pblk;
    da.max_grid      = p->max_grid;
    da.num_cus       = p->num_cus;
    da.oversub       = p->block_oversub;
    da.stream        = (hipStream_t)stream;
    hipError_t e = dispatch_dot(p, da);
    if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
    for(int j = L.n - 2; j >= 0; j--) { /* the inverse's column passes on c; the last one ends the transform (N^-1) */
      int rc = launch_one_pass(p, L.p[j], d_c + off, nb, true, false, false, j == 0, stream, ls);
      if(rc) return rc;
    }
  }
  return NTT_OK;
}

extern "C" int ntt_inv_dot_batch(const ntt_plan *p, uint64_t *d_c, int k, const uint64_t *const *d_ahat,
                                 const uint64_t *const *d_bhat, uint64_t batch, unsigned flags, void *stream)
{
  return inv_dot(p, d_c, k, d_ahat, d_bhat, batch, flags, stream);
}

extern "C" int ntt_inv_product_batch(const ntt_plan *p, uint64_t *d_c, const uint64_t *d_ahat, const uint64_t *d_bhat,
                                     uint64_t batch, unsigned flags, void *stream)
{
  return inv_dot(p, d_c, 1, &d_ahat, &d_bhat, batch, flags, stream);
}

/* c = inv(fwd(a) (.) b^): the product kernels' form with one operand already transformed (fused_product_kernel /
 * team_product_kernel without BOTH / FOUR); plans those kernels are not built for take fwd, pointwise, inv */
static int mul_transformed(const ntt_plan *p, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat, uint64_t batch, unsigned flags,
                           void *stream, const LimbSet *set = nullptr)
{
  if(!p || !d_c || !d_a || !d_bhat) return fail(NTT_ERR_ARG, "null argument");
  if(flags & ~(unsigned)NTT_MUL_LAZY_IN) return fail(NTT_ERR_ARG, "unknown flag");
  if(batch == 0) return NTT_OK;
  if(!p->has_fwd || !p->has_inv) return fail(NTT_ERR_ARG, "plan lacks a table");
  if(d_a == d_bhat) return fail(NTT_ERR_ARG, "the coefficient operand and the transformed operand must be different buffers");
  /* (the kernels take b^ as lazy words v + 2q: a canonical word c is the lazy word of v = c - 2q, so one form serves both) */
  if(fused_product_applies(p, d_c, d_bhat, d_a, batch)) {
    return fused_product(p, d_c, const_cast<uint64_t *>(d_bhat), d_a, batch, stream, set, true);
  }
  const bool lazy = (flags & NTT_MUL_LAZY_IN) != 0;
  if(set && set->n > 1) {
    /* a limb set of the wide integer policy: a's forward transform over the set (lazy words), then the products inside the
     * inverse transform's first pass (dot_inv_kernel) -- two launches for all limbs */
    if(!multi_limb_plan(p) || !dot_kernel_applies(p)) {
      return fail(NTT_ERR_UNSUPPORTED, "one launch over several limbs needs the FP64 policies or the wide integer policy");
    }
    int rcs = run_transform(p, d_a, batch, false, false, stream, true, set);
    const uint64_t *pa = d_a;
    if(!rcs) rcs = inv_dot(p, d_c, 1, &pa, &d_bhat, batch, NTT_MUL_LAZY_IN, stream, set, set->stride);
    return rcs;
  }
  const LimbSet own = set ? *set : LimbSet{p->limbrec.data(), 1, 0, 0};
  int rc = run_transform(p, d_a, batch, false, false, stream, p->arith != NTT_ARITH_U64_R4, &own);
  if(!rc) rc = pointwise_launch(p, d_c, d_a, d_bhat, batch, stream, lazy || p->arith != NTT_ARITH_U64_R4, own.pstride);
  if(!rc) rc = run_transform(p, d_c, batch, true, false, stream, false, &own);
  return rc;
}

extern "C" int ntt_mul_transformed_batch(const ntt_plan *p, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat, uint64_t batch,
                                         unsigned flags, void *stream)
{
  return mul_transformed(p, d_c, d_a, d_bhat, batch, flags, stream);
}

/* c^ = fwd(a) (.) b^ (+ c^), the result staying in the NTT domain: the forward transform with the product where it would
 * reduce and store its outputs (fwd_mul_kernel).  One launch up to N = 2^14; above, the forward column passes run on a in
 * place (a is scratch there) and the product rides in the block pass, the forward transform's last one.  Plans without the
 * kernel: forward transform of a in place, then a pointwise (accumulate) launch. */
/* the one-launch form of c^ = fwd(a) (.) b^ (+ c^) at N = 2^15..2^17 against the per-chunk launches: the automatic choice and the lag
 * between the two passes of a queue, as measured (profiles/r05/domain_bench_xcd_local_mul.txt) */
static bool mul_team_pays(const ntt_plan *p, uint64_t polys, bool bcast, bool acc)
{
  /* +8..16 % (FP64), +16..28 % (wide integer policy) on large batches, all four operand kinds alike; even at 2^26 coefficients
   * per operand (2048 / 1024 / 512 polynomials at 2^15 / 2^16 / 2^17: +0 / +5 / +9 %), a loss below
   * (domain_bench_xcd_local_mul_batch.txt) */
  (void)bcast, (void)acc;
  return polys >= 512 && (polys << p->m) >= (1ull << 26);
}
/* (flat between 8 and 16 polynomials at 2^15 and 2^16; 2^17 loses 1-2 % per step beyond 8) */
static int mul_team_lag(const ntt_plan *p) { return p->m == kTeamBlock + 3 ? 10 : 8; }

static int fwd_mul(const ntt_plan *p, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat, uint64_t batch, unsigned flags,
                   void *stream, const LimbSet *set = nullptr, uint64_t b_limb_stride = 0)
{
  if(!p || !d_c || !d_a || !d_bhat) return fail(NTT_ERR_ARG, "null argument");
  if(flags & ~(unsigned)(NTT_MUL_LAZY_IN | NTT_MUL_B_BROADCAST | NTT_MUL_ACCUMULATE)) return fail(NTT_ERR_ARG, "unknown flag");
  if(batch == 0) return NTT_OK;
  if(!p->has_fwd) return fail(NTT_ERR_ARG, "plan lacks the forward table");
  const bool lazy = (flags & NTT_MUL_LAZY_IN) != 0, bcast = (flags & NTT_MUL_B_BROADCAST) != 0, acc = (flags & NTT_MUL_ACCUMULATE) != 0;
  if(acc && d_c == d_a) return fail(NTT_ERR_ARG, "an accumulator cannot alias the coefficient operand");
  const LimbSet ls = set ? *set : LimbSet{p->limbrec.data(), 1, 0, 0};
  USE_DEVICE(p->device);
  if(!dot_kernel_applies(p) || (ls.n > 1 && !multi_limb_plan(p))) {
    if(ls.n != 1) return fail(NTT_ERR_UNSUPPORTED, "one launch over several limbs needs the FP64 policies or the wide integer policy");
    int rc = run_transform(p, d_a, batch, false, false, stream, false, &ls);
    if(rc) return rc;
    const uint64_t n  = batch * p->N;
    const dim3     g(grid_pw(n, p->max_grid)), t(256);
    hipStream_t    st = (hipStream_t)stream;
    const PwLayout lay{(uint32_t)p->m, poly_words(p, ls), bcast ? 0 : poly_words(p, ls)};
#define NTT_PW_MUL(A, CONSTS)                                                                                              \
  do {                                                                                                                     \
    if(lazy) {                                                                                                             \
      if(acc) hipLaunchKernelGGL((pointwise_acc_kernel<A, true, true>), g, t, 0, st, d_c, d_a, d_bhat, n, lay, p->q, CONSTS); \
      else hipLaunchKernelGGL((pointwise_acc_kernel<A, true, false>), g, t, 0, st, d_c, d_a, d_bhat, n, lay, p->q, CONSTS);  \
    } else {                                                                                                               \
      if(acc) hipLaunchKernelGGL((pointwise_acc_kernel<A, false, true>), g, t, 0, st, d_c, d_a, d_bhat, n, lay, p->q, CONSTS); \
      else hipLaunchKernelGGL((pointwise_acc_kernel<A, false, false>), g, t, 0, st, d_c, d_a, d_bhat, n, lay, p->q, CONSTS); \
    }                                                                                                                      \
  } while(0)
    if(p->arith == NTT_ARITH_F64) NTT_PW_MUL(ArithF64, p->cf);
    else NTT_PW_MUL(ArithU64, p->cu);
#undef NTT_PW_MUL
    HIP_TRY(hipGetLastError());
    return NTT_OK;
  }
  const int      pblk  = p->m > kFusedMax ? (p->block_log ? p->block_log : multi_pass_block(p->m, false, p->arith == NTT_ARITH_F64)) : p->m;
  const PassList L     = make_passes(p->m, false, pblk);
  uint64_t       chunk = batch;
  if(L.n > 1) {
    chunk = ((uint64_t)p->chunk_mib << 20) / (p->N * sizeof(uint64_t) * (uint64_t)ls.n);
    if(chunk < 1) chunk = 1;
    if(chunk > batch) chunk = batch;
  }
  const int ic = eff_int_cls(p), kc = eff_kcls(p);
  const auto dispatch = [&](const MulArgs &ma) {
    return p->arith == NTT_ARITH_U64 ? (ic == 3   ? launch_fwd_mul<ArithU64X<3>, 3>(ma)
                                        : ic == 1 ? launch_fwd_mul<ArithU64X<1>, 1>(ma)
                                        : ic == 0 ? launch_fwd_mul<ArithU64X<0>, 0>(ma)
                                                  : launch_fwd_mul<ArithU64, 0>(ma))
           : kc == kWideClass        ? launch_fwd_mul<ArithF64W, 0>(ma)
           : kc == 18                ? launch_fwd_mul<ArithF64, 18>(ma)
           : kc == 1                 ? launch_fwd_mul<ArithF64, 1>(ma)
                                     : launch_fwd_mul<ArithF64, 0>(ma);
  };
  /* N = 2^15..2^17, batches that keep the eight queues busy: the forward column stages of a and the blocks with the product as
   * the items of ONE launch (team_mul_kernel) instead of two launches per chunk.  NTT_OPT_XCD_LOCAL 1 / 0 forces either form; a
   * stream being captured without a control block of its own takes the per-chunk launches (team_buffer). */
  {
    const bool     int_wide = p->arith == NTT_ARITH_U64 && p->int_cls >= 0;
    const uint64_t polys    = batch * (uint64_t)ls.n;
    bool team = (p->arith == NTT_ARITH_F64 || int_wide) && p->m >= kTeamBlock + 3 && p->m <= kTeamBlock + 5 && !p->block_log && polys >= 64 &&
                polys < (1ull << 29) && ls.n <= kMaxLimbs;
    if(team && p->xcd_local >= 0) team = p->xcd_local == 1;
    else if(team) team = mul_team_pays(p, polys, bcast, acc);
    void *ctl = nullptr;
    if(team) {
      int rc = team_buffer(p, stream, polys, &ctl);
      if(rc) return rc;
    }
    if(ctl) {
      MulArgs ma{};
      ma.a             = d_a;
      ma.b             = d_bhat;
      ma.out           = d_c;
      ma.lazy_in       = lazy;
      ma.b_bcast       = bcast;
      ma.accumulate    = acc;
      ma.limbs         = ls.d;
      ma.nlimbs        = ls.n;
      ma.limb_stride   = ls.stride;
      ma.poly_stride   = ls.pstride;
      ma.b_limb_stride = ls.n > 1 ? b_limb_stride : 0;
      ma.batch         = batch;
      ma.logn          = (uint32_t)p->m;
      ma.block_log     = (uint32_t)kTeamBlock;
      ma.max_grid      = p->max_grid;
      ma.num_cus       = p->num_cus;
      ma.team_ctl      = ctl;
      ma.team_lag      = p->team_lag ? p->team_lag : mul_team_lag(p);
      ma.team_wpc      = p->team_wpc;
      ma.stream        = (hipStream_t)stream;
      hipError_t e = dispatch(ma);
      if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
      return NTT_OK;
    }
  }
  for(uint64_t first = 0; first < batch; first += chunk) {
    const uint64_t nb  = batch - first < chunk ? batch - first : chunk;
    const uint64_t off = first * poly_words(p, ls);
    for(int j = 0; j + 1 < L.n; j++) { /* forward column passes of a (every pass but the last, which is the block pass) */
      int rc = launch_one_pass(p, L.p[j], d_a + off, nb, false, false, false, false, stream, ls);
      if(rc) return rc;
    }
    MulArgs ma{};
    ma.a             = d_a + off;
    ma.b             = bcast ? d_bhat : d_bhat + off;
    ma.out           = d_c + off;
    ma.lazy_in       = lazy;
    ma.b_bcast       = bcast;
    ma.accumulate    = acc;
    ma.limbs         = ls.d;
    ma.nlimbs        = ls.n;
    ma.limb_stride   = ls.stride;
    ma.poly_stride   = ls.pstride;
    ma.b_limb_stride = ls.n > 1 ? b_limb_stride : 0;
    ma.batch         = nb;
    ma.logn          = (uint32_t)p->m;
    ma.block_log     = (uint32_t)pblk;
    ma.max_grid      = p->max_grid;
    ma.num_cus       = p->num_cus;
    ma.oversub       = p->block_oversub;
    ma.stream        = (hipStream_t)stream;
    hipError_t e = dispatch(ma);
    if(e != hipSuccess) return fail(NTT_ERR_HIP, std::string("kernel launch: ") + hipGetErrorString(e));
  }
  return NTT_OK;
}

extern "C" int ntt_fwd_mul_batch(const ntt_plan *p, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat, uint64_t batch,
                                 unsigned flags, void *stream)
{
  return fwd_mul(p, d_c, d_a, d_bhat, batch, flags, stream);
}

static int rns_fwd_mul(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat, uint64_t batch, unsigned flags,
                       void *stream, const Layout &lay)
{
  int rc = rns_check(nlimbs, plans);
  if(rc || batch == 0) return rc;
  if(!d_c || !d_a || !d_bhat) return fail(NTT_ERR_ARG, "null argument"); /* (before any limb offset is added) */
  rc = layout_check(plans[0]->N, nlimbs, batch, lay);
  if(rc) return rc;
  const uint64_t bslab = (flags & NTT_MUL_B_BROADCAST) ? plans[0]->N : lay.limb; /* a broadcast operand is [limb][N] */
  return rns_for_runs(
    nlimbs, plans, lay, [&](const ntt_plan *p, int) { return rns_one_launch_pays(p, batch) && dot_kernel_applies(p); },
    [&](int first, const LimbSet &ls) {
      return fwd_mul(plans[first], d_c + (uint64_t)first * lay.limb, d_a + (uint64_t)first * lay.limb, d_bhat + (uint64_t)first * bslab, batch, flags,
                     stream, &ls, bslab);
    },
    [&](int l) {
      const LimbSet own = own_set(plans[l], lay);
      return fwd_mul(plans[l], d_c + (uint64_t)l * lay.limb, d_a + (uint64_t)l * lay.limb, d_bhat + (uint64_t)l * bslab, batch, flags, stream, &own);
    });
}

extern "C" int ntt_rns_fwd_mul_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat,
                                     uint64_t batch, unsigned flags, void *stream)
{
  return rns_fwd_mul(nlimbs, plans, d_c, d_a, d_bhat, batch, flags, stream, limb_major(plans, nlimbs, batch));
}

extern "C" int ntt_rns_fwd_mul_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat,
                                             uint64_t limb_stride, uint64_t poly_stride, uint64_t batch, unsigned flags, void *stream)
{
  return rns_fwd_mul(nlimbs, plans, d_c, d_a, d_bhat, batch, flags, stream, Layout{limb_stride, poly_stride});
}

static int rns_inv_dot(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, int k, const uint64_t *const *d_ahat, const uint64_t *const *d_bhat,
                       uint64_t batch, unsigned flags, void *stream, const Layout &lay)
{
  int rc = rns_check(nlimbs, plans);
  if(rc || batch == 0) return rc;
  if(k < 1 || k > kMaxDot || !d_ahat || !d_bhat) return fail(NTT_ERR_ARG, "number of operand pairs must be 1 .. 32");
  if(!d_c) return fail(NTT_ERR_ARG, "null argument");
  for(int i = 0; i < k; i++) {
    if(!d_ahat[i] || !d_bhat[i]) return fail(NTT_ERR_ARG, "null operand"); /* (before any limb offset is added) */
  }
  rc = layout_check(plans[0]->N, nlimbs, batch, lay);
  if(rc) return rc;
  const uint64_t bslab = (flags & NTT_MUL_B_BROADCAST) ? plans[0]->N : lay.limb; /* a broadcast operand is [limb][N] */
  auto call = [&](int first, const LimbSet &ls, bool run) {
    const uint64_t *la[kMaxDot], *lb[kMaxDot];
    for(int i = 0; i < k; i++) {
      la[i] = d_ahat[i] + (uint64_t)first * lay.limb;
      lb[i] = d_bhat[i] + (uint64_t)first * bslab;
    }
    return inv_dot(plans[first], d_c + (uint64_t)first * lay.limb, k, la, lb, batch, flags, stream, &ls, run ? bslab : 0);
  };
  return rns_for_runs(
    nlimbs, plans, lay, [&](const ntt_plan *p, int) { return rns_one_launch_pays(p, batch) && dot_kernel_applies(p); },
    [&](int first, const LimbSet &ls) { return call(first, ls, true); }, [&](int l) { return call(l, own_set(plans[l], lay), false); });
}

extern "C" int ntt_rns_inv_dot_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, int k, const uint64_t *const *d_ahat,
                                     const uint64_t *const *d_bhat, uint64_t batch, unsigned flags, void *stream)
{
  return rns_inv_dot(nlimbs, plans, d_c, k, d_ahat, d_bhat, batch, flags, stream, limb_major(plans, nlimbs, batch));
}

extern "C" int ntt_rns_inv_dot_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, int k, const uint64_t *const *d_ahat,
                                             const uint64_t *const *d_bhat, uint64_t limb_stride, uint64_t poly_stride, uint64_t batch,
                                             unsigned flags, void *stream)
{
  return rns_inv_dot(nlimbs, plans, d_c, k, d_ahat, d_bhat, batch, flags, stream, Layout{limb_stride, poly_stride});
}

static int rns_mul_transformed(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat, uint64_t batch,
                               unsigned flags, void *stream, const Layout &lay)
{
  int rc = rns_check(nlimbs, plans);
  if(rc || batch == 0) return rc;
  if(!d_c || !d_a || !d_bhat) return fail(NTT_ERR_ARG, "null argument"); /* (before any limb offset is added) */
  rc = layout_check(plans[0]->N, nlimbs, batch, lay);
  if(rc) return rc;
  /* one launch over a run: the fused product kernels (FP64 policies), or -- limbs of the wide integer policy -- the forward
   * transform and the products-inside-the-inverse launch, each over the whole run */
  return rns_for_runs(
    nlimbs, plans, lay,
    [&](const ntt_plan *p, int) {
      const bool int_set = p->arith == NTT_ARITH_U64 && dot_kernel_applies(p) && d_a != d_bhat;
      return rns_one_launch_pays(p, batch) && (fused_product_applies(p, d_c, d_bhat, d_a, batch) || int_set);
    },
    [&](int first, const LimbSet &ls) {
      const uint64_t off = (uint64_t)first * lay.limb;
      return mul_transformed(plans[first], d_c + off, d_a + off, d_bhat + off, batch, flags, stream, &ls);
    },
    [&](int l) {
      const uint64_t off = (uint64_t)l * lay.limb;
      const LimbSet  own = own_set(plans[l], lay);
      return mul_transformed(plans[l], d_c + off, d_a + off, d_bhat + off, batch, flags, stream, &own);
    });
}

extern "C" int ntt_rns_mul_transformed_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat,
                                             uint64_t batch, unsigned flags, void *stream)
{
  return rns_mul_transformed(nlimbs, plans, d_c, d_a, d_bhat, batch, flags, stream, limb_major(plans, nlimbs, batch));
}

extern "C" int ntt_rns_mul_transformed_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat,
                                                     uint64_t limb_stride, uint64_t poly_stride, uint64_t batch, unsigned flags, void *stream)
{
  return rns_mul_transformed(nlimbs, plans, d_c, d_a, d_bhat, batch, flags, stream, Layout{limb_stride, poly_stride});
}

/* ---- one plan, polynomials poly_stride words apart (one limb of a caller-native layout) ---- */
extern "C" int ntt_transform_batch_strided(const ntt_plan *p, uint64_t *d_a, uint64_t poly_stride, uint64_t batch, unsigned flags, void *stream)
{
  if(flags & ~(unsigned)(NTT_FLAG_INVERSE | NTT_FLAG_WIDE_IN | NTT_FLAG_LAZY_OUT)) return fail(NTT_ERR_ARG, "unknown flag");
  if(!p) return fail(NTT_ERR_ARG, "null argument");
  if(poly_stride < p->N || poly_stride > (1ull << 40)) return fail(NTT_ERR_ARG, "layout: the polynomial stride must be at least N words");
  const LimbSet own{p->limbrec.data(), 1, 0, poly_stride};
  return run_transform(p, d_a, batch, (flags & NTT_FLAG_INVERSE) != 0, (flags & NTT_FLAG_WIDE_IN) != 0, stream, (flags & NTT_FLAG_LAZY_OUT) != 0, &own);
}

/* ---- pointer batches: one device pointer per polynomial ----
 * The reference's batching precedent hands over one array per polynomial: fwd_ntt_ref_harvey_lazy_dbl(a1[], a2[], ...)
 * (include/ntt_reference.h:44-49, src/ntt_reference.c:71-91).  Generalised to `count` pointers placed anywhere: ONE launch chain
 * for the whole batch -- the kernels read a polynomial's address from a device table where they would multiply its index by the
 * stride (ntt_core.h poly_offset; round 6: rounds 1-5 cut the sorted pointers into arithmetic progressions and launched every
 * progression by itself, i.e. 4096 separately allocated polynomials = 4096 launches).
 *   host array  (ntt_transform_ptrs):     sorted (the polynomials are independent and transformed in place: their order is free, and
 *                address order is what the memory system likes), checked for overlap, uploaded through the plan's staging buffer of
 *                the stream.  A batch that IS one arithmetic progression keeps the strided launch (no table).  While the stream is
 *                being captured nothing can be uploaded: progression by progression then -- use the device-array form in graphs.
 *   device array (ntt_transform_dev_ptrs): taken as it is -- no copy, no check (overlaps are the caller's business), capturable. */
/* ascending order.  The host side of a pointer batch is on the caller's clock (the kernels wait for the table): a batch that arrives
 * sorted costs one pass; large shuffled batches an LSD radix sort over the bits in which the addresses differ (65536 pointers: 0.3 ms
 * where std::sort took 2.3) */
static void sort_addresses(std::vector<uintptr_t> &v)
{
  if(std::is_sorted(v.begin(), v.end())) return;
  if(v.size() < 4096) {
    std::sort(v.begin(), v.end());
    return;
  }
  uintptr_t lo = v[0], hi = v[0];
  for(uintptr_t a : v) lo = a < lo ? a : lo, hi = a > hi ? a : hi;
  std::vector<uintptr_t> tmp(v.size());
  for(unsigned shift = 3; shift < 64 && ((hi - lo) >> shift) != 0; shift += 11) { /* (8-byte aligned: the low three bits carry nothing) */
    size_t count[2049] = {};
    for(uintptr_t a : v) count[(((a - lo) >> shift) & 2047u) + 1]++;
    for(int i = 0; i < 2048; i++) count[i + 1] += count[i];
    for(uintptr_t a : v) tmp[count[((a - lo) >> shift) & 2047u]++] = a;
    v.swap(tmp);
  }
}

struct PtrRun {
  uint64_t *first;
  uint64_t  stride, count; /* words between consecutive polynomials (0: a single one), polynomials */
};
static int ptr_sorted(uint64_t N, int nlimbs, uint64_t limb_stride, uint64_t *const *h_polys, uint64_t count, std::vector<uintptr_t> &v)
{
  if(!h_polys) return fail(NTT_ERR_ARG, "null argument");
  v.resize(count);
  for(uint64_t i = 0; i < count; i++) {
    if(!h_polys[i] || ((uintptr_t)h_polys[i] & 7)) return fail(NTT_ERR_ARG, "pointer batch: null or misaligned polynomial pointer");
    v[i] = (uintptr_t)h_polys[i];
  }
  sort_addresses(v);
  /* overlap: every limb image [pointer + l * limb_stride, + N) of every polynomial is an interval of N words; sorted by start,
   * two of them intersect iff two neighbours do.  (Exact also for pointers INTO a [limb][batch][N] slab, whose polynomials
   * interleave without overlapping.) */
  std::vector<uintptr_t> img;
  const std::vector<uintptr_t> *starts = &v;
  if(nlimbs > 1) {
    img.reserve((size_t)count * (size_t)nlimbs);
    for(uint64_t i = 0; i < count; i++) {
      for(int l = 0; l < nlimbs; l++) img.push_back(v[i] + (uintptr_t)l * (uintptr_t)limb_stride * 8u);
    }
    sort_addresses(img);
    starts = &img;
  }
  for(size_t i = 0; i + 1 < starts->size(); i++) {
    if((*starts)[i + 1] - (*starts)[i] < N * 8) return fail(NTT_ERR_ARG, "pointer batch: polynomials overlap (or a pointer is listed twice)");
  }
  return NTT_OK;
}
/* the sorted addresses as maximal arithmetic progressions (the fallback while capturing, and the test for "one progression").  A
 * run of two is not taken greedily: {0, 100N, 101N, 102N} is 1 + 3, not 2 + 2 -- when the pair's successor starts a longer
 * progression of another step, the first pointer goes by itself.  Steps beyond the strided entry points' bound (2^40 words)
 * never join a run. */
static void ptr_runs(const std::vector<uintptr_t> &v, std::vector<PtrRun> &runs)
{
  const uint64_t count = v.size();
  const auto     step  = [&](uint64_t i) { return (uint64_t)(v[i + 1] - v[i]); };
  const auto     ok    = [&](uint64_t d) { return d / 8 <= (1ull << 40); };
  for(uint64_t i = 0; i < count;) {
    uint64_t j = i, d = 0;
    if(i + 1 < count && ok(step(i))) {
      d = step(i);
      j = i + 1;
      while(j + 1 < count && step(j) == d) j++;
      if(j == i + 1 && j + 2 < count && step(j) != d && step(j) == step(j + 1)) j = i, d = 0; /* the pair's second member starts a longer run */
    }
    runs.push_back(PtrRun{reinterpret_cast<uint64_t *>(v[i]), d / 8, j - i + 1});
    i = j + 1;
  }
}
/* uploads the sorted addresses into the (plan, stream) device table; *out = the table, or null while the stream is being captured */
static int ptr_table(const ntt_plan *p, void *stream, const std::vector<uintptr_t> &v, const uint64_t **out)
{
  *out = nullptr;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if(hipStreamIsCapturing((hipStream_t)stream, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) return NTT_OK;
  std::lock_guard<std::mutex> lock(p->team_mu);
  ntt_plan::PtrBuf *pb = nullptr;
  for(ntt_plan::PtrBuf &b : p->ptr_bufs) {
    if(b.stream == stream) pb = &b;
  }
  if(!pb) {
    p->ptr_bufs.push_back(ntt_plan::PtrBuf{stream, nullptr, nullptr, 0, nullptr});
    pb = &p->ptr_bufs.back();
    HIP_TRY(hipEventCreateWithFlags(&pb->copied, hipEventDisableTiming));
  }
  if(pb->words < v.size()) {
    const size_t words = v.size() * 2;
    uint64_t *   d = nullptr, *h = nullptr;
    HIP_TRY(hipMalloc(&d, words * 8));
    if(hipHostMalloc(&h, words * 8, hipHostMallocDefault) != hipSuccess) {
      (void)hipFree(d);
      return fail(NTT_ERR_NOMEM, "pointer batch: pinned staging buffer");
    }
    if(pb->d) p->ptr_retired.push_back({pb->d, pb->h}); /* launches and copies still queued may be using them */
    pb->d = d, pb->h = h, pb->words = words;
  } else {
    HIP_TRY(hipEventSynchronize(pb->copied)); /* the previous upload has left the staging buffer (an event never recorded is complete) */
  }
  for(size_t i = 0; i < v.size(); i++) pb->h[i] = (uint64_t)v[i];
  HIP_TRY(hipMemcpyAsync(pb->d, pb->h, v.size() * 8, hipMemcpyHostToDevice, (hipStream_t)stream));
  HIP_TRY(hipEventRecord(pb->copied, (hipStream_t)stream));
  *out = pb->d;
  return NTT_OK;
}
/* typical distance between neighbours of a sorted batch, in words: the hint the XCD-local launches number their queues by */
static uint64_t ptr_spacing(const std::vector<uintptr_t> &v, uint64_t fallback)
{
  return v.size() > 1 ? (uint64_t)(v[v.size() / 2] - v[v.size() / 2 - 1]) / 8 : fallback;
}

extern "C" int ntt_transform_ptrs(const ntt_plan *p, uint64_t *const *h_polys, uint64_t count, unsigned flags, void *stream)
{
  if(flags & ~(unsigned)(NTT_FLAG_INVERSE | NTT_FLAG_WIDE_IN | NTT_FLAG_LAZY_OUT)) return fail(NTT_ERR_ARG, "unknown flag");
  if(!p) return fail(NTT_ERR_ARG, "null argument");
  if(count == 0) return NTT_OK;
  USE_DEVICE(p->device);
  std::vector<uintptr_t> v;
  int rc = ptr_sorted(p->N, 1, 0, h_polys, count, v);
  if(rc) return rc;
  std::vector<PtrRun> runs;
  ptr_runs(v, runs);
  const bool inverse = (flags & NTT_FLAG_INVERSE) != 0, wide = (flags & NTT_FLAG_WIDE_IN) != 0, lazy = (flags & NTT_FLAG_LAZY_OUT) != 0;
  const uint64_t *tab = nullptr;
  if(runs.size() > 1) {
    rc = ptr_table(p, stream, v, &tab);
    if(rc) return rc;
  }
  if(tab) {
    const LimbSet set{p->limbrec.data(), 1, 0, 0, tab};
    return run_transform(p, nullptr, count, inverse, wide, stream, lazy, &set);
  }
  for(size_t r = 0; !rc && r < runs.size(); r++) {
    const LimbSet own{p->limbrec.data(), 1, 0, runs[r].stride};
    rc = run_transform(p, runs[r].first, runs[r].count, inverse, wide, stream, lazy, &own);
  }
  return rc;
}

extern "C" int ntt_transform_dev_ptrs(const ntt_plan *p, const uint64_t *const *d_polys, uint64_t count, unsigned flags, void *stream)
{
  if(flags & ~(unsigned)(NTT_FLAG_INVERSE | NTT_FLAG_WIDE_IN | NTT_FLAG_LAZY_OUT)) return fail(NTT_ERR_ARG, "unknown flag");
  if(!p || (!d_polys && count)) return fail(NTT_ERR_ARG, "null argument");
  if(count == 0) return NTT_OK;
  const LimbSet set{p->limbrec.data(), 1, 0, 0, reinterpret_cast<const uint64_t *>(d_polys)};
  return run_transform(p, nullptr, count, (flags & NTT_FLAG_INVERSE) != 0, (flags & NTT_FLAG_WIDE_IN) != 0, stream, (flags & NTT_FLAG_LAZY_OUT) != 0, &set);
}

/* the same for RNS polynomials: h_polys[i] points at limb 0 of polynomial i, its limbs limb_stride words apart */
extern "C" int ntt_rns_transform_ptrs(int nlimbs, ntt_plan *const *plans, uint64_t *const *h_polys, uint64_t count, uint64_t limb_stride,
                                      unsigned flags, void *stream)
{
  if(flags & ~(unsigned)NTT_FLAG_INVERSE) return fail(NTT_ERR_ARG, "unknown flag (RNS transforms take NTT_FLAG_INVERSE only)");
  int rc = rns_check(nlimbs, plans);
  if(rc || count == 0) return rc;
  const uint64_t N = plans[0]->N;
  if(limb_stride < N || limb_stride > (1ull << 40)) return fail(NTT_ERR_ARG, "layout: the limb stride must be at least N words");
  USE_DEVICE(plans[0]->device);
  const uint64_t span = (uint64_t)(nlimbs - 1) * limb_stride + N; /* words one RNS polynomial covers */
  const bool inverse  = (flags & NTT_FLAG_INVERSE) != 0;
  std::vector<uintptr_t> v;
  rc = ptr_sorted(N, nlimbs, limb_stride, h_polys, count, v);
  if(rc) return rc;
  std::vector<PtrRun> runs;
  ptr_runs(v, runs);
  /* a progression the strided entry point takes whole is limb-major (the limbs' slabs apart: pointers into a [limb][batch][N]
   * slab) or polynomial-major (every polynomial's limbs inside its own record); any other spacing is legal here -- the
   * images were checked one by one above -- but not expressible as ONE layout */
  const auto whole = [&](const PtrRun &r) { return r.count <= 1 || limb_stride >= (r.count - 1) * r.stride + N || r.stride >= span; };
  const uint64_t *tab = nullptr;
  if(runs.size() > 1 || !whole(runs[0])) {
    rc = ptr_table(plans[0], stream, v, &tab);
    if(rc) return rc;
  }
  if(tab) return rns_transform(nlimbs, plans, nullptr, count, inverse, stream, Layout{limb_stride, ptr_spacing(v, span), tab});
  for(size_t r = 0; !rc && r < runs.size(); r++) {
    const uint64_t d = runs[r].stride ? runs[r].stride : span;
    if(whole(runs[r])) {
      rc = rns_transform(nlimbs, plans, runs[r].first, runs[r].count, inverse, stream, Layout{limb_stride, runs[r].count > 1 ? d : span});
    } else {
      for(uint64_t i = 0; !rc && i < runs[r].count; i++) rc = rns_transform(nlimbs, plans, runs[r].first + i * d, 1, inverse, stream, Layout{limb_stride, span});
    }
  }
  return rc;
}

extern "C" int ntt_rns_transform_dev_ptrs(int nlimbs, ntt_plan *const *plans, const uint64_t *const *d_polys, uint64_t count, uint64_t limb_stride,
                                          unsigned flags, void *stream)
{
  if(flags & ~(unsigned)NTT_FLAG_INVERSE) return fail(NTT_ERR_ARG, "unknown flag (RNS transforms take NTT_FLAG_INVERSE only)");
  int rc = rns_check(nlimbs, plans);
  if(rc || count == 0) return rc;
  if(!d_polys) return fail(NTT_ERR_ARG, "null argument");
  const uint64_t N = plans[0]->N;
  if(limb_stride < N || limb_stride > (1ull << 40)) return fail(NTT_ERR_ARG, "layout: the limb stride must be at least N words");
  /* (spacing hint: a polynomial's limbs side by side unless the limb stride says they are slabs apart) */
  const uint64_t span = (uint64_t)(nlimbs - 1) * limb_stride + N;
  return rns_transform(nlimbs, plans, nullptr, count, (flags & NTT_FLAG_INVERSE) != 0, stream,
                       Layout{limb_stride, limb_stride >= count * N ? N : span, reinterpret_cast<const uint64_t *>(d_polys)});
}

extern "C" int ntt_fill_uniform(int device, uint64_t *d_a, uint64_t n, uint64_t q, uint64_t seed, uint64_t offset,
                                void *stream)
{
  int rc = check_device(device);
  if(rc) return rc;
  if(!d_a || q == 0) return fail(NTT_ERR_ARG, "bad argument");
  USE_DEVICE(device);
  hipLaunchKernelGGL(fill_uniform_kernel, dim3(grid_for(n, 256 * 32)), dim3(256), 0, (hipStream_t)stream, d_a, n, q, seed,
                     offset);
  HIP_TRY(hipGetLastError());
  return NTT_OK;
}

extern "C" int ntt_poly_checksum(int device, uint64_t *d_out, const uint64_t *d_a, uint64_t N, uint64_t batch,
                                 void *stream)
{
  int rc = check_device(device);
  if(rc) return rc;
  if(!d_out || !d_a) return fail(NTT_ERR_ARG, "null argument");
  if(batch == 0) return NTT_OK;
  USE_DEVICE(device);
  const unsigned g = batch > 8192 ? 8192u : (unsigned)batch;
  hipLaunchKernelGGL(checksum_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, d_out, d_a, N, batch);
  HIP_TRY(hipGetLastError());
  return NTT_OK;
}

struct alignas(16) U64x2 {
  uint64_t a, b;
};
__global__ void __launch_bounds__(256) rmw_probe_kernel(U64x2 *a, uint64_t n2, uint64_t mask)
{
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * blockDim.x) {
    U64x2 v = a[i];
    v.a ^= mask;
    v.b ^= mask;
    a[i] = v;
  }
}

extern "C" int ntt_rmw_probe(int device, uint64_t *d_a, uint64_t n, uint64_t mask, void *stream)
{
  int rc = check_device(device);
  if(rc) return rc;
  if(!d_a || (n & 1) || ((uintptr_t)d_a & 15)) return fail(NTT_ERR_ARG, "rmw probe: null, odd length or unaligned buffer");
  if(n == 0) return NTT_OK;
  USE_DEVICE(device);
  /* one workgroup per 4 KiB up to 65536 workgroups: the fastest of the grids tried (profiles/r02/skeleton.txt) */
  hipLaunchKernelGGL(rmw_probe_kernel, dim3(grid_for(n / 2, 65536u)), dim3(256), 0, (hipStream_t)stream, (U64x2 *)d_a, n / 2, mask);
  HIP_TRY(hipGetLastError());
  return NTT_OK;
}

/* out-of-place copy, 16 bytes per lane, plain grid-stride: the shape the microarchitecture guide quotes its achievable HBM rate
 * for (MI355X_MICROARCH.md: about 6.3 TB/s of read + written bytes) */
__global__ void __launch_bounds__(256) copy_probe_kernel(U64x2 *dst, const U64x2 *src, uint64_t n2)
{
  /* (one 16-byte word per thread where the grid allows it, i.e. always in practice: the fastest of eight copy shapes measured,
   * 6.3 TB/s -- a grid-stride loop over the same slab 4.8-5.5, tools/copy_variants.hip) */
  for(uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int ntt_copy_probe(int device, uint64_t *d_dst, const uint64_t *d_src, uint64_t n, void *stream)
{
  int rc = check_device(device);
  if(rc) return rc;
  if(!d_dst || !d_src || (n & 1) || (((uintptr_t)d_dst | (uintptr_t)d_src) & 15)) return fail(NTT_ERR_ARG, "copy probe: null, odd length or unaligned buffer");
  if(n == 0) return NTT_OK;
  USE_DEVICE(device);
  hipLaunchKernelGGL(copy_probe_kernel, dim3(grid_for(n / 2, 1u << 24)), dim3(256), 0, (hipStream_t)stream, (U64x2 *)d_dst, (const U64x2 *)d_src, n / 2);
  HIP_TRY(hipGetLastError());
  return NTT_OK;
}

/* The memory shape of the 2^14 block kernels without their arithmetic (tools/skel.hip "T1024 C16 ld16 st1 mode1": the best
 * memory-only skeleton measured, profiles/r02/skeleton.txt): one persistent 1024-thread workgroup per CU, a block of 2^14
 * words per iteration as eight 16-byte loads per thread with the NEXT block's loads in flight (register prefetch), XOR,
 * eight 16-byte stores -- every load and store instruction of a wave covers one contiguous KiB. */
__global__ void __launch_bounds__(1024, 4) shape_probe_kernel(uint64_t *a, uint64_t nblocks, uint64_t mask)
{
  constexpr int  LOGN = 14;
  const uint32_t t    = threadIdx.x;
  uint64_t       b    = blockIdx.x;
  if(b >= nblocks) return;
  u64x2 raw[8];
  {
    const __amdgpu_buffer_rsrc_t r = block_rsrc<LOGN>(a + (b << LOGN));
#pragma unroll
    for(int h = 0; h < 8; h++) raw[h] = buffer_load_u64x2(r, t * 16u, (uint32_t)h * 16384u);
  }
  typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
  for(; b < nblocks; b += gridDim.x) {
    u64x2 x[8];
#pragma unroll
    for(int h = 0; h < 8; h++) x[h] = u64x2{raw[h].a ^ mask, raw[h].b ^ mask};
    const bool     more = b + gridDim.x < nblocks;
    const uint64_t nb   = more ? b + gridDim.x : b;
    {
      const __amdgpu_buffer_rsrc_t r = block_rsrc<LOGN>(a + (nb << LOGN), more);
#pragma unroll
      for(int h = 0; h < 8; h++) raw[h] = buffer_load_u64x2(r, t * 16u, (uint32_t)h * 16384u);
    }
    const __amdgpu_buffer_rsrc_t w = block_rsrc<LOGN>(a + (b << LOGN));
#pragma unroll
    for(int h = 0; h < 8; h++) {
      v4u32 v;
      v.x = (unsigned)x[h].a;
      v.y = (unsigned)(x[h].a >> 32);
      v.z = (unsigned)x[h].b;
      v.w = (unsigned)(x[h].b >> 32);
      __builtin_amdgcn_raw_buffer_store_b128(v, w, (int)(t * 16u), (int)((uint32_t)h * 16384u), 0);
    }
  }
}

extern "C" int ntt_shape_probe(int device, uint64_t *d_a, uint64_t n, uint64_t mask, void *stream)
{
  int rc = check_device(device);
  if(rc) return rc;
  if(!d_a || (n & ((1ull << 14) - 1)) || ((uintptr_t)d_a & 15)) return fail(NTT_ERR_ARG, "shape probe: null, unaligned or not a multiple of 2^14 words");
  if(n == 0) return NTT_OK;
  USE_DEVICE(device);
  int cus = 256;
  {
    hipDeviceProp_t prop;
    if(hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
  }
  const uint64_t nblocks = n >> 14;
  const unsigned grid    = (unsigned)(nblocks < (uint64_t)cus ? nblocks : (uint64_t)cus);
  hipLaunchKernelGGL(shape_probe_kernel, dim3(grid), dim3(1024), 0, (hipStream_t)stream, d_a, nblocks, mask);
  HIP_TRY(hipGetLastError());
  return NTT_OK;
}

/* ------------------------------------------------------------------ */
/* thin HIP wrappers                                                   */
/* ------------------------------------------------------------------ */
extern "C" const char *ntt_last_error(void) { return g_err.c_str(); }
extern "C" const char *ntt_version(void) { return "ntt_mi355x 0.1 (gfx950)"; }

extern "C" int ntt_device_count(void)
{
  int        n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if(e != hipSuccess) return fail(NTT_ERR_NO_DEVICE, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  return n;
}

#define DEV_PROLOG(device)          \
  {                                 \
    int rc_ = check_device(device); \
    if(rc_) return rc_;             \
  }                                 \
  USE_DEVICE(device)

extern "C" int ntt_dev_malloc(int device, void **d_ptr, size_t bytes)
{
  DEV_PROLOG(device);
  if(!d_ptr) return fail(NTT_ERR_ARG, "null argument");
  hipError_t e = hipMalloc(d_ptr, bytes);
  if(e == hipErrorOutOfMemory) return fail(NTT_ERR_NOMEM, "hipMalloc: out of device memory");
  HIP_TRY(e);
  return NTT_OK;
}
extern "C" int ntt_dev_mem_info(int device, size_t *free_bytes, size_t *total_bytes)
{
  DEV_PROLOG(device);
  if(!free_bytes || !total_bytes) return fail(NTT_ERR_ARG, "null argument");
  HIP_TRY(hipMemGetInfo(free_bytes, total_bytes));
  return NTT_OK;
}
extern "C" int ntt_dev_free(int device, void *d_ptr)
{
  DEV_PROLOG(device);
  HIP_TRY(hipFree(d_ptr));
  return NTT_OK;
}
extern "C" int ntt_h2d(int device, void *d_dst, const void *h_src, size_t bytes)
{
  DEV_PROLOG(device);
  HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return NTT_OK;
}
extern "C" int ntt_d2h(int device, void *h_dst, const void *d_src, size_t bytes)
{
  DEV_PROLOG(device);
  HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return NTT_OK;
}
extern "C" int ntt_stream_create(int device, void **stream)
{
  DEV_PROLOG(device);
  hipStream_t s;
  HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = (void *)s;
  return NTT_OK;
}
extern "C" int ntt_stream_destroy(int device, void *stream)
{
  DEV_PROLOG(device);
  HIP_TRY(hipStreamDestroy((hipStream_t)stream));
  return NTT_OK;
}
extern "C" int ntt_stream_sync(int device, void *stream)
{
  DEV_PROLOG(device);
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return NTT_OK;
}
extern "C" int ntt_event_create(int device, void **event)
{
  DEV_PROLOG(device);
  hipEvent_t ev;
  HIP_TRY(hipEventCreate(&ev));
  *event = (void *)ev;
  return NTT_OK;
}
extern "C" int ntt_event_destroy(int device, void *event)
{
  DEV_PROLOG(device);
  HIP_TRY(hipEventDestroy((hipEvent_t)event));
  return NTT_OK;
}
extern "C" int ntt_event_record(int device, void *event, void *stream)
{
  DEV_PROLOG(device);
  HIP_TRY(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return NTT_OK;
}
extern "C" int ntt_event_elapsed_ms(int device, void *start, void *stop, float *ms)
{
  DEV_PROLOG(device);
  HIP_TRY(hipEventSynchronize((hipEvent_t)stop));
  HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return NTT_OK;
}

/* ------------------------------------------------------------------ */
/* multi-GPU                                                           */
/* ------------------------------------------------------------------ */
static std::mutex g_own_stream_mu; /* lazy creation of ntt_plan::own_stream */

extern "C" int ntt_batch_multi(int ndev, ntt_plan *const *plans, uint64_t *const *d_a, const uint64_t *batch,
                               int inverse)
{
  if(ndev <= 0 || !plans || !d_a || !batch) return fail(NTT_ERR_ARG, "bad argument");
  for(int g = 0; g < ndev; g++) {
    if(!plans[g]) return fail(NTT_ERR_ARG, "null plan");
  }
  int launched = 0, rc = NTT_OK;
  for(int g = 0; g < ndev && !rc; g++) {
    ntt_plan *p = plans[g];
    {
      std::lock_guard<std::mutex> lock(g_own_stream_mu);
      if(!p->own_stream) {
        DeviceGuard guard(p->device);
        if(!guard.ok || hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking) != hipSuccess) {
          rc = fail(NTT_ERR_HIP, "ntt_batch_multi: stream creation failed");
          break;
        }
      }
    }
    rc = run_transform(p, d_a[g], batch[g], inverse != 0, false, (void *)p->own_stream);
    if(!rc) launched = g + 1;
  }
  /* join every stream that received work -- also when a later device failed: the caller must not get control back
   * while transforms it did not ask to abandon are still writing into its buffers */
  std::string first_error = rc ? g_err : std::string();
  for(int g = 0; g < launched; g++) {
    DeviceGuard guard(plans[g]->device);
    if(!guard.ok || hipStreamSynchronize(plans[g]->own_stream) != hipSuccess) {
      if(!rc) {
        rc          = NTT_ERR_HIP;
        first_error = "ntt_batch_multi: hipStreamSynchronize failed";
      }
    }
  }
  if(rc) return fail(rc, first_error);
  return NTT_OK;
}

/* RNS products across devices (BASELINE config 5 "on 8 GPUs" from one C call): shard g -- nlimbs plans on ONE device,
 * slabs [limb][batch[g]][N] resident there -- is enqueued on that device's own stream; the call returns when every shard is
 * done.  Like ntt_batch_multi: no collective, polynomials are independent. */
extern "C" int ntt_rns_mul_multi(int ndev, int nlimbs, ntt_plan *const *plans, uint64_t *const *d_c, uint64_t *const *d_a,
                                 uint64_t *const *d_b, const uint64_t *batch)
{
  if(ndev <= 0 || nlimbs <= 0 || !plans || !d_c || !d_a || !d_b || !batch) return fail(NTT_ERR_ARG, "bad argument");
  int launched = 0, rc = NTT_OK;
  for(int g = 0; g < ndev && !rc; g++) {
    ntt_plan *const *pg = plans + (size_t)g * nlimbs; /* plans[g * nlimbs + l]: limb l on device g */
    rc                  = rns_check(nlimbs, pg);
    if(rc) break;
    ntt_plan *p = pg[0];
    {
      std::lock_guard<std::mutex> lock(g_own_stream_mu);
      if(!p->own_stream) {
        DeviceGuard guard(p->device);
        if(!guard.ok || hipStreamCreateWithFlags(&p->own_stream, hipStreamNonBlocking) != hipSuccess) {
          rc = fail(NTT_ERR_HIP, "ntt_rns_mul_multi: stream creation failed");
          break;
        }
      }
    }
    rc = ntt_rns_negacyclic_mul_batch(nlimbs, pg, d_c[g], d_a[g], d_b[g], batch[g], (void *)p->own_stream);
    if(!rc) launched = g + 1;
  }
  std::string first_error = rc ? g_err : std::string();
  for(int g = 0; g < launched; g++) { /* join every stream that received work, also when a later device failed */
    ntt_plan *p = plans[(size_t)g * nlimbs];
    DeviceGuard guard(p->device);
    if(!guard.ok || hipStreamSynchronize(p->own_stream) != hipSuccess) {
      if(!rc) {
        rc          = NTT_ERR_HIP;
        first_error = "ntt_rns_mul_multi: hipStreamSynchronize failed";
      }
    }
  }
  if(rc) return fail(rc, first_error);
  return NTT_OK;
}

/* ------------------------------------------------------------------ */
/* parameter helpers (reference: tests/test_cases.h:113-142)           */
/* ------------------------------------------------------------------ */
static bool h_is_prime(uint64_t n)
{
  static const uint64_t wit[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if(n < 2) return false;
  for(uint64_t b : wit) {
    if(n % b == 0) return n == b;
  }
  uint64_t d = n - 1;
  int      r = 0;
  while(!(d & 1)) {
    d >>= 1;
    r++;
  }
  for(uint64_t b : wit) {
    uint64_t x = h_powmod(b, d, n);
    if(x == 1 || x == n - 1) continue;
    bool comp = true;
    for(int i = 1; i < r && comp; i++) {
      x = h_mulmod(x, x, n);
      if(x == n - 1) comp = false;
    }
    if(comp) return false;
  }
  return true;
}

extern "C" uint64_t ntt_min_root(uint64_t q, uint64_t N)
{
  if(!is_pow2(N) || q < 3 || (q - 1) % (2 * N) || !h_is_prime(q)) return 0;
  const uint64_t cof = (q - 1) / (2 * N);
  uint64_t       g   = 0;
  /* half of all residues qualify for a prime q; the bound only guards against misuse */
  for(uint64_t x = 2; x < q && x < 100000 && !g; x++) {
    const uint64_t c = h_powmod(x, cof, q);
    if(h_powmod(c, N, q) == q - 1) g = c;
  }
  if(!g) return 0;
  const uint64_t g2   = h_mulmod(g, g, q);
  uint64_t       best = g, cur = g;
  for(uint64_t i = 0; i < N; i++) {
    best = cur < best ? cur : best;
    cur  = h_mulmod(cur, g2, q);
  }
  return best;
}

extern "C" uint64_t ntt_find_prime(unsigned bits, uint64_t N, unsigned skip)
{
  if(bits < 4 || bits > 61 || !is_pow2(N)) return 0;
  const uint64_t step = 2 * N;
  for(uint64_t p = (((1ull << bits) - 1) / step) * step + 1; p > step; p -= step) {
    if(h_is_prime(p)) {
      if(!skip) return p;
      skip--;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* reference-signature entry points (host pointers, one polynomial)    */
/* ------------------------------------------------------------------ */
namespace {

/* One cached plan of the reference-signature entry points, with everything a call needs privately: its staging buffer,
 * its stream and its lock.  The reference's functions are re-entrant and may be called from several threads at once
 * (they only touch their arguments); here callers that use DIFFERENT tables run concurrently (each on its entry's
 * stream), callers that share a table queue on that entry's lock.  The cache itself is guarded by g_mu only while it
 * is searched or changed, never across a transform. */
struct CompatPlan {
  uint64_t  N = 0, q = 0, key = 0, ninv = 0, stride = 0;
  uint64_t  first[2] = {0, 0}, last = 0; /* table entries 0, 1 and the last one: compared besides the hash */
  const uint64_t *w_ptr = nullptr, *wcon_ptr = nullptr; /* where the caller's tables lay the last time this entry served: a call with
                                                         * the same pointers starts on this entry while its tables are still being hashed */
  bool      inverse = false;
  int       device = 0, arith = 0;
  ntt_plan *plan = nullptr;
  uint64_t  last_use = 0;
  std::mutex  mu;
  uint64_t *  stage       = nullptr;
  size_t      stage_bytes = 0;
  uint64_t *  hstage      = nullptr; /* pinned, device-mapped host buffer: single-pass transforms run on it in place (zero copy) */
  size_t      hstage_bytes = 0;
  hipStream_t stream      = nullptr;
  ~CompatPlan()
  {
    if(plan) {
      DeviceGuard guard(device);
      if(hstage) (void)hipHostFree(hstage);
      if(stage) (void)hipFree(stage);
      if(stream) (void)hipStreamDestroy(stream);
      ntt_plan_destroy(plan);
    }
  }
};

std::mutex                               g_mu;
std::vector<std::shared_ptr<CompatPlan>> g_plans;
uint64_t                                 g_use_clock = 0;
constexpr size_t                         kCompatPlansMax = 32; /* least-recently-used plan is dropped beyond this */

/* Identifies a caller table by ALL of its entries (position-sensitive rotate-xor-add hash over eight
 * independent lanes so the host compiler vectorises it: ~0.1 ms at N = 2^17, a fraction of the two PCIe
 * copies of the same call).  A sampled digest would hand a stale plan to a table that was edited in place
 * or that differs only in unsampled slots -- silently wrong results on a path whose only job is
 * correctness. */
uint64_t table_key(const uint64_t *w, uint64_t n, uint64_t stride)
{
  /* eight independent rotate-xor-add lanes (no multiplies in the loop: the host compiler vectorises it), folded with
   * multiplicative mixing at the end; every entry and its position influence the result */
  uint64_t h[8] = {0x9e3779b97f4a7c15ULL, 0xbf58476d1ce4e5b9ULL, 0x94d049bb133111ebULL, 0xcbf29ce484222325ULL,
                   0x2545f4914f6cdd1dULL, 0xd6e8feb86659fd93ULL, 0xa0761d6478bd642fULL, 0xe7037ed1a0b428dbULL};
  uint64_t i    = 0;
  if(stride == 1) {
    for(; i + 8 <= n; i += 8) {
      for(int l = 0; l < 8; l++) h[l] = (((h[l] << 7) | (h[l] >> 57)) ^ w[i + l]) + h[l];
    }
  }
  for(; i < n; i++) {
    uint64_t &x = h[i & 7];
    x           = (((x << 7) | (x >> 57)) ^ w[i * stride]) + x;
  }
  uint64_t r = n;
  for(int l = 0; l < 8; l++) {
    r = (r ^ h[l]) * 0xff51afd7ed558ccdULL;
    r ^= r >> 33;
  }
  return r;
}

[[noreturn]] void die(const char *fn)
{
  fprintf(stderr, "libntt_mi355x: %s failed: %s\n", fn, g_err.c_str());
  abort();
}

/* Which arithmetic serves the reference-signature entry points.
 *   default  the reference's own integer arithmetic on the caller's tables AND precomputations as they
 *            are: Harvey radix-2 butterflies for the *_ref_harvey / *_seal signatures, the radix-4
 *            butterflies on the 2N-entry expanded table for *_radix4 / *_radix4x4 (2^6 <= N <= 2^14; other
 *            sizes fall back to radix-2 on the table's even slots).  Even the LAZY outputs equal the
 *            reference's bit for bit.
 *   NTT_COMPAT_ARITH=f64  the FP64 engine where q allows it (outputs reduced: a legal lazy value) --
 *            lets the reference's own drivers exercise the throughput kernels. */
enum CompatKind { kCompatR2 = 0, kCompatR4 = 1 };

/* NTT_DEVICE / NTT_COMPAT_ARITH, read ONCE (the first reference-signature call of the process; a function-local static is
 * initialised exactly once, also with several threads arriving together): later setenv calls of the host program neither race with
 * the library nor change it */
struct CompatConfig {
  int  device;
  bool f64;
  bool zero_copy;
};
const CompatConfig &compat_config()
{
  static const CompatConfig cfg = [] {
    const char *d = getenv("NTT_DEVICE"), *a = getenv("NTT_COMPAT_ARITH"), *z = getenv("NTT_COMPAT_ZERO_COPY");
    return CompatConfig{d ? atoi(d) : 0, a && !strcmp(a, "f64"), !(z && !strcmp(z, "0"))};
  }();
  return cfg;
}

int compat_arith(uint64_t q, uint64_t N, CompatKind kind, bool inverse)
{
  if(compat_config().f64 && h_f64_eligible(q)) return NTT_ARITH_F64;
  const int m = (int)h_log2(N);
  /* the reference's radix-4 butterflies at every size that has them (two passes above 2^14) */
  (void)inverse;
  if(kind == kCompatR4 && m >= kFusedMin && m <= kRadix4Max && q < (1ull << 60)) return NTT_ARITH_U64_R4;
  return NTT_ARITH_U64;
}

/* fwd_ntt_radix4x4_lazy at log2 N = 4k+3 (ntt_core.h, r4x4_layer_*; src/ntt_radix4x4.c:53-111): 2k radix-4 layers, the
 * radix-2 stage on distance-4 pairs, the last radix-4 layer -- one launch each on the entry's stream */
__global__ void __launch_bounds__(256) r4x4_r4_layer_kernel(uint64_t *a, const TwU64 *e, uint64_t blocks, uint64_t span, ArithU64::consts c)
{
  const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if(id < blocks * span) r4x4_layer_r4(a, e, blocks, span, id, c);
}
__global__ void __launch_bounds__(256) r4x4_r2_layer_kernel(uint64_t *a, const TwU64 *e, uint64_t N, ArithU64::consts c)
{
  const uint64_t id = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if(id < N / 2) r4x4_layer_r2(a, e, N, id, c);
}
int run_r4x4_layers(const ntt_plan *p, uint64_t *d_a, hipStream_t st)
{
  const uint64_t N = p->N;
  const TwU64 *  e = (const TwU64 *)p->d_fwd;
  uint64_t       blocks = 1, span = N / 4;
  for(; blocks < (N >> 3); blocks *= 4, span /= 4) {
    hipLaunchKernelGGL(r4x4_r4_layer_kernel, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, st, d_a, e, blocks, span, p->cu);
  }
  hipLaunchKernelGGL(r4x4_r2_layer_kernel, dim3((unsigned)((N / 2 + 255) / 256)), dim3(256), 0, st, d_a, e, N, p->cu);
  hipLaunchKernelGGL(r4x4_r4_layer_kernel, dim3((unsigned)((N / 4 + 255) / 256)), dim3(256), 0, st, d_a, e, N / 4, (uint64_t)1, p->cu);
  HIP_TRY(hipGetLastError());
  return NTT_OK;
}

/* w, w_con: caller tables.  kind R2: N-entry radix-2 tables.  kind R4: 2N-entry expanded tables
 * (pre_compute.h:85-105), whose even slots are the radix-2 entries. */
void compat_run(const char *fn, uint64_t *a1, uint64_t *a2, uint64_t N, uint64_t q, const uint64_t *w,
                const uint64_t *w_con, CompatKind kind, bool inverse, uint64_t ninv, bool r4x4 = false)
{
  if(!is_pow2(N) || N < 2 || !a1 || !w) { /* (the table's slot 1 is read below: checked before anything is touched) */
    g_err = "N must be a power of two >= 2 and the pointers non-null";
    die(fn);
  }
  const int      device  = compat_config().device;
  const int      arith   = compat_arith(q, N, kind, inverse);
  const uint64_t entries = kind == kCompatR4 ? 2 * N : N;
  /* the integer policies use the caller's precomputation too: it is part of the key.  (Hashing happens outside any lock.) */
  const auto hash_tables = [&]() { return table_key(w, entries, 1) ^ (arith != NTT_ARITH_F64 && w_con ? table_key(w_con, entries, 1) * 3 : 0); };
  const auto cheap_match = [&](const CompatPlan &c) {
    return c.N == N && c.q == q && c.stride == (uint64_t)kind && c.inverse == inverse && c.ninv == ninv && c.device == device &&
           c.arith == arith && c.first[0] == w[0] && c.first[1] == w[1] && c.last == w[entries - 1];
  };
  const uint64_t batch = a2 ? 2 : 1;
  const size_t   bytes = (size_t)batch * N * sizeof(uint64_t);
  /* copies and kernels queue on the entry's stream; ONE synchronisation at the end (the _dbl form's second polynomial rides in
   * the same queue).  enqueue: staging buffer, H2D, the transform; finish: D2H and the synchronisation.  (The caller holds the
   * entry's lock and a DeviceGuard.) */
  /* One pass over the data (N <= 2^14, block kernels): the kernel reads the polynomial from a pinned, device-mapped host buffer and
   * writes it back there -- every word crosses PCIe once in each direction inside the kernel, no DMA submissions, no device
   * staging.  Transforms of several passes keep the device staging buffer (their intermediates must not cross PCIe). */
  const bool layered_ = r4x4 && !inverse && arith == NTT_ARITH_U64_R4 && (h_log2(N) & 3) == 3;
  const bool zero_copy = compat_config().zero_copy && !layered_ && h_log2(N) >= kFusedMin && h_log2(N) <= kFusedMax;
  const auto enqueue = [&](CompatPlan *ent) {
    if(!ent->stream && hipStreamCreateWithFlags(&ent->stream, hipStreamNonBlocking) != hipSuccess) {
      g_err = "hipStreamCreate";
      die(fn);
    }
    if(zero_copy) {
      if(bytes > ent->hstage_bytes) {
        if(ent->hstage) (void)hipHostFree(ent->hstage);
        ent->hstage       = nullptr;
        ent->hstage_bytes = 0;
        if(hipHostMalloc((void **)&ent->hstage, bytes, hipHostMallocMapped) != hipSuccess) {
          g_err = "hipHostMalloc staging buffer";
          die(fn);
        }
        ent->hstage_bytes = bytes;
      }
      memcpy(ent->hstage, a1, N * 8);
      if(a2) memcpy(ent->hstage + N, a2, N * 8);
      void *dev = nullptr;
      if(hipHostGetDevicePointer(&dev, ent->hstage, 0) != hipSuccess) {
        g_err = "hipHostGetDevicePointer";
        die(fn);
      }
      if(run_transform(ent->plan, (uint64_t *)dev, batch, inverse, true, (void *)ent->stream, !inverse)) die(fn);
      return;
    }
    if(bytes > ent->stage_bytes) {
      if(ent->stage) (void)hipFree(ent->stage);
      ent->stage       = nullptr;
      ent->stage_bytes = 0;
      if(hipMalloc((void **)&ent->stage, bytes) != hipSuccess) {
        g_err = "hipMalloc staging buffer";
        die(fn);
      }
      ent->stage_bytes = bytes;
    }
    uint64_t *const   stage = ent->stage;
    const hipStream_t st    = ent->stream;
    bool ok = hipMemcpyAsync(stage, a1, N * 8, hipMemcpyHostToDevice, st) == hipSuccess;
    if(ok && a2) ok = hipMemcpyAsync(stage + N, a2, N * 8, hipMemcpyHostToDevice, st) == hipSuccess;
    if(!ok) {
      g_err = "hipMemcpy H2D";
      die(fn);
    }
    /* lazy inputs accepted, lazy outputs returned: the *_lazy contract (include/ntt_reference.h:13-17) */
    /* (the radix-4x4 formulation has lazy words of its own only when log2 N = 4k+3: see run_r4x4_layers) */
    const bool layered = r4x4 && !inverse && arith == NTT_ARITH_U64_R4 && (h_log2(N) & 3) == 3;
    if(layered ? run_r4x4_layers(ent->plan, stage, st) : run_transform(ent->plan, stage, batch, inverse, true, (void *)st, !inverse)) die(fn);
  };
  const auto finish = [&](CompatPlan *ent) {
    if(zero_copy) {
      if(hipStreamSynchronize(ent->stream) != hipSuccess) {
        g_err = std::string("kernel execution: ") + hipGetErrorString(hipGetLastError());
        die(fn);
      }
      memcpy(a1, ent->hstage, N * 8);
      if(a2) memcpy(a2, ent->hstage + N, N * 8);
      return;
    }
    bool ok = hipMemcpyAsync(a1, ent->stage, N * 8, hipMemcpyDeviceToHost, ent->stream) == hipSuccess;
    if(ok && a2) ok = hipMemcpyAsync(a2, ent->stage + N, N * 8, hipMemcpyDeviceToHost, ent->stream) == hipSuccess;
    if(ok) ok = hipStreamSynchronize(ent->stream) == hipSuccess;
    if(!ok) {
      g_err = std::string("hipMemcpy D2H / kernel execution: ") + hipGetErrorString(hipGetLastError());
      die(fn);
    }
  };
  DeviceGuard guard(device);
  if(!guard.ok) {
    g_err = "hipSetDevice";
    die(fn);
  }
  /* Speculation: a call that hands over the SAME table pointers as an entry served before (and agrees with it in every parameter
   * and in the three sampled entries) starts on that entry at once -- upload and transform are queued -- and hashes ALL table
   * entries meanwhile (256-512 KiB at 2^14: as long as the transform itself).  Only when the full hash confirms the entry is the
   * result copied back; if the caller edited the table in place, the speculative work is drained and dropped (the caller's
   * polynomial has not been touched) and the call proceeds as a miss.  Nothing is ever served on a sampled digest alone. */
  uint64_t key      = 0;
  bool     have_key = false;
  {
    std::shared_ptr<CompatPlan> spec;
    {
      std::lock_guard<std::mutex> lock(g_mu);
      for(const std::shared_ptr<CompatPlan> &c : g_plans) {
        if(c->w_ptr == w && c->wcon_ptr == w_con && cheap_match(*c)) spec = c;
      }
    }
    if(spec) {
      std::lock_guard<std::mutex> run_lock(spec->mu);
      enqueue(spec.get());
      key      = hash_tables();
      have_key = true;
      if(key == spec->key) {
        finish(spec.get());
        std::lock_guard<std::mutex> lock(g_mu);
        spec->last_use = ++g_use_clock;
        return;
      }
      if(hipStreamSynchronize(spec->stream) != hipSuccess) { /* the tables changed under the same pointers: forget the result */
        g_err = "hipStreamSynchronize";
        die(fn);
      }
    }
  }
  if(!have_key) key = hash_tables();
  std::shared_ptr<CompatPlan> ent;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    for(const std::shared_ptr<CompatPlan> &c : g_plans) {
      /* the 64-bit hash AND the parameters AND three entries of the table itself: a hash collision alone cannot hand a
       * caller somebody else's tables */
      if(c->key == key && cheap_match(*c)) {
        ent         = c;
        c->last_use = ++g_use_clock;
      }
    }
  }
  if(!ent) {
    /* built outside the cache lock (table conversion + upload take a while); two threads that miss on the same table
     * both build, the second insert wins the lookup from then on and the first entry ages out */
    TableSet              ts;
    std::vector<uint64_t> tab, con;
    if(arith == NTT_ARITH_U64_R4) {
      tab.assign(w, w + 2 * N);
      if(w_con) con.assign(w_con, w_con + 2 * N);
      (inverse ? ts.einv : ts.efwd)         = tab;
      (inverse ? ts.einv_con : ts.efwd_con) = con;
    } else {
      const uint64_t stride = kind == kCompatR4 ? 2 : 1;
      tab.resize(N);
      for(uint64_t k = 0; k < N; k++) tab[k] = w[k * stride];
      if(w_con && arith == NTT_ARITH_U64) {
        con.resize(N);
        for(uint64_t k = 0; k < N; k++) con[k] = w_con[k * stride];
      }
      (inverse ? ts.inv : ts.fwd)         = tab;
      (inverse ? ts.inv_con : ts.fwd_con) = con;
    }
    ntt_plan *plan = nullptr;
    if(plan_build(&plan, device, N, q, 0, ts, arith, inverse ? ninv : 0)) die(fn);
    ent           = std::make_shared<CompatPlan>();
    ent->N        = N;
    ent->q        = q;
    ent->key      = key;
    ent->ninv     = ninv;
    ent->stride   = (uint64_t)kind;
    ent->first[0] = w[0];
    ent->first[1] = w[1];
    ent->last     = w[entries - 1];
    ent->inverse  = inverse;
    ent->device   = device;
    ent->arith    = arith;
    ent->plan     = plan;
    std::lock_guard<std::mutex> lock(g_mu);
    ent->last_use = ++g_use_clock;
    if(g_plans.size() >= kCompatPlansMax) {
      size_t lru = 0;
      for(size_t k = 1; k < g_plans.size(); k++) {
        if(g_plans[k]->last_use < g_plans[lru]->last_use) lru = k;
      }
      g_plans.erase(g_plans.begin() + (long)lru); /* destroyed when its last user lets go of it */
    }
    g_plans.push_back(ent);
  }
  std::lock_guard<std::mutex> run_lock(ent->mu);
  {
    std::lock_guard<std::mutex> lock(g_mu); /* (read under g_mu by the speculative lookup; lock order entry -> cache everywhere) */
    ent->w_ptr    = w;                      /* the next call with these pointers speculates on this entry */
    ent->wcon_ptr = w_con;
  }
  enqueue(ent.get());
  finish(ent.get());
}

} // namespace

/* frees what the reference-signature entry points keep between calls (cached plans with their device
 * tables, the staging buffer).  Safe to call at any time; the next call rebuilds what it needs. */
extern "C" void ntt_compat_release(void)
{
  std::vector<std::shared_ptr<CompatPlan>> drop;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    drop.swap(g_plans);
  }
  drop.clear(); /* entries still in use by another thread are destroyed when that call returns */
}
extern "C" int ntt_compat_cached_plans(void)
{
  std::lock_guard<std::mutex> lock(g_mu);
  return (int)g_plans.size();
}

extern "C" {

void fwd_ntt_ref_harvey_lazy(uint64_t a[], uint64_t N, uint64_t q, const uint64_t w[], const uint64_t w_con[])
{
  compat_run("fwd_ntt_ref_harvey_lazy", a, nullptr, N, q, w, w_con, kCompatR2, false, 0);
}

void fwd_ntt_ref_harvey_lazy_dbl(uint64_t a1[], uint64_t a2[], uint64_t N, uint64_t q, const uint64_t w[],
                                 const uint64_t w_con[])
{
  compat_run("fwd_ntt_ref_harvey_lazy_dbl", a1, a2, N, q, w, w_con, kCompatR2, false, 0);
}

void inv_ntt_ref_harvey(uint64_t a[], uint64_t N, uint64_t q, mul_op_t n_inv, uint64_t word_size,
                        const uint64_t w[], const uint64_t w_con[])
{
  (void)word_size; /* 64 for every in-scope caller (tests/test_cases.h:231,244) */
  compat_run("inv_ntt_ref_harvey", a, nullptr, N, q, w, w_con, kCompatR2, true, (uint64_t)n_inv.op);
}

void fwd_ntt_radix4_lazy(uint64_t a[], uint64_t N, uint64_t q, const uint64_t w[], const uint64_t w_con[])
{
  compat_run("fwd_ntt_radix4_lazy", a, nullptr, N, q, w, w_con, kCompatR4, false, 0);
}

void inv_ntt_radix4(uint64_t a[], uint64_t N, uint64_t q, mul_op_t n_inv, const uint64_t w[],
                    const uint64_t w_con[])
{
  compat_run("inv_ntt_radix4", a, nullptr, N, q, w, w_con, kCompatR4, true, (uint64_t)n_inv.op);
}

/* the radix-16 blocking of the reference (src/ntt_radix4x4.c:41-114) is a cache-friendlier order of fwd_ntt_radix4_lazy's
 * butterflies -- on the device the register-resident stage groups play that role, and the values are the same -- except
 * for its remainder handling when log2 N = 4k+3, which leaves other lazy words: those sizes run layer by layer
 * (run_r4x4_layers), so that this symbol too returns the reference's words bit for bit at every size from 2^6 */
void fwd_ntt_radix4x4_lazy(uint64_t a[], uint64_t N, uint64_t q, const uint64_t w[], const uint64_t w_con[])
{
  compat_run("fwd_ntt_radix4x4_lazy", a, nullptr, N, q, w, w_con, kCompatR4, false, 0, true);
}

void fwd_ntt_seal_lazy(uint64_t a[], uint64_t N, uint64_t q, const uint64_t w[], const uint64_t w_con[])
{
  compat_run("fwd_ntt_seal_lazy", a, nullptr, N, q, w, w_con, kCompatR2, false, 0);
}

void inv_ntt_seal(uint64_t a[], uint64_t N, uint64_t q, uint64_t n_inv, uint64_t n_inv_con, const uint64_t w[],
                  const uint64_t w_con[])
{
  (void)n_inv_con;
  compat_run("inv_ntt_seal", a, nullptr, N, q, w, w_con, kCompatR2, true, n_inv);
}

} /* extern "C" */
