/*
 * ntt_kernels.h -- gfx950 kernels built from the templates of ntt_core.h.
 *
 * fused_kernel : one workgroup transforms whole 2^LOGN blocks.  16 coefficients
 *                per thread live in VGPRs for up to four stages at a time; the
 *                block crosses LDS once per stage group and HBM exactly twice
 *                (one coalesced read, one coalesced write): 16 bytes of HBM
 *                traffic per coefficient per transform, the algorithmic minimum
 *                (SURVEY 8d).  No MFMA: 53/64-bit modular butterflies are
 *                element-wise VALU work.
 * column_kernel: strided passes for the leading stages of N > 2^14 and for tiny N.
 *
 * Launch geometry (wave64, 256 CUs): LOGN=14 -> 1024 threads (16 waves, 4 per
 * SIMD, <=128 VGPRs) and ~128 KiB of the CU's 160 KiB LDS; smaller blocks use
 * 256..512-thread workgroups so that several are resident per CU and one
 * group's HBM latency hides under another's butterflies.
 */
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_core.h"

namespace ntt {

template <int LOGN> struct Geom {
  using P                  = Plan<LOGN>;
  static constexpr int WG  = P::T < 256 ? 256 : P::T; /* threads per workgroup    */
  static constexpr int BPW = WG / P::T;               /* blocks per workgroup     */
  /* waves per SIMD the register allocator may assume (VGPR budget 512/x) */
  static constexpr int WPS = 4;
};

__device__ __forceinline__ void wave_sync()
{
  /* LDS operations of one wave execute in issue order; only the compiler has to
   * be told not to move accesses across the exchange */
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class A, int LOGN, int GW, int GR>
__device__ __forceinline__ void exchange(typename A::val (&x)[kE], uint32_t t, typename A::val *lds)
{
  using P = Plan<LOGN>;
#ifdef NTT_SAFE_BARRIERS
  constexpr bool local = false;
#else
  constexpr bool local = P::WAVE_LOCAL(GW, GR);
#endif
  if constexpr(local) {
    lds_scatter<A, LOGN, GW, GR>(x, t, lds);
    wave_sync();
    lds_gather<A, LOGN, GW, GR>(x, t, lds);
    wave_sync();
  } else {
    __syncthreads(); /* every wave has finished reading the previous layout */
    lds_scatter<A, LOGN, GW, GR>(x, t, lds);
    __syncthreads();
    lds_gather<A, LOGN, GW, GR>(x, t, lds);
  }
}

template <class A, int LOGN, bool INV, int KSH>
__global__ void __launch_bounds__(Geom<LOGN>::WG, Geom<LOGN>::WPS) fused_kernel(const Params<A> p)
{
  using P                 = Plan<LOGN>;
  using G                 = Geom<LOGN>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, INV, KSH>();
  __shared__ typename A::val lds_all[G::BPW * P::LDS_ELEMS];

  const uint32_t     tid = threadIdx.x;
  const uint32_t     sub = tid >> P::LT;
  const uint32_t     t   = tid & (P::T - 1);
  typename A::val *  lds = lds_all + sub * P::LDS_ELEMS;
  const uint32_t     bmask = (1u << p.s0) - 1u;

  for(uint64_t b0 = (uint64_t)blockIdx.x * G::BPW; b0 < p.nblocks; b0 += (uint64_t)gridDim.x * G::BPW) {
    uint64_t   b    = b0 + sub;
    const bool live = b < p.nblocks;
    if(!live) b = p.nblocks - 1; /* idle lanes shadow a real block, never store */
    const uint32_t blk  = (uint32_t)b & bmask;
    uint64_t *     base = p.a + (b << LOGN);
    typename A::val x[kE];
    if constexpr(!INV) {
      global_load_first<A, LOGN, false>(x, t, base, p.wide != 0, p.c);
      run_group<A, LOGN, 0, false, MASK>(x, t, blk, p);
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        exchange<A, LOGN, GI, GI + 1>(x, t, lds);
        run_group<A, LOGN, GI + 1, false, MASK>(x, t, blk, p);
      });
      if(live) global_store_last<A, LOGN, false>(x, t, base, p.c);
    } else {
      global_load_last<A, LOGN, true>(x, t, base, p.wide != 0, p.c);
      run_group<A, LOGN, P::NG - 1, true, MASK>(x, t, blk, p);
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = P::NG - 1 - decltype(gg)::value;
        exchange<A, LOGN, GI, GI - 1>(x, t, lds);
        run_group<A, LOGN, GI - 1, true, MASK>(x, t, blk, p);
      });
      if(live) global_store_first<A, LOGN, true>(x, t, base, p.c);
    }
  }
}

template <class A, int R, bool INV, int KSH>
__global__ void __launch_bounds__(256) column_kernel(uint64_t *a, uint64_t batch, uint32_t logn, uint32_t S,
                                                     uint32_t wide, uint32_t lastinv,
                                                     const typename A::tw *tab, const typename A::consts c)
{
  constexpr uint32_t MASK  = column_mask<A, R, INV, KSH>();
  const uint32_t     lcols = logn - R;
  const uint64_t     total = batch << lcols;
  for(uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total;
      g += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t poly = g >> lcols;
    const uint32_t col  = (uint32_t)(g & ((1ull << lcols) - 1));
    column_pass_thread<A, R, INV, MASK>(a + (poly << logn), col, logn, S, wide != 0, lastinv != 0, tab, c);
  }
}

/* ------------------------------------------------------------------ */
/* type-erased launch interface (one translation unit per policy/class) */
/* ------------------------------------------------------------------ */
struct PassArgs {
  uint64_t *  a;
  const void *tw;     /* device table of A::tw             */
  const void *consts; /* host pointer to A::consts         */
  uint64_t    batch;
  uint32_t    logn;   /* whole transform                   */
  int         fused;  /* Pass::fused                       */
  int         r;      /* Pass::r                           */
  int         s;      /* Pass::s                           */
  int         inverse;
  int         wide;
  int         lastinv;
  int         max_grid; /* cap on workgroups (0 = default) */
  hipStream_t stream;
};

template <class A, int KSH> hipError_t launch_pass(const PassArgs &pa);

template <class A, int LOGN, bool INV, int KSH> hipError_t launch_fused(const PassArgs &pa)
{
  using G = Geom<LOGN>;
  Params<A> p{};
  p.a       = pa.a;
  p.tw      = static_cast<const typename A::tw *>(pa.tw);
  p.c       = *static_cast<const typename A::consts *>(pa.consts);
  p.logn    = pa.logn;
  p.s0      = (uint32_t)pa.s;
  p.wide    = (uint32_t)pa.wide;
  p.lastinv = (uint32_t)pa.lastinv;
  p.nblocks = pa.batch << pa.s;
  uint64_t wgs = (p.nblocks + G::BPW - 1) / G::BPW;
  const uint64_t cap = pa.max_grid > 0 ? (uint64_t)pa.max_grid : (1ull << 20);
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  hipLaunchKernelGGL((fused_kernel<A, LOGN, INV, KSH>), dim3((unsigned)wgs), dim3(G::WG), 0, pa.stream, p);
  return hipGetLastError();
}

template <class A, int R, bool INV, int KSH> hipError_t launch_column(const PassArgs &pa)
{
  const uint64_t total = pa.batch << (pa.logn - R);
  uint64_t       wgs   = (total + 255) / 256;
  const uint64_t cap   = pa.max_grid > 0 ? (uint64_t)pa.max_grid : (1ull << 22);
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  hipLaunchKernelGGL((column_kernel<A, R, INV, KSH>), dim3((unsigned)wgs), dim3(256), 0, pa.stream, pa.a,
                     pa.batch, pa.logn, (uint32_t)pa.s, (uint32_t)pa.wide, (uint32_t)pa.lastinv,
                     static_cast<const typename A::tw *>(pa.tw),
                     *static_cast<const typename A::consts *>(pa.consts));
  return hipGetLastError();
}

/* body of launch_pass<A,KSH>; each instantiating .hip file expands this once */
#define NTT_DEFINE_LAUNCH_PASS(A, KSH)                                                   \
  template <> hipError_t launch_pass<A, KSH>(const PassArgs &pa)                         \
  {                                                                                      \
    if(pa.fused) {                                                                       \
      switch(pa.r) {                                                                     \
        NTT_FUSED_CASES(A, KSH)                                                          \
        default: return hipErrorInvalidValue;                                            \
      }                                                                                  \
    }                                                                                    \
    switch(pa.r) {                                                                       \
      case 1: return pa.inverse ? launch_column<A, 1, true, KSH>(pa) : launch_column<A, 1, false, KSH>(pa); \
      case 2: return pa.inverse ? launch_column<A, 2, true, KSH>(pa) : launch_column<A, 2, false, KSH>(pa); \
      case 3: return pa.inverse ? launch_column<A, 3, true, KSH>(pa) : launch_column<A, 3, false, KSH>(pa); \
      case 4: return pa.inverse ? launch_column<A, 4, true, KSH>(pa) : launch_column<A, 4, false, KSH>(pa); \
      default: return hipErrorInvalidValue;                                              \
    }                                                                                    \
  }

#define NTT_FUSED_CASE(A, KSH, LN) \
  case LN: return pa.inverse ? launch_fused<A, LN, true, KSH>(pa) : launch_fused<A, LN, false, KSH>(pa);

#define NTT_FUSED_CASES(A, KSH)                                                        \
  NTT_FUSED_CASE(A, KSH, 6) NTT_FUSED_CASE(A, KSH, 7) NTT_FUSED_CASE(A, KSH, 8)        \
  NTT_FUSED_CASE(A, KSH, 9) NTT_FUSED_CASE(A, KSH, 10) NTT_FUSED_CASE(A, KSH, 11)      \
  NTT_FUSED_CASE(A, KSH, 12) NTT_FUSED_CASE(A, KSH, 13) NTT_FUSED_CASE(A, KSH, 14)

} /* namespace ntt */
