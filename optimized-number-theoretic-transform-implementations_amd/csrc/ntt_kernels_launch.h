/*
 * ntt_kernels_launch.h -- column_kernel (strided passes) and the type-erased launch interface: PassArgs / ProdArgs / DotArgs / MulArgs, the launchers that choose
 * grids and kernel variants, and the NTT_DEFINE_LAUNCH_* macros the inst_*.hip translation units expand.
 * Part of ntt_kernels.h (included from there, in this order: block, team, products, launch); not a header of its own.
 */
#pragma once

namespace ntt {

template <class A, int R, bool INV, int KSH, bool MULTI = false>
__global__ void __launch_bounds__(256) column_kernel(const KArgs<A> k)
{
  /* k.nblocks = polynomials per limb, k.s0 = first global stage of the pass */
  uint32_t           bid, gdim, limb_;
  const Params<A>    p     = limb_params<A, INV, MULTI>(k, bid, gdim, limb_);
  constexpr uint32_t MASK  = column_mask<A, R, INV, KSH>();
  const uint32_t     lcols = p.logn - R;
  const uint64_t     total = p.nblocks << lcols;
  for(uint64_t g = (uint64_t)bid * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gdim * blockDim.x) {
    const uint64_t poly = g >> lcols;
    const uint32_t col  = (uint32_t)(g & ((1ull << lcols) - 1));
    if constexpr(A::kRadix4) {
      /* (even stage count: launch_pass refuses anything else for this policy) */
      if constexpr(R % 2 == 0) column_pass_thread_r4<A, R, INV>(p.a + poly_offset<false>(poly, p.pstride, p.ptab), col, p.logn, p.s0, p.tw, p.c, p.lazy != 0);
    } else {
      column_pass_thread<A, R, INV, MASK>(p.a + poly_offset<false>(poly, p.pstride, p.ptab), col, p.logn, p.s0, p.wide != 0, p.lastinv != 0, p.tw, p.c, p.lazy != 0);
    }
  }
}

/* ------------------------------------------------------------------ */
/* type-erased launch interface (one translation unit per policy/class) */
/* ------------------------------------------------------------------ */
struct PassArgs {
  uint64_t *  a;
  const void *limbs;       /* HOST array of LimbRec<A>, one per limb (copied into the kernel arguments) */
  int         nlimbs;      /* >= 1 */
  uint64_t    limb_stride; /* words between consecutive limbs' slabs   */
  uint64_t    poly_stride; /* words between consecutive polynomials of a limb (0 = dense: N) */
  const uint64_t *ptab;    /* pointer batch: DEVICE table of per-polynomial word offsets (a = null: entries are addresses / 8), `batch` entries;
                            * null = the progression above */
  uint64_t    batch;       /* polynomials per limb                     */
  uint32_t    logn;   /* whole transform                   */
  int         fused;  /* Pass::fused; 2 = both passes of a 2^16 / 2^17 transform in one workgroup (r = m - 14); 3 = both passes as
                       * items of one launch with the intermediate kept in the XCD's L2 (team_kernel, r = m - 12); 4 = a 2^15-point
                       * transform in one pass, the polynomial in the registers of one workgroup (onepass_kernel) */
  int         r;      /* Pass::r                           */
  int         s;      /* Pass::s                           */
  int         inverse;
  int         wide;
  int         lastinv;
  int         lazy;     /* the caller asked for lazy outputs of the whole transform */
  int         ends;     /* this pass is the last one of the transform */
  int         max_grid; /* cap on workgroups (0 = default) */
  int         num_cus;  /* compute units of the device     */
  int         oversub;  /* persistent block kernels: workgroups per resident slot (0 = block_oversub's default) */
  void *      team_ctl; /* fused == 3: device memory for the queues and counters (TeamCtl + batch counters) */
  int         team_lag, team_wpc;
  hipStream_t stream;
};

template <class A, int KSH> hipError_t launch_pass(const PassArgs &pa);

/* fused product (fused_product_kernel): c = inv(fwd(b) * ahat), whole polynomials of 2^14 points */
struct ProdArgs {
  uint64_t *      b;
  const uint64_t *ahat;
  uint64_t *      out;
  const void *    limbs;       /* HOST array of LimbRec<A> */
  int             nlimbs;
  uint64_t        limb_stride;
  uint64_t        poly_stride; /* words between consecutive polynomials of a limb, the same for all three operands (0 = dense: N) */
  uint64_t        batch;       /* per limb */
  uint32_t        logn;
  uint32_t        block_log; /* N > 2^14: log2 of the blocks (12, 13 or 14); the column passes around the launch cover logn - block_log stages */
  int             a_lazy;
  int             max_grid, num_cus;
  int             oversub;  /* as PassArgs::oversub */
  void *          team_ctl; /* launch_team_product: device memory for the queues and 2 * batch counters */
  int             team_lag, team_wpc;
  int             four; /* launch_team_product: ahat holds a's COEFFICIENTS; the launch transforms both operands */
  int             both; /* launch_product, N <= 2^14: the same for the fused product kernels */
  int             ptrs; /* launch_product, N <= 2^14, both: b, ahat and out are DEVICE TABLES of polynomial addresses (the PTRS kernels); nlimbs > 1: every
                             * polynomial's limbs limb_stride words apart behind its entry (launch_team_product: one limb) */
  uint64_t        ptr_limb_off; /* ptrs: words from every table entry to the (first) limb of this launch */
  hipStream_t     stream;
};
template <class A, int KSH> hipError_t launch_product(const ProdArgs &pa);
template <class A, int KSH> hipError_t launch_team_product(const ProdArgs &pa);

/* c = inverse block pass of sum_i a_i^ (.) b_i^ (dot_inv_kernel) */
struct DotArgs {
  uint64_t *             out;
  const uint64_t *const *a; /* HOST arrays of npairs device pointers (limb 0's slabs) */
  const uint64_t *const *b;
  int                    npairs;
  int                    lazy_in, b_bcast;
  const void *           limbs; /* HOST array of LimbRec<A> */
  int                    nlimbs;
  uint64_t               limb_stride, b_limb_stride;
  uint64_t               poly_stride; /* words between consecutive polynomials of a limb: every a_i^, c, and every b_i^ that is not broadcast (0 = dense: N) */
  uint64_t               batch;     /* per limb */
  uint32_t               logn;
  uint32_t               block_log; /* N > 2^14: log2 of the blocks (12 or 14); the inverse column passes follow as launches of their own */
  int                    max_grid, num_cus;
  int                    oversub; /* as PassArgs::oversub */
  void *                 team_ctl; /* N = 2^15..2^17: non-null = both passes as items of ONE launch (team_dot_kernel); TeamCtl + nlimbs * batch counters */
  int                    team_lag, team_wpc;
  int                    ptrs; /* out, every a[i] and every b[i] that is not broadcast are DEVICE TABLES of polynomial addresses; N <= 2^14 with nlimbs > 1: every
                                    * polynomial's limbs limb_stride words apart behind its entry (team_dot_kernel: one limb) */
  uint64_t               ptr_limb_off; /* ptrs: words from every table entry to the (first) limb of this launch */
  hipStream_t            stream;
};
template <class A, int KSH> hipError_t launch_dot(const DotArgs &da);

/* c^ = forward block pass of a, times b^ (+ c^) (fwd_mul_kernel) */
struct MulArgs {
  uint64_t *      a;   /* coefficients (N > 2^14: after the forward column passes) */
  const uint64_t *b;   /* b^ */
  uint64_t *      out; /* c^ */
  int             lazy_in, b_bcast, accumulate;
  const void *    limbs; /* HOST array of LimbRec<A> */
  int             nlimbs;
  uint64_t        limb_stride, b_limb_stride;
  uint64_t        poly_stride; /* words between consecutive polynomials of a limb: a, c^, and b^ unless broadcast (0 = dense: N) */
  uint64_t        batch;
  uint32_t        logn;
  uint32_t        block_log; /* N > 2^14: log2 of the blocks (12 or 14) */
  int             max_grid, num_cus;
  int             oversub; /* as PassArgs::oversub */
  void *          team_ctl; /* N = 2^15..2^17: non-null = column items and row items with the product as ONE launch (team_mul_kernel); a = the caller's coefficients */
  int             team_lag, team_wpc;
  int             one_pass; /* N = 2^15, FP64 policies: the transform in one pass with the product at its output (onepass_mul_kernel) */
  int             ptrs; /* a, out and b (unless broadcast) are DEVICE TABLES of polynomial addresses; N <= 2^14 with nlimbs > 1: every polynomial's limbs
                             * limb_stride words apart behind its entry (onepass_mul_kernel, team_mul_kernel: one limb) */
  uint64_t        ptr_limb_off; /* ptrs: words from every table entry to the (first) limb of this launch */
  hipStream_t     stream;
};
template <class A, int KSH> hipError_t launch_fwd_mul(const MulArgs &ma);

/* What a pass stores: the last pass of a transform honours the caller's lazy flag; every earlier pass of an
 * integer policy keeps the reference's lazy ranges in HBM (no reduction between stages, as in
 * src/ntt_reference.c:17-30 -- which also makes the final lazy values the reference's bit for bit).  The FP64
 * policy ignores the run-time flag (its passes exchange canonical words). */
inline int pass_lazy(const PassArgs &pa) { return pa.ends ? pa.lazy : 1; }

/* the MULTI kernel variants exist for the scheduled FP64 policy, its 52-bit form and the wide integer policy (RNS bases of
 * 54..60-bit primes: a ciphertext is a few polynomials x tens of such limbs -- one launch instead of one chain per prime) */
template <class A> constexpr bool multi_limb_built() { return A::kCompact || A::kIntWide; }

/* Workgroups launched per resident slot of a persistent block kernel.  One workgroup per slot (the r01..r04 grids) lets the four
 * 256-thread workgroups that share a CU at 2^12 run IN PHASE for the whole launch: they start together, do identical work and
 * meet at the memory system, the LDS pipe and their barriers at the same time.  With several times as many workgroups as
 * slots a slot is refilled whenever its workgroup runs out of blocks, at a time of its own, and the phases of a CU's workgroups
 * decorrelate: measured (profiles/r05/grid_sweep.txt, three alternating repetitions on one box) 2^12 forward 0.592 -> 0.622 of
 * the roofline at 8 workgroups per slot, inverse 0.617 -> 0.641, flat from 8 to 16, 0.61 with one block per workgroup (no
 * prefetch across blocks left); the 1024-thread kernels (2^13, 2^14: one workgroup per CU, 16 waves in step by construction)
 * measured no gain (0.591 at 1, 2, 4 per slot, 0.586 at 8) and keep one.  requested > 0 (NTT_OPT_BLOCK_OVERSUB) overrides. */
template <int LOGN, int WG> constexpr int block_oversub_default(bool whole_polynomials)
{
  return (LOGN == 12 && WG == 256 && whole_polynomials) ? 8 : 1;
}
template <int LOGN, int WG> inline uint64_t block_oversub(int requested, bool whole_polynomials)
{
  return (uint64_t)(requested > 0 ? requested : block_oversub_default<LOGN, WG>(whole_polynomials));
}

template <class A> KArgs<A> make_kargs(const PassArgs &pa)
{
  KArgs<A> k{};
  k.a            = pa.a;
  const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(pa.limbs);
  for(int l = 0; l < (pa.nlimbs > 0 ? pa.nlimbs : 1) && l < kMaxLimbs; l++) k.limbs[l] = recs[l];
  k.limb_stride  = pa.limb_stride;
  k.poly_stride  = pa.poly_stride ? pa.poly_stride : (1ull << pa.logn);
  k.wgs_per_limb = 1;
  k.logn         = pa.logn;
  k.s0           = 0;
  k.wide         = (uint32_t)pa.wide;
  k.lastinv      = (uint32_t)pa.lastinv;
  k.lazy         = (uint32_t)pa.lazy;
  k.nblocks      = pa.batch;
  k.ptab         = pa.ptab;
  return k;
}

template <class A, int LOGN, bool INV, int KSH> hipError_t launch_fused(const PassArgs &pa)
{
  using G = Geom<LOGN, INV, flavor_of<A>()>;
  KArgs<A> p  = make_kargs<A>(pa);
  p.s0        = (uint32_t)pa.s;
  p.lazy      = (uint32_t)pass_lazy(pa);
  p.nblocks   = pa.batch << pa.s;
  const uint64_t nl = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
  uint64_t wgs = (p.nblocks + G::BPW - 1) / G::BPW;
  uint64_t cap = 1ull << 20;
  if(G::PERSISTENT) {
    /* persistent prefetching loop: exactly the resident workgroups (LDS- and
     * wave-limited), each striding over the blocks */
    constexpr int by_lds    = G::WG_PER_CU0;
    constexpr int by_waves  = (G::WPS * 4 * 64) / G::WG;
    constexpr int per_cu    = by_lds < by_waves ? by_lds : by_waves;
    cap                     = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * block_oversub<LOGN, G::WG>(pa.oversub, pa.s == 0);
  }
  if(!G::PERSISTENT && G::LDS_TW > 0) {
    /* tables are filled once per workgroup: a few workgroups per resident slot, each looping */
    if(pa.s != 0) return hipErrorInvalidValue;
    constexpr int per_cu = G::WG_PER_CU0 < 8 ? G::WG_PER_CU0 : 8;
    /* Workgroups that loop over the slab in step produce their loads and stores in bursts; how well the memory system takes
     * them depends on the allocation (the "two modes" of 2^8..2^10: 0.63 or 0.72 of the roofline from one hipMalloc block to
     * the next, profiles/r05/small_size_modes.txt).  At 2^8 and 2^9, where a table fill is cheap, sixteen times as many
     * workgroups (one or two iterations each on a 6 GiB slab) lift the slow mode by 5-7 % (0.633 -> 0.678, 0.636 -> 0.667;
     * inverse +4.5 %) and leave the fast one where it was; 2^10 and 2^11 lose what the larger tables cost, 2^6 and 2^7 are mixed:
     * unchanged (profiles/r05/small_size_grid.txt). */
    const int per_slot = pa.oversub > 0 ? pa.oversub : ((LOGN == 8 || LOGN == 9) ? 64 : 4); /* (NTT_OPT_BLOCK_OVERSUB: sweeps) */
    cap                  = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * (uint64_t)per_slot;
    /* 2^10: about six iterations per workgroup on large batches (32768 workgroups on a 6 GiB slab: slow mode 0.633 -> 0.653; the
     * 8192 of smaller batches stay, where more workgroups lost 2 %) */
    if(LOGN == 10 && pa.oversub <= 0 && wgs / 6 > cap) cap = wgs / 6;
  }
  if(pa.max_grid > 0) cap = (uint64_t)pa.max_grid;
  cap = cap / nl > 0 ? cap / nl : 1; /* the limbs of one launch share the resident workgroups */
  /* a persistent workgroup must always see the same block position inside the
   * polynomial (its LDS twiddle table depends on it): the grid, which is its
   * stride, is a multiple of the 2^s blocks per polynomial (nblocks always is) */
  if(G::BPW == 1 && pa.s > 0) {
    if(cap < (1ull << pa.s)) cap = 1ull << pa.s;
    cap &= ~((1ull << pa.s) - 1);
  }
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  p.wgs_per_limb = (uint32_t)wgs;
  const dim3 grid((unsigned)wgs, (unsigned)nl), wg(G::WG); /* (MULTI variants: blockIdx.y is the limb) */
  if(nl > 1) {
    /* several limbs in one launch: the MULTI variants, built for the FP64 policies (the ones RNS bases use) */
    if constexpr(multi_limb_built<A>()) {
      if constexpr(INV) {
        if(pa.lastinv) {
          hipLaunchKernelGGL((fused_kernel<A, LOGN, true, KSH, true, false, true>), grid, wg, 0, pa.stream, p);
        } else if constexpr(LOGN == kFusedLarge || LOGN == kFusedSmallBlock) {
          hipLaunchKernelGGL((fused_kernel<A, LOGN, true, KSH, false, false, true>), grid, wg, 0, pa.stream, p);
        } else {
          return hipErrorInvalidValue;
        }
      } else {
        if constexpr(A::kTracksBounds) { /* (lazy outputs: a kernel variant for the FP64 policies, a run-time flag for the integer ones) */
          if(pa.ends && pa.lazy) {
            hipLaunchKernelGGL((fused_kernel<A, LOGN, false, KSH, false, true, true>), grid, wg, 0, pa.stream, p);
            return hipGetLastError();
          }
        }
        hipLaunchKernelGGL((fused_kernel<A, LOGN, false, KSH, false, false, true>), grid, wg, 0, pa.stream, p);
      }
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
  if constexpr(INV) {
    /* the inverse kernel exists in two variants: ending a whole transform (N^-1 folded into
     * its last group) -- every block size -- and, for the block size used below column
     * passes, not ending it */
    if(pa.lastinv || A::kRadix4) {
      /* (radix-4 formulation: N^-1 is a pass of its own, fused into the LAST pass's store -- the blocks of a larger
       * transform run the same kernel with the multiplier record of 1: ntt_host.hip, limbrec_mid) */
      hipLaunchKernelGGL((fused_kernel<A, LOGN, true, KSH, true>), grid, wg, 0, pa.stream, p);
    } else if constexpr(LOGN == kFusedLarge || LOGN == kFusedSmallBlock) {
      hipLaunchKernelGGL((fused_kernel<A, LOGN, true, KSH, false>), grid, wg, 0, pa.stream, p);
    } else {
      return hipErrorInvalidValue;
    }
  } else {
    if constexpr(A::kTracksBounds) {
      if(pa.ends && pa.lazy) {
        hipLaunchKernelGGL((fused_kernel<A, LOGN, false, KSH, false, true>), grid, wg, 0, pa.stream, p);
        return hipGetLastError();
      }
    }
    hipLaunchKernelGGL((fused_kernel<A, LOGN, false, KSH, false, false>), grid, wg, 0, pa.stream, p);
  }
  return hipGetLastError();
}

/* pa.r = LEAD (1..3): the whole transform of 2^(14+LEAD) points in one launch; pa.batch polynomials */
/* Built for the FP64 policy at N = 2^16 and 2^17 (BASELINE configs 3 and 5).  The integer policy's larger
 * temporaries and the N = 2^15 inverse do not fit the 128-register budget of a 1024-thread workgroup without
 * scratch: those cases stay on the one-launch-per-pass path (ntt_host.hip: two_phase_applies). */
template <class A, int LEAD> constexpr bool two_phase_built() { return A::kTracksBounds && LEAD >= 2; }

template <class A, int LEAD, bool INV, int KSH> hipError_t launch_twophase(const PassArgs &pa)
{
  if constexpr(!two_phase_built<A, LEAD>()) {
    return hipErrorNotSupported;
  } else {
  if(pa.nlimbs > 1) return hipErrorNotSupported; /* (RNS sets take the per-pass launches) */
  KArgs<A> p = make_kargs<A>(pa);
  p.s0       = (uint32_t)LEAD;
  p.lastinv  = (uint32_t)pa.inverse;
  p.nblocks  = pa.batch;
  uint64_t wgs = pa.batch;
  uint64_t cap = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256);
  if(pa.max_grid > 0) cap = (uint64_t)pa.max_grid;
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  p.wgs_per_limb = (uint32_t)wgs;
  hipLaunchKernelGGL((twophase_kernel<A, LEAD, INV, KSH>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, p);
  return hipGetLastError();
  }
}

/* N = 2^15 in one pass (onepass_kernel): one persistent 1024-thread workgroup per CU, pa.batch polynomials per limb */
template <class A> constexpr bool onepass_built() { return A::kCompact && A::kTracksBounds; }
template <class A, bool INV, int KSH> hipError_t launch_onepass(const PassArgs &pa)
{
  if constexpr(!onepass_built<A>()) {
    return hipErrorNotSupported;
  } else {
    if(pa.logn != (uint32_t)kFusedLarge + 1) return hipErrorNotSupported; /* (a lazy call gets canonical words: inside the lazy ranges) */
    const uint64_t nl = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
    if(nl > (uint64_t)kMaxLimbs) return hipErrorNotSupported;
    KArgs<A> p = make_kargs<A>(pa);
    p.s0       = 1;
    p.lastinv  = (uint32_t)pa.inverse;
    p.lazy     = 0;
    p.nblocks  = pa.batch;
    uint64_t wgs = pa.batch;
    uint64_t cap = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256);
    if(pa.max_grid > 0) cap = (uint64_t)pa.max_grid;
    cap = cap / nl > 0 ? cap / nl : 1;
    if(wgs > cap) wgs = cap;
    if(wgs == 0) return hipSuccess;
    p.wgs_per_limb = (uint32_t)wgs;
    const dim3 grid((unsigned)wgs, (unsigned)nl);
    if(nl > 1) hipLaunchKernelGGL((onepass_kernel<A, INV, KSH, true>), grid, dim3(1024), 0, pa.stream, p);
    else hipLaunchKernelGGL((onepass_kernel<A, INV, KSH, false>), grid, dim3(1024), 0, pa.stream, p);
    return hipGetLastError();
  }
}

/* pa.r = LEAD (3..5), pa.batch polynomials of 2^(12 + LEAD) points per limb; pa.team_ctl: TeamCtl with nlimbs * batch counters,
 * zeroed here.  Several limbs (an RNS set, [limb][batch][N]): the MULTI variant, the queues run over all limbs' polynomials. */
/* Zeroes a control block (queue heads, owners, per-polynomial counters) in front of an XCD-local launch -- as a KERNEL, not as an
 * asynchronous memset: captured into a HIP graph, a memset node in front of the kernel node did not always take effect before the
 * kernel's first workgroups read the counters (stale counters of the previous replay: second-pass items that do not wait, or
 * queues that look exhausted -- found by replaying a captured NTT-domain product between other work, round 5:
 * tests/test_gpu_parity.py::test_one_launch_ntt_domain_products_captured_in_a_hip_graph).  A kernel in front of a kernel on the
 * same stream is ordered in a graph exactly as outside one. */
static __global__ void __launch_bounds__(256) team_ctl_clear_kernel(unsigned *w, size_t n)
{
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) w[i] = 0u;
}
static inline hipError_t team_ctl_clear(void *ctl, size_t bytes, hipStream_t stream)
{
  const size_t n = (bytes + 3) / 4;
  size_t       g = (n + 255) / 256;
  if(g > 64) g = 64;
  hipLaunchKernelGGL(team_ctl_clear_kernel, dim3((unsigned)g), dim3(256), 0, stream, static_cast<unsigned *>(ctl), n);
  return hipGetLastError();
}

template <class A, int LEAD, bool INV, int KSH> hipError_t launch_team(const PassArgs &pa)
{
  if constexpr(!(A::kCompact || A::kIntWide)) {
    return hipErrorNotSupported;
  } else {
    const uint64_t nl = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
    if(nl > (uint64_t)kMaxLimbs || !pa.team_ctl || pa.wide || pa.lazy || nl * pa.batch >= (1ull << 31)) return hipErrorNotSupported;
    KTeam<A> kt{};
    kt.k         = make_kargs<A>(pa);
    kt.k.lazy    = 0; /* canonical out (the integer policies read this flag at run time) */
    kt.k.lastinv = (uint32_t)pa.inverse;
    kt.k.nblocks = pa.batch;
    kt.ctl       = static_cast<TeamCtl *>(pa.team_ctl);
    kt.lag       = (uint32_t)(pa.team_lag > 0 ? pa.team_lag : 6);
    kt.nlimbs    = (uint32_t)nl;
    kt.poly_major = nl > 1 && kt.k.poly_stride > kt.k.limb_stride;
    kt.split_rcp  = team_split_rcp(kt.poly_major ? nl : pa.batch);
    const size_t bytes = sizeof(TeamCtl) + (size_t)(nl * pa.batch) * sizeof(unsigned);
    hipError_t   e     = team_ctl_clear(pa.team_ctl, bytes, pa.stream);
    if(e != hipSuccess) return e;
    /* four workgroups per CU: 40,580 bytes of LDS each (32.9 KB exchange buffer + 7.5 KB table), at most 128 VGPRs */
    uint64_t wgs = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * (pa.team_wpc > 0 ? pa.team_wpc : 4);
    if(pa.max_grid > 0) wgs = (uint64_t)pa.max_grid;
    kt.k.wgs_per_limb = (uint32_t)wgs;
    if(nl > 1) hipLaunchKernelGGL((team_kernel<A, LEAD, INV, KSH, true>), dim3((unsigned)wgs), dim3(256), 0, pa.stream, kt);
    else hipLaunchKernelGGL((team_kernel<A, LEAD, INV, KSH, false>), dim3((unsigned)wgs), dim3(256), 0, pa.stream, kt);
    return hipGetLastError();
  }
}

template <class A, int R, bool INV, int KSH> hipError_t launch_column(const PassArgs &pa)
{
  KArgs<A> p = make_kargs<A>(pa);
  p.s0       = (uint32_t)pa.s;
  p.lazy     = (uint32_t)pass_lazy(pa);
  p.nblocks  = pa.batch;
  const uint64_t nl    = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
  const uint64_t total = pa.batch << (pa.logn - R);
  uint64_t       wgs   = (total + 255) / 256;
  uint64_t       cap   = pa.max_grid > 0 ? (uint64_t)pa.max_grid : (1ull << 22);
  cap                  = cap / nl > 0 ? cap / nl : 1;
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  p.wgs_per_limb = (uint32_t)wgs;
  if(nl > 1) {
    if constexpr(multi_limb_built<A>()) {
      hipLaunchKernelGGL((column_kernel<A, R, INV, KSH, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(256), 0, pa.stream, p);
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
  hipLaunchKernelGGL((column_kernel<A, R, INV, KSH>), dim3((unsigned)wgs), dim3(256), 0, pa.stream, p);
  return hipGetLastError();
}

template <class A, int KSH> hipError_t launch_product_impl(const ProdArgs &pa)
{
  if constexpr(!A::kCompact) {
    return hipErrorNotSupported;
  } else {
    if(pa.logn < 8 || pa.logn > 17) return hipErrorNotSupported;
    const uint32_t blog = pa.logn <= 14 ? pa.logn : (pa.block_log ? pa.block_log : 14u);
    if(blog < 12 && pa.logn > 14) return hipErrorInvalidValue;
    const uint32_t s0 = pa.logn - blog; /* leading stages done by column passes around this launch */
    KProd<A> pp{};
    pp.f.a            = pa.b;
    const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(pa.limbs);
    for(int l = 0; l < (pa.nlimbs > 0 ? pa.nlimbs : 1) && l < kMaxLimbs; l++) pp.f.limbs[l] = recs[l];
    pp.f.limb_stride  = pa.limb_stride;
    pp.f.poly_stride  = pa.poly_stride ? pa.poly_stride : (1ull << pa.logn);
    pp.f.wgs_per_limb = 1;
    pp.f.logn         = pa.logn;
    pp.f.s0           = s0;
    pp.f.nblocks      = pa.batch << s0;
    pp.ahat           = pa.ahat;
    pp.out            = pa.out;
    pp.a_lazy         = (uint32_t)pa.a_lazy;
    const uint64_t nl = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
    uint64_t wgs = pp.f.nblocks;
    uint64_t cap = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256);
    if(pa.max_grid > 0) cap = (uint64_t)pa.max_grid;
    cap = cap / nl > 0 ? cap / nl : 1;
    /* a workgroup keeps the tables of ONE block position: its stride is a multiple of the blocks per polynomial */
    if(cap < (1ull << s0)) cap = 1ull << s0;
    cap &= ~((1ull << s0) - 1);
    if(wgs > cap) wgs = cap;
    if(wgs == 0) return hipSuccess;
    /* a^ always arrives as the lazy words ntt_fwd_batch_lazy leaves (the canonical-operand variant is not built) -- or not
     * at all: pa.both, whole polynomials, a's coefficients in pa.ahat */
    if(!pa.a_lazy && !pa.both) return hipErrorNotSupported;
    if(pa.both && s0 != 0 && blog != 12 && blog != 14) return hipErrorInvalidValue;
    if(pa.ptrs) {
      /* separately held polynomials: the three operand pointers are tables (fused_product_kernel's PTRS form) */
      /* (several limbs: the limbs of every polynomial pa.limb_stride words apart behind its table entry -- the MULTI instances,
       * whose limb_params add limb * limb_stride to the offset in pp.f.a) */
      if(!pa.both || s0 != 0) return hipErrorNotSupported;
      pp.f.ptab = reinterpret_cast<const uint64_t *>(pa.b);
      pp.f.a    = reinterpret_cast<uint64_t *>((uintptr_t)pa.ptr_limb_off * 8u);
    }
    if(blog < 12) {
      switch(pa.logn) {
#define NTT_SMALL_PRODUCT(LN)                                                                                       \
  case LN: {                                                                                                        \
    using GS = Geom<LN, false, 3>;                                                                                  \
    constexpr int per_cu = GS::WG_PER_CU0 < 8 ? GS::WG_PER_CU0 : 8;                                                 \
    uint64_t      g      = (pp.f.nblocks + GS::BPW - 1) / GS::BPW;                                                  \
    uint64_t      gcap   = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * 4;           \
    if(pa.max_grid > 0) gcap = (uint64_t)pa.max_grid;                                                               \
    gcap = gcap / nl > 0 ? gcap / nl : 1;                                                                           \
    if(g > gcap) g = gcap;                                                                                          \
    pp.f.wgs_per_limb = (unsigned)g;                                                                                \
    if(pa.ptrs) {                                                                                                   \
      if(nl > 1) hipLaunchKernelGGL((fused_product_small_kernel<A, LN, KSH, true, true, true>), dim3((unsigned)g, (unsigned)nl), dim3(GS::WG), 0, pa.stream, pp); \
      else hipLaunchKernelGGL((fused_product_small_kernel<A, LN, KSH, false, true, true>), dim3((unsigned)g), dim3(GS::WG), 0, pa.stream, pp); \
      return hipGetLastError();                                                                                     \
    }                                                                                                               \
    if(pa.both) {                                                                                                   \
      if(nl > 1) hipLaunchKernelGGL((fused_product_small_kernel<A, LN, KSH, true, true>), dim3((unsigned)g, (unsigned)nl), dim3(GS::WG), 0, pa.stream, pp); \
      else hipLaunchKernelGGL((fused_product_small_kernel<A, LN, KSH, false, true>), dim3((unsigned)g), dim3(GS::WG), 0, pa.stream, pp); \
      return hipGetLastError();                                                                                     \
    }                                                                                                               \
    if(nl > 1) hipLaunchKernelGGL((fused_product_small_kernel<A, LN, KSH, true>), dim3((unsigned)g, (unsigned)nl), dim3(GS::WG), 0, pa.stream, pp); \
    else hipLaunchKernelGGL((fused_product_small_kernel<A, LN, KSH>), dim3((unsigned)g), dim3(GS::WG), 0, pa.stream, pp); \
    return hipGetLastError();                                                                                       \
  }
        NTT_SMALL_PRODUCT(8)
        NTT_SMALL_PRODUCT(9)
        NTT_SMALL_PRODUCT(10)
        NTT_SMALL_PRODUCT(11)
#undef NTT_SMALL_PRODUCT
        default: return hipErrorNotSupported;
      }
    }
    if(blog == 12) {
      using G12 = Geom<12, false, 3>;
      constexpr int per_cu = G12::WG_PER_CU0 < G12::WPS ? G12::WG_PER_CU0 : G12::WPS; /* 256-thread workgroups: one wave per SIMD each */
      /* (whole polynomials: several workgroups per resident slot, as for the transforms -- block_oversub; measured 0.369 -> 0.401 of
       * the 24N roofline at 8 per slot, profiles/r05/oversub_sweep.txt) */
      uint64_t      cap12  = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * per_cu * block_oversub<12, G12::WG>(pa.oversub, s0 == 0);
      if(pa.max_grid > 0) cap12 = (uint64_t)pa.max_grid;
      cap12 = cap12 / nl > 0 ? cap12 / nl : 1;
      if(cap12 < (1ull << s0)) cap12 = 1ull << s0;
      cap12 &= ~((1ull << s0) - 1);
      wgs = pp.f.nblocks < cap12 ? pp.f.nblocks : cap12;
      pp.f.wgs_per_limb = (uint32_t)wgs;
      if(pa.ptrs) {
        if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, true, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G12::WG), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, true, false, true, true>), dim3((unsigned)wgs), dim3(G12::WG), 0, pa.stream, pp);
        return hipGetLastError();
      }
      if(pa.both) {
        if(s0 == 0) {
          if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G12::WG), 0, pa.stream, pp);
          else hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, true, false, true>), dim3((unsigned)wgs), dim3(G12::WG), 0, pa.stream, pp);
        } else {
          if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, false, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G12::WG), 0, pa.stream, pp);
          else hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, false, false, true>), dim3((unsigned)wgs), dim3(G12::WG), 0, pa.stream, pp);
        }
        return hipGetLastError();
      }
      if(nl > 1) {
        if(s0 == 0) hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G12::WG), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, false, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G12::WG), 0, pa.stream, pp);
      } else {
        if(s0 == 0) hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, true>), dim3((unsigned)wgs), dim3(G12::WG), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, false>), dim3((unsigned)wgs), dim3(G12::WG), 0, pa.stream, pp);
      }
      return hipGetLastError();
    }
    if(blog == 13) {
      using G13 = Geom<13, false, 3>;
      /* one 512-thread workgroup per CU by LDS (64 KB exchange buffer + 30 KB table); a second one does not fit */
      if(s0 != 0) return hipErrorNotSupported; /* (2^13-point blocks of a larger product: measured no faster than 2^14, not built) */
      pp.f.wgs_per_limb = (uint32_t)wgs;
      if(pa.ptrs) {
        if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 13, KSH, true, true, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G13::WG), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 13, KSH, true, true, false, true, true>), dim3((unsigned)wgs), dim3(G13::WG), 0, pa.stream, pp);
        return hipGetLastError();
      }
      if(pa.both) {
        if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 13, KSH, true, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G13::WG), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 13, KSH, true, true, false, true>), dim3((unsigned)wgs), dim3(G13::WG), 0, pa.stream, pp);
        return hipGetLastError();
      }
      if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 13, KSH, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G13::WG), 0, pa.stream, pp);
      else hipLaunchKernelGGL((fused_product_kernel<A, 13, KSH, true, true>), dim3((unsigned)wgs), dim3(G13::WG), 0, pa.stream, pp);
      return hipGetLastError();
    }
    pp.f.wgs_per_limb = (uint32_t)wgs;
    if(pa.ptrs) {
      if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, true, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(1024), 0, pa.stream, pp);
      else hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, true, false, true, true>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, pp);
      return hipGetLastError();
    }
    if(pa.both) {
      if(s0 == 0) {
        if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(1024), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, true, false, true>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, pp);
      } else {
        if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, false, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(1024), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, false, false, true>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, pp);
      }
      return hipGetLastError();
    }
    if(nl > 1) {
      if(s0 == 0) hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(1024), 0, pa.stream, pp);
      else hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, false, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(1024), 0, pa.stream, pp);
    } else {
      if(s0 == 0) hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, true>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, pp);
      else hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, false>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, pp);
    }
    return hipGetLastError();
  }
}

/* a product at N = 2^15..2^17 as one launch (team_product_kernel); pa.team_ctl: TeamProdCtl + 2 * nlimbs * batch counters.
 * Several limbs ([limb][batch][N] slabs, limb_stride apart): the MULTI variants -- ONE launch for a whole RNS product. */
template <class A, int KSH> hipError_t launch_team_product_impl(const ProdArgs &pa)
{
  if constexpr(!A::kCompact) {
    return hipErrorNotSupported;
  } else {
    const uint64_t nl = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
    if(nl > (uint64_t)kMaxLimbs || !pa.team_ctl || (!pa.a_lazy && !pa.four) || pa.logn < kTeamBlock + 3 || pa.logn > kTeamBlock + 5 ||
       nl * pa.batch >= (1ull << 30)) {
      return hipErrorNotSupported;
    }
    KTeamProd<A> kt{};
    kt.k.f.a            = pa.b;
    const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(pa.limbs);
    for(uint64_t l = 0; l < nl; l++) kt.k.f.limbs[l] = recs[l];
    kt.k.f.limb_stride  = nl > 1 ? pa.limb_stride : 0;
    kt.k.f.poly_stride  = pa.poly_stride ? pa.poly_stride : (1ull << pa.logn);
    kt.k.f.logn         = pa.logn;
    kt.k.f.s0           = pa.logn - kTeamBlock;
    kt.k.f.nblocks      = pa.batch;
    kt.k.ahat           = pa.ahat;
    kt.k.out            = pa.out;
    kt.k.a_lazy         = 1;
    kt.ctl              = static_cast<TeamProdCtl *>(pa.team_ctl);
    kt.lag              = (uint32_t)(pa.team_lag > 0 ? pa.team_lag : 8);
    kt.nlimbs           = (uint32_t)nl;
    kt.poly_major       = nl > 1 && kt.k.f.poly_stride > kt.k.f.limb_stride;
    kt.split_rcp        = team_split_rcp(kt.poly_major ? nl : pa.batch);
    const size_t bytes = sizeof(TeamProdCtl) + 2 * (size_t)(nl * pa.batch) * sizeof(unsigned);
    hipError_t   e     = team_ctl_clear(pa.team_ctl, bytes, pa.stream);
    if(e != hipSuccess) return e;
    /* four workgroups per CU (121 VGPRs, 40.6 KB of LDS each) */
    uint64_t wgs = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * (pa.team_wpc > 0 ? pa.team_wpc : 4);
    if(pa.max_grid > 0) wgs = (uint64_t)pa.max_grid;
    kt.k.f.wgs_per_limb = (uint32_t)wgs;
    const dim3 g((unsigned)wgs), t(256);
#define NTT_TEAM_PROD(LEADV, FOURV)                                                                                \
  do {                                                                                                             \
    if(nl > 1) hipLaunchKernelGGL((team_product_kernel<A, LEADV, KSH, FOURV, true>), g, t, 0, pa.stream, kt);      \
    else hipLaunchKernelGGL((team_product_kernel<A, LEADV, KSH, FOURV, false>), g, t, 0, pa.stream, kt);           \
  } while(0)
    if(pa.ptrs) {
      /* separately held polynomials: b, ahat (a's coefficients) and out are device tables (team_product_kernel's PTRS form) */
      /* (several limbs: every polynomial's limbs pa.limb_stride words apart behind its table entry -- the MULTI instances) */
      if(!pa.four) return hipErrorNotSupported;
      kt.k.f.ptab = reinterpret_cast<const uint64_t *>(pa.b);
      kt.k.f.a    = reinterpret_cast<uint64_t *>((uintptr_t)pa.ptr_limb_off * 8u);
#define NTT_TEAM_PROD_PTRS(LEADV)                                                                                  \
  do {                                                                                                             \
    if(nl > 1) hipLaunchKernelGGL((team_product_kernel<A, LEADV, KSH, true, true, true>), g, t, 0, pa.stream, kt); \
    else hipLaunchKernelGGL((team_product_kernel<A, LEADV, KSH, true, false, true>), g, t, 0, pa.stream, kt);      \
  } while(0)
      switch(pa.logn - kTeamBlock) {
        case 3: NTT_TEAM_PROD_PTRS(3); break;
        case 4: NTT_TEAM_PROD_PTRS(4); break;
        default: NTT_TEAM_PROD_PTRS(5); break;
      }
#undef NTT_TEAM_PROD_PTRS
      return hipGetLastError();
    }
    if(pa.four) {
      /* ahat = a itself (coefficients): both forward transforms happen inside the launch */
      switch(pa.logn - kTeamBlock) {
        case 3: NTT_TEAM_PROD(3, true); break;
        case 4: NTT_TEAM_PROD(4, true); break;
        default: NTT_TEAM_PROD(5, true); break;
      }
      return hipGetLastError();
    }
    switch(pa.logn - kTeamBlock) {
      case 3: NTT_TEAM_PROD(3, false); break;
      case 4: NTT_TEAM_PROD(4, false); break;
      default: NTT_TEAM_PROD(5, false); break;
    }
#undef NTT_TEAM_PROD
    return hipGetLastError();
  }
}

template <class A, int LOGN, int KSH, bool LASTINV> hipError_t launch_dot_blocks(const DotArgs &da)
{
  using G = Geom<LOGN, true, flavor_of<A>()>;
  KDot<A> kd{};
  kd.k.a                 = da.out;
  const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(da.limbs);
  const uint64_t    nl   = (uint64_t)(da.nlimbs > 0 ? da.nlimbs : 1);
  for(uint64_t l = 0; l < nl && l < (uint64_t)kMaxLimbs; l++) kd.k.limbs[l] = recs[l];
  kd.k.limb_stride = da.limb_stride;
  kd.k.poly_stride = da.poly_stride ? da.poly_stride : (1ull << da.logn);
  kd.k.logn        = da.logn;
  kd.k.s0          = da.logn - (uint32_t)LOGN;
  kd.k.lastinv     = LASTINV ? 1u : 0u;
  kd.k.lazy        = LASTINV ? 0u : 1u;
  kd.k.nblocks     = da.batch << kd.k.s0;
  kd.npairs        = (uint32_t)da.npairs;
  kd.lazy_in       = (uint32_t)da.lazy_in;
  kd.b_bcast       = (uint32_t)da.b_bcast;
  kd.b_limb_stride = da.b_limb_stride;
  for(int i = 0; i < da.npairs && i < kMaxDot; i++) {
    kd.a[i] = da.a[i];
    kd.b[i] = da.b[i];
  }
  /* the grid of the inverse block kernel (launch_fused): resident workgroups striding over the blocks */
  uint64_t wgs = (kd.k.nblocks + G::BPW - 1) / G::BPW;
  uint64_t cap = 1ull << 20;
  if(G::PERSISTENT) {
    constexpr int by_lds   = G::WG_PER_CU0;
    constexpr int by_waves = (G::WPS * 4 * 64) / G::WG;
    constexpr int per_cu   = by_lds < by_waves ? by_lds : by_waves;
    cap                    = (uint64_t)(da.num_cus > 0 ? da.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * block_oversub<LOGN, G::WG>(da.oversub, kd.k.s0 == 0);
  }
  if(!G::PERSISTENT && G::LDS_TW > 0) {
    if(kd.k.s0 != 0) return hipErrorInvalidValue;
    constexpr int per_cu = G::WG_PER_CU0 < 8 ? G::WG_PER_CU0 : 8;
    cap                  = (uint64_t)(da.num_cus > 0 ? da.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * 4;
  }
  if(da.max_grid > 0) cap = (uint64_t)da.max_grid;
  cap = cap / nl > 0 ? cap / nl : 1;
  if(G::BPW == 1 && kd.k.s0 > 0) { /* a workgroup keeps the tables of ONE block position */
    if(cap < (1ull << kd.k.s0)) cap = 1ull << kd.k.s0;
    cap &= ~((1ull << kd.k.s0) - 1);
  }
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  kd.k.wgs_per_limb = (uint32_t)wgs;
  if(da.ptrs) {
    if constexpr(LASTINV) {
      kd.k.ptab = reinterpret_cast<const uint64_t *>(da.out);
      kd.k.a    = reinterpret_cast<uint64_t *>((uintptr_t)da.ptr_limb_off * 8u);
      if(nl > 1) {
        /* the limbs of an RNS set behind every table entry, da.limb_stride words apart: one launch over all of them */
        if constexpr(multi_limb_built<A>()) {
          hipLaunchKernelGGL((dot_inv_kernel<A, LOGN, KSH, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G::WG), 0, da.stream, kd);
          return hipGetLastError();
        } else {
          return hipErrorNotSupported;
        }
      }
      hipLaunchKernelGGL((dot_inv_kernel<A, LOGN, KSH, true, false, true>), dim3((unsigned)wgs), dim3(G::WG), 0, da.stream, kd);
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
  if(nl > 1) {
    if constexpr(multi_limb_built<A>()) {
      hipLaunchKernelGGL((dot_inv_kernel<A, LOGN, KSH, LASTINV, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G::WG), 0, da.stream, kd);
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
  hipLaunchKernelGGL((dot_inv_kernel<A, LOGN, KSH, LASTINV, false>), dim3((unsigned)wgs), dim3(G::WG), 0, da.stream, kd);
  return hipGetLastError();
}

/* the NTT-domain product at N = 2^15..2^17 as ONE launch (team_dot_kernel); da.team_ctl: TeamCtl + nlimbs * batch counters, zeroed here */
template <class A, int KSH> hipError_t launch_team_dot(const DotArgs &da)
{
  if constexpr(!(A::kCompact || A::kIntWide)) {
    return hipErrorNotSupported;
  } else {
    const uint64_t nl = (uint64_t)(da.nlimbs > 0 ? da.nlimbs : 1);
    if(nl > (uint64_t)kMaxLimbs || !da.team_ctl || da.logn < (uint32_t)kTeamBlock + 3 || da.logn > (uint32_t)kTeamBlock + 5 ||
       nl * da.batch >= (1ull << 31) || da.npairs < 1 || da.npairs > kMaxDot) {
      return hipErrorNotSupported;
    }
    KTeamDot<A> kt{};
    kt.d.k.a                = da.out;
    const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(da.limbs);
    for(uint64_t l = 0; l < nl; l++) kt.d.k.limbs[l] = recs[l];
    kt.d.k.limb_stride = nl > 1 ? da.limb_stride : 0;
    kt.d.k.poly_stride = da.poly_stride ? da.poly_stride : (1ull << da.logn);
    kt.d.k.logn        = da.logn;
    kt.d.k.s0          = da.logn - (uint32_t)kTeamBlock;
    kt.d.k.lastinv     = 1;
    kt.d.k.lazy        = 0;
    kt.d.k.nblocks     = da.batch;
    kt.d.npairs        = (uint32_t)da.npairs;
    kt.d.lazy_in       = (uint32_t)da.lazy_in;
    kt.d.b_bcast       = (uint32_t)da.b_bcast;
    kt.d.b_limb_stride = nl > 1 ? da.b_limb_stride : 0;
    for(int i = 0; i < da.npairs; i++) {
      kt.d.a[i] = da.a[i];
      kt.d.b[i] = da.b[i];
    }
    kt.ctl        = static_cast<TeamCtl *>(da.team_ctl);
    kt.lag        = (uint32_t)(da.team_lag > 0 ? da.team_lag : 8);
    kt.nlimbs     = (uint32_t)nl;
    kt.poly_major = nl > 1 && kt.d.k.poly_stride > kt.d.k.limb_stride;
    kt.split_rcp  = team_split_rcp(kt.poly_major ? nl : da.batch);
    const size_t bytes = sizeof(TeamCtl) + (size_t)(nl * da.batch) * sizeof(unsigned);
    hipError_t   e     = team_ctl_clear(da.team_ctl, bytes, da.stream);
    if(e != hipSuccess) return e;
    uint64_t wgs = (uint64_t)(da.num_cus > 0 ? da.num_cus : 256) * (da.team_wpc > 0 ? da.team_wpc : 4);
    if(da.max_grid > 0) wgs = (uint64_t)da.max_grid;
    kt.d.k.wgs_per_limb = (uint32_t)wgs;
    const dim3 g((unsigned)wgs), t(256);
    if(da.ptrs) {
      kt.d.k.ptab = reinterpret_cast<const uint64_t *>(da.out);
      kt.d.k.a    = reinterpret_cast<uint64_t *>((uintptr_t)da.ptr_limb_off * 8u);
#define NTT_TEAM_DOT_PTRS(LEADV)                                                                             \
  do {                                                                                                       \
    if(nl > 1) hipLaunchKernelGGL((team_dot_kernel<A, LEADV, KSH, true, true>), g, t, 0, da.stream, kt);     \
    else hipLaunchKernelGGL((team_dot_kernel<A, LEADV, KSH, false, true>), g, t, 0, da.stream, kt);          \
  } while(0)
      switch(da.logn - kTeamBlock) {
        case 3: NTT_TEAM_DOT_PTRS(3); break;
        case 4: NTT_TEAM_DOT_PTRS(4); break;
        default: NTT_TEAM_DOT_PTRS(5); break;
      }
#undef NTT_TEAM_DOT_PTRS
      return hipGetLastError();
    }
#define NTT_TEAM_DOT(LEADV)                                                                                  \
  do {                                                                                                       \
    if(nl > 1) hipLaunchKernelGGL((team_dot_kernel<A, LEADV, KSH, true>), g, t, 0, da.stream, kt);           \
    else hipLaunchKernelGGL((team_dot_kernel<A, LEADV, KSH, false>), g, t, 0, da.stream, kt);                \
  } while(0)
    switch(da.logn - kTeamBlock) {
      case 3: NTT_TEAM_DOT(3); break;
      case 4: NTT_TEAM_DOT(4); break;
      default: NTT_TEAM_DOT(5); break;
    }
#undef NTT_TEAM_DOT
    return hipGetLastError();
  }
}

template <class A, int KSH> hipError_t launch_dot_impl(const DotArgs &da)
{
  if(da.npairs < 1 || da.npairs > kMaxDot || da.nlimbs > kMaxLimbs) return hipErrorInvalidValue;
  if(da.team_ctl) return launch_team_dot<A, KSH>(da);
  if(da.logn > (uint32_t)kFusedMax) {
    if(da.ptrs) return hipErrorNotSupported;
    if(da.block_log == (uint32_t)kFusedSmallBlock) return launch_dot_blocks<A, kFusedSmallBlock, KSH, false>(da);
    if(da.block_log == (uint32_t)kFusedLarge) return launch_dot_blocks<A, kFusedLarge, KSH, false>(da);
    return hipErrorInvalidValue;
  }
  switch(da.logn) {
#define NTT_DOT_CASE(LN) \
  case LN: return launch_dot_blocks<A, LN, KSH, true>(da);
    NTT_DOT_CASE(6) NTT_DOT_CASE(7) NTT_DOT_CASE(8) NTT_DOT_CASE(9) NTT_DOT_CASE(10) NTT_DOT_CASE(11) NTT_DOT_CASE(12) NTT_DOT_CASE(13)
    NTT_DOT_CASE(14)
#undef NTT_DOT_CASE
    default: return hipErrorNotSupported;
  }
}

template <class A, int LOGN, int KSH> hipError_t launch_fwd_mul_blocks(const MulArgs &ma)
{
  using G = Geom<LOGN, false, flavor_of<A>()>;
  KMul<A> km{};
  km.k.a                 = ma.a;
  const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(ma.limbs);
  const uint64_t    nl   = (uint64_t)(ma.nlimbs > 0 ? ma.nlimbs : 1);
  for(uint64_t l = 0; l < nl && l < (uint64_t)kMaxLimbs; l++) km.k.limbs[l] = recs[l];
  km.k.limb_stride = ma.limb_stride;
  km.k.poly_stride = ma.poly_stride ? ma.poly_stride : (1ull << ma.logn);
  km.k.logn        = ma.logn;
  km.k.s0          = ma.logn - (uint32_t)LOGN;
  km.k.nblocks     = ma.batch << km.k.s0;
  km.b             = ma.b;
  km.out           = ma.out;
  km.b_limb_stride = ma.b_limb_stride;
  km.lazy_in       = (uint32_t)ma.lazy_in;
  km.b_bcast       = (uint32_t)ma.b_bcast;
  km.accumulate    = (uint32_t)ma.accumulate;
  /* the grid of the forward block kernel (launch_fused) */
  uint64_t wgs = (km.k.nblocks + G::BPW - 1) / G::BPW;
  uint64_t cap = 1ull << 20;
  if(G::PERSISTENT) {
    constexpr int by_lds   = G::WG_PER_CU0;
    constexpr int by_waves = (G::WPS * 4 * 64) / G::WG;
    constexpr int per_cu   = by_lds < by_waves ? by_lds : by_waves;
    cap                    = (uint64_t)(ma.num_cus > 0 ? ma.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * block_oversub<LOGN, G::WG>(ma.oversub, km.k.s0 == 0);
  }
  if(!G::PERSISTENT && G::LDS_TW > 0) {
    if(km.k.s0 != 0) return hipErrorInvalidValue;
    constexpr int per_cu = G::WG_PER_CU0 < 8 ? G::WG_PER_CU0 : 8;
    cap                  = (uint64_t)(ma.num_cus > 0 ? ma.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * 4;
  }
  if(ma.max_grid > 0) cap = (uint64_t)ma.max_grid;
  cap = cap / nl > 0 ? cap / nl : 1;
  if(G::BPW == 1 && km.k.s0 > 0) { /* a workgroup keeps the tables of ONE block position */
    if(cap < (1ull << km.k.s0)) cap = 1ull << km.k.s0;
    cap &= ~((1ull << km.k.s0) - 1);
  }
  if(G::BPW > 1 && km.k.s0 > 0) return hipErrorInvalidValue; /* (two blocks per workgroup: whole polynomials only) */
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  km.k.wgs_per_limb = (uint32_t)wgs;
  if(ma.ptrs) {
    if(km.k.s0 != 0) return hipErrorNotSupported;
    km.k.ptab = reinterpret_cast<const uint64_t *>(ma.a);
    km.k.a    = reinterpret_cast<uint64_t *>((uintptr_t)ma.ptr_limb_off * 8u);
    if(nl > 1) {
      /* the limbs of an RNS set behind every table entry, ma.limb_stride words apart: one launch over all of them */
      if constexpr(multi_limb_built<A>()) {
        hipLaunchKernelGGL((fwd_mul_kernel<A, LOGN, KSH, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G::WG), 0, ma.stream, km);
        return hipGetLastError();
      } else {
        return hipErrorNotSupported;
      }
    }
    hipLaunchKernelGGL((fwd_mul_kernel<A, LOGN, KSH, false, true>), dim3((unsigned)wgs), dim3(G::WG), 0, ma.stream, km);
    return hipGetLastError();
  }
  if(nl > 1) {
    if constexpr(multi_limb_built<A>()) {
      hipLaunchKernelGGL((fwd_mul_kernel<A, LOGN, KSH, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G::WG), 0, ma.stream, km);
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
  hipLaunchKernelGGL((fwd_mul_kernel<A, LOGN, KSH, false>), dim3((unsigned)wgs), dim3(G::WG), 0, ma.stream, km);
  return hipGetLastError();
}

template <class A, int KSH> hipError_t launch_team_mul(const MulArgs &ma)
{
  if constexpr(!(A::kCompact || A::kIntWide)) {
    return hipErrorNotSupported;
  } else {
    const uint64_t nl = (uint64_t)(ma.nlimbs > 0 ? ma.nlimbs : 1);
    if(nl > (uint64_t)kMaxLimbs || !ma.team_ctl || ma.logn < (uint32_t)kTeamBlock + 3 || ma.logn > (uint32_t)kTeamBlock + 5 ||
       nl * ma.batch >= (1ull << 31)) {
      return hipErrorNotSupported;
    }
    KTeamMul<A> kt{};
    kt.m.k.a               = ma.a;
    const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(ma.limbs);
    for(uint64_t l = 0; l < nl; l++) kt.m.k.limbs[l] = recs[l];
    kt.m.k.limb_stride = nl > 1 ? ma.limb_stride : 0;
    kt.m.k.poly_stride = ma.poly_stride ? ma.poly_stride : (1ull << ma.logn);
    kt.m.k.logn        = ma.logn;
    kt.m.k.s0          = ma.logn - (uint32_t)kTeamBlock;
    kt.m.k.lazy        = 0;
    kt.m.k.nblocks     = ma.batch;
    kt.m.b             = ma.b;
    kt.m.out           = ma.out;
    kt.m.b_limb_stride = nl > 1 ? ma.b_limb_stride : 0;
    kt.m.lazy_in       = (uint32_t)ma.lazy_in;
    kt.m.b_bcast       = (uint32_t)ma.b_bcast;
    kt.m.accumulate    = (uint32_t)ma.accumulate;
    kt.ctl             = static_cast<TeamCtl *>(ma.team_ctl);
    kt.lag             = (uint32_t)(ma.team_lag > 0 ? ma.team_lag : 8);
    kt.nlimbs          = (uint32_t)nl;
    kt.poly_major      = nl > 1 && kt.m.k.poly_stride > kt.m.k.limb_stride;
    kt.split_rcp       = team_split_rcp(kt.poly_major ? nl : ma.batch);
    const size_t bytes = sizeof(TeamCtl) + (size_t)(nl * ma.batch) * sizeof(unsigned);
    hipError_t   e     = team_ctl_clear(ma.team_ctl, bytes, ma.stream);
    if(e != hipSuccess) return e;
    uint64_t wgs = (uint64_t)(ma.num_cus > 0 ? ma.num_cus : 256) * (ma.team_wpc > 0 ? ma.team_wpc : 4);
    if(ma.max_grid > 0) wgs = (uint64_t)ma.max_grid;
    kt.m.k.wgs_per_limb = (uint32_t)wgs;
    const dim3 g((unsigned)wgs), t(256);
    if(ma.ptrs) {
      kt.m.k.ptab = reinterpret_cast<const uint64_t *>(ma.a);
      kt.m.k.a    = reinterpret_cast<uint64_t *>((uintptr_t)ma.ptr_limb_off * 8u);
#define NTT_TEAM_MUL_PTRS(LEADV)                                                                             \
  do {                                                                                                       \
    if(nl > 1) hipLaunchKernelGGL((team_mul_kernel<A, LEADV, KSH, true, true>), g, t, 0, ma.stream, kt);     \
    else hipLaunchKernelGGL((team_mul_kernel<A, LEADV, KSH, false, true>), g, t, 0, ma.stream, kt);          \
  } while(0)
      switch(ma.logn - kTeamBlock) {
        case 3: NTT_TEAM_MUL_PTRS(3); break;
        case 4: NTT_TEAM_MUL_PTRS(4); break;
        default: NTT_TEAM_MUL_PTRS(5); break;
      }
#undef NTT_TEAM_MUL_PTRS
      return hipGetLastError();
    }
#define NTT_TEAM_MUL(LEADV)                                                                                  \
  do {                                                                                                       \
    if(nl > 1) hipLaunchKernelGGL((team_mul_kernel<A, LEADV, KSH, true>), g, t, 0, ma.stream, kt);           \
    else hipLaunchKernelGGL((team_mul_kernel<A, LEADV, KSH, false>), g, t, 0, ma.stream, kt);                \
  } while(0)
    switch(ma.logn - kTeamBlock) {
      case 3: NTT_TEAM_MUL(3); break;
      case 4: NTT_TEAM_MUL(4); break;
      default: NTT_TEAM_MUL(5); break;
    }
#undef NTT_TEAM_MUL
    return hipGetLastError();
  }
}

/* N = 2^15 in one pass with the product at the output (onepass_mul_kernel): one persistent 1024-thread workgroup per CU */
template <class A, int KSH> hipError_t launch_onepass_mul(const MulArgs &ma)
{
  if constexpr(!onepass_built<A>()) {
    return hipErrorNotSupported;
  } else {
    if(ma.logn != (uint32_t)kFusedLarge + 1) return hipErrorNotSupported;
    const uint64_t nl = (uint64_t)(ma.nlimbs > 0 ? ma.nlimbs : 1);
    KMul<A> km{};
    km.k.a = ma.a;
    const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(ma.limbs);
    for(uint64_t i = 0; i < nl; i++) km.k.limbs[i] = recs[i];
    km.k.limb_stride = ma.limb_stride;
    km.k.poly_stride = ma.poly_stride ? ma.poly_stride : (1ull << ma.logn);
    km.k.logn        = ma.logn;
    km.k.s0          = 1;
    km.k.nblocks     = ma.batch;
    km.b             = ma.b;
    km.out           = ma.out;
    km.b_limb_stride = ma.b_limb_stride;
    km.lazy_in       = (uint32_t)ma.lazy_in;
    km.b_bcast       = (uint32_t)ma.b_bcast;
    km.accumulate    = (uint32_t)ma.accumulate;
    uint64_t wgs = ma.batch;
    uint64_t cap = (uint64_t)(ma.num_cus > 0 ? ma.num_cus : 256);
    if(ma.max_grid > 0) cap = (uint64_t)ma.max_grid;
    cap = cap / nl > 0 ? cap / nl : 1;
    if(wgs > cap) wgs = cap;
    if(wgs == 0) return hipSuccess;
    km.k.wgs_per_limb = (uint32_t)wgs;
    const dim3 grid((unsigned)wgs, (unsigned)nl);
    if(ma.ptrs) {
      if(nl > 1) return hipErrorNotSupported;
      km.k.ptab = reinterpret_cast<const uint64_t *>(ma.a);
      km.k.a    = reinterpret_cast<uint64_t *>((uintptr_t)ma.ptr_limb_off * 8u);
      hipLaunchKernelGGL((onepass_mul_kernel<A, KSH, false, true>), grid, dim3(1024), 0, ma.stream, km);
      return hipGetLastError();
    }
    if(nl > 1) hipLaunchKernelGGL((onepass_mul_kernel<A, KSH, true>), grid, dim3(1024), 0, ma.stream, km);
    else hipLaunchKernelGGL((onepass_mul_kernel<A, KSH, false>), grid, dim3(1024), 0, ma.stream, km);
    return hipGetLastError();
  }
}

template <class A, int KSH> hipError_t launch_fwd_mul_impl(const MulArgs &ma)
{
  if(ma.nlimbs > kMaxLimbs) return hipErrorInvalidValue;
  if(ma.ptrs && !ma.one_pass && !ma.team_ctl && ma.logn > (uint32_t)kFusedMax) return hipErrorNotSupported;
  if(ma.one_pass) return launch_onepass_mul<A, KSH>(ma);
  if(ma.team_ctl) return launch_team_mul<A, KSH>(ma);
  if(ma.logn > (uint32_t)kFusedMax) {
    if(ma.block_log == (uint32_t)kFusedSmallBlock) return launch_fwd_mul_blocks<A, kFusedSmallBlock, KSH>(ma);
    if(ma.block_log == (uint32_t)kFusedLarge) return launch_fwd_mul_blocks<A, kFusedLarge, KSH>(ma);
    return hipErrorInvalidValue;
  }
  switch(ma.logn) {
#define NTT_MUL_CASE(LN) \
  case LN: return launch_fwd_mul_blocks<A, LN, KSH>(ma);
    NTT_MUL_CASE(6) NTT_MUL_CASE(7) NTT_MUL_CASE(8) NTT_MUL_CASE(9) NTT_MUL_CASE(10) NTT_MUL_CASE(11) NTT_MUL_CASE(12) NTT_MUL_CASE(13)
    NTT_MUL_CASE(14)
#undef NTT_MUL_CASE
    default: return hipErrorNotSupported;
  }
}

#define NTT_DEFINE_LAUNCH_FWD_MUL(A, KSH) \
  template <> hipError_t launch_fwd_mul<A, KSH>(const MulArgs &ma) { return launch_fwd_mul_impl<A, KSH>(ma); }

#define NTT_DEFINE_LAUNCH_DOT(A, KSH) \
  template <> hipError_t launch_dot<A, KSH>(const DotArgs &da) { return launch_dot_impl<A, KSH>(da); }

#define NTT_DEFINE_LAUNCH_PRODUCT(A, KSH) \
  template <> hipError_t launch_product<A, KSH>(const ProdArgs &pa) { return launch_product_impl<A, KSH>(pa); }
/* (a translation unit of its own per policy: inst_team_*.hip) */
#define NTT_DEFINE_LAUNCH_TEAM_PRODUCT(A, KSH) \
  template <> hipError_t launch_team_product<A, KSH>(const ProdArgs &pa) { return launch_team_product_impl<A, KSH>(pa); }

/* body of launch_pass<A,KSH>; each instantiating .hip file expands this once */
#define NTT_DEFINE_LAUNCH_PASS(A, KSH)                                                   \
  template <> hipError_t launch_pass<A, KSH>(const PassArgs &pa)                         \
  {                                                                                      \
    if(pa.fused == 4) return pa.inverse ? launch_onepass<A, true, KSH>(pa) : launch_onepass<A, false, KSH>(pa); \
    if(pa.fused == 3) {                                                                  \
      switch(pa.r) {                                                                     \
        case 3: return pa.inverse ? launch_team<A, 3, true, KSH>(pa) : launch_team<A, 3, false, KSH>(pa); \
        case 4: return pa.inverse ? launch_team<A, 4, true, KSH>(pa) : launch_team<A, 4, false, KSH>(pa); \
        case 5: return pa.inverse ? launch_team<A, 5, true, KSH>(pa) : launch_team<A, 5, false, KSH>(pa); \
        default: return hipErrorInvalidValue;                                            \
      }                                                                                  \
    }                                                                                    \
    if(pa.fused == 2) {                                                                  \
      switch(pa.r) {                                                                     \
        case 1: return pa.inverse ? launch_twophase<A, 1, true, KSH>(pa) : launch_twophase<A, 1, false, KSH>(pa); \
        case 2: return pa.inverse ? launch_twophase<A, 2, true, KSH>(pa) : launch_twophase<A, 2, false, KSH>(pa); \
        case 3: return pa.inverse ? launch_twophase<A, 3, true, KSH>(pa) : launch_twophase<A, 3, false, KSH>(pa); \
        default: return hipErrorInvalidValue;                                            \
      }                                                                                  \
    }                                                                                    \
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

/* the radix-4 formulation (ArithU64R4): block passes, and column passes of one or two radix-4 levels before (forward) or
 * after (inverse) them (ntt_passplan.h: make_passes_r4) */
#define NTT_DEFINE_LAUNCH_PASS_RADIX4(A, KSH)                                            \
  template <> hipError_t launch_pass<A, KSH>(const PassArgs &pa)                         \
  {                                                                                      \
    if(pa.fused == 1) {                                                                  \
      switch(pa.r) {                                                                     \
        NTT_FUSED_CASES(A, KSH)                                                          \
        default: return hipErrorInvalidValue;                                            \
      }                                                                                  \
    }                                                                                    \
    if(pa.fused || pa.s != 0) return hipErrorInvalidValue;                               \
    switch(pa.r) {                                                                       \
      case 2: return pa.inverse ? launch_column<A, 2, true, KSH>(pa) : launch_column<A, 2, false, KSH>(pa); \
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
