/*
 * ntt_kernels_products.h -- the product kernels either side of the path: coefficient-domain products (fused_product_kernel, team_product_kernel),
 * NTT-domain products (dot_inv_kernel, team_dot_kernel), forward transform with the product at its output (fwd_mul_kernel, team_mul_kernel).
 * Part of ntt_kernels.h (included from there, in this order: block, team, products, launch); not a header of its own.
 */
#pragma once

namespace ntt {

/* ------------------------------------------------------------------ */
/* fused product: c = a * b in Z_q[X]/(X^N+1), b never leaves the CU     */
/* ------------------------------------------------------------------ */
/*
 * The caller-side step either side of the path.  a^ = fwd(a) is in HBM (one ordinary forward
 * launch).  This kernel then does, per polynomial and without touching HBM in between:
 *     load b -> forward transform (14 stages) -> times a^ (read once, in the layout the last forward group
 *     already has) -> inverse transform (14 stages, N^-1 folded) -> store c
 * The forward transform ends and the inverse begins in the same thread <-> index layout (runs of four
 * consecutive coefficients per lane), so the product needs no exchange.  HBM traffic of a product: 16N (fwd a)
 * + 24N (this kernel) = 40N bytes instead of 72N for four launches (and 3 -> 2 launches); the kernel itself is
 * bound by its 2 x 14 stages of butterflies, the 24N bytes hide behind them.
 * Twiddles: forward half as fused_kernel (scalar cache / LDS table / registers); the inverse half reads its
 * per-lane group from the SAME LDS table in mirrored order (load_stage_tw MIRROR: w^-1[2^s+j] = -w[2^(s+1)-1-j]),
 * so no second table is needed in LDS; the first inverse group's 12 twiddles are requested while the forward
 * half finishes and reuse the registers of the forward half's last group.
 * Reference primitive this generalises: fast_mul_mod_q (include/internal/fast_mul_operators.h:56-60).
 */
template <class A> struct ProdParams {
  Params<A>              f;      /* forward tables; a = b's coefficients in, nblocks = polynomials */
  const typename A::tw * tw_i;   /* inverse full records (+16 folded N^-1 records) */
  const typename A::ctw *tw8_i;  /* inverse compact */
  const uint64_t *       ahat;   /* fwd(a): canonical, or lazy [0,4q) when a_lazy */
  uint64_t *             out;    /* c; may alias b or ahat */
  uint32_t               a_lazy;
};
/* kernel argument of the product kernels (limb 0's pointers; KArgs::limb_stride separates the limbs of all three slabs) */
template <class A> struct KProd {
  KArgs<A>        f;
  const uint64_t *ahat;
  uint64_t *      out;
  uint32_t        a_lazy;
};
/* PTRS: k.ahat / k.out are device TABLES of polynomial addresses (tab_poly), the same for every limb; the limb's offset travels in
 * pp.f.a (k.f.a + limb * k.f.limb_stride, k.f.a an offset from a null base) */
template <class A, bool MULTI, bool PTRS = false> __device__ __forceinline__ ProdParams<A> limb_prod_params(const KProd<A> &k, uint32_t &bid, uint32_t &gdim)
{
  uint32_t      limb;
  ProdParams<A> pp;
  pp.f = limb_params<A, false, MULTI>(k.f, bid, gdim, limb);
  pp.tw_i   = k.f.limbs[limb].tw_i;
  pp.tw8_i  = k.f.limbs[limb].tw8_i;
  pp.ahat   = PTRS ? k.ahat : k.ahat + (uint64_t)limb * k.f.limb_stride;
  pp.out    = PTRS ? k.out : k.out + (uint64_t)limb * k.f.limb_stride;
  pp.a_lazy = k.a_lazy;
  return pp;
}

/* PTRS forms of the product kernels (round 6): whole polynomials held SEPARATELY -- every operand pointer of the kernel arguments
 * is a DEVICE TABLE of polynomial addresses (a plain array of device pointers, as for the transforms: poly_offset); `limb` carries
 * the words from every table entry to the workgroup's limb as an offset from a null base, where the slab forms carry an operand's
 * base (the MULTI instances serve the limbs of an RNS set in one launch -- limb_params / team_limb add limb * limb_stride to that
 * offset; the one-pass product at 2^15 takes one limb per launch).  UNIFORM as in poly_offset: the polynomial's index is the same for the whole wave (one s_load_dwordx2). */
template <bool UNIFORM> __device__ __forceinline__ uint64_t *tab_poly(const void *tab, uint64_t poly, const uint64_t *limb)
{
  return const_cast<uint64_t *>(limb) + poly_offset<UNIFORM>(poly, 0, reinterpret_cast<const uint64_t *>(tab));
}

/* WHOLE: the block is the whole polynomial (N = 2^14).  !WHOLE: the blocks of a larger transform (N = 2^15..2^17,
 * pp.f.s0 = log2 N - 14 leading stages done by column passes before and after this launch): the product is
 * element-wise, so it fuses block by block just the same -- per limb col(a), blocks(a), col(b), THIS, col^-1(c):
 * 88N bytes instead of 120N.  A workgroup then always sees the same block position (its stride is a multiple of the
 * blocks per polynomial), whose forward table slice it keeps in LDS; the mirrored read would need the slice of the
 * complementary position, so the inverse half takes that one group's twiddles from global memory instead. */
/* BOTH: pp.ahat holds a's COEFFICIENTS (blocks of a larger product: a after its column passes) and the kernel takes them through the forward stages too -- a^
 * waits, as doubles, in the 32 VGPRs that hold the prefetched a^ words otherwise, so the register budget is the same; a^
 * never exists in memory (24N instead of 40N bytes per product, one launch instead of two) and a is left untouched. */
/* PTRS (whole polynomials, BOTH): kp.ahat / kp.f.ptab / kp.out are the tables of a, b and c, kp.f.a = the first limb's offset (MULTI:
 * the limbs of an RNS set kp.f.limb_stride words apart behind every table entry). */
template <class A, int LOGN, int KSH, bool ALAZY, bool WHOLE, bool MULTI = false, bool BOTH = false, bool PTRS = false>
__global__ void __launch_bounds__((Geom<LOGN, false, 3>::WG), (Geom<LOGN, false, 3>::WPS))
  fused_product_kernel(const KProd<A> kp)
{
  static_assert(!PTRS || (WHOLE && BOTH), "pointer tables: coefficient-domain products of whole polynomials");
  /* (!WHOLE && BOTH: the blocks of a larger product; pp.ahat then holds what a's column passes left, as pf.a does for b) */
  uint32_t            bid, gdim;
  const ProdParams<A> pp = limb_prod_params<A, MULTI, PTRS>(kp, bid, gdim);
  using P = Plan<LOGN>;
  using G = Geom<LOGN, false, 3>;
  static_assert(A::kCompact && G::BPW == 1 && (LOGN == 14 || LOGN == 12 || (LOGN == 13 && WHOLE)),
                "built for the FP64 policy on blocks of 2^12 and 2^14 points and whole polynomials of 2^13");
  constexpr uint32_t MASKF = fused_mask<A, LOGN, false, KSH>();
  constexpr uint32_t MASKI = fused_mask<A, LOGN, true, KSH>() | (WHOLE ? kLastInvFlag : 0u);
  constexpr int      GL    = P::NG - 1;
  static_assert(G::TBL(GL - 1) > 0 && G::TBL(GL) == 0, "twiddle placement this kernel assumes");
  __shared__ typename A::val lds_all[P::LDS_ELEMS + G::LDS_TW];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + P::LDS_ELEMS);
  const lds_ctw_ptr<A>   ltw  = (lds_ctw_ptr<A>)tabl;
  const uint32_t         tid  = threadIdx.x;
  /* whole polynomials: the stage offset and size are compile-time constants (as run-time kernel arguments they
   * cost scalar registers the two halves do not have) */
  Params<A> pf = pp.f;
  if constexpr(WHOLE) {
    pf.s0   = 0;
    pf.logn = LOGN;
  }
  pf.wide      = 0;
  pf.lazy      = 0;
  Params<A> pi = pf;
  pi.tw                       = pp.tw_i;
  pi.tw8                      = pp.tw8_i;
  pi.lastinv                  = WHOLE ? 1 : 0;
  const uint64_t stride = gdim;
  uint64_t       b      = bid;
  if(b >= pf.nblocks) return;
  /* (the loop's comparisons as scalar subtractions where the two VGPRs of a vector comparison are the ones that spill: `below`) */
  constexpr bool SCMP = BOTH && A::kWide52 && LOGN == 14;
  const uint32_t blk = WHOLE ? 0u : ((uint32_t)b & ((1u << pf.s0) - 1u)); /* the same for every block of this workgroup */
  fill_lds_tables<A, LOGN, false, G>(tabl, pf, blk, tid);
  __syncthreads();
  /* where block bb of the three operands starts */
  const auto at_a = [&](uint64_t bb) -> const uint64_t * {
    if constexpr(PTRS) return tab_poly<true>(pp.ahat, bb, pf.a);
    else return pp.ahat + blk_off<LOGN>(pf, bb);
  };
  const auto at_b = [&](uint64_t bb) -> const uint64_t * {
    if constexpr(PTRS) return tab_poly<true>(pf.ptab, bb, pf.a);
    else return pf.a + blk_off<LOGN>(pf, bb);
  };
  const auto at_c = [&](uint64_t bb) -> uint64_t * {
    if constexpr(PTRS) return tab_poly<true>(pp.out, bb, pf.a);
    else return pp.out + blk_off<LOGN>(pf, bb);
  };
  uint64_t raw[kE];
  prefetch_first<LOGN>(raw, tid, BOTH ? at_a(b) : at_b(b));
  pin_raw(raw);
  for(; SCMP ? below(b, pf.nblocks) : b < pf.nblocks; b += stride) {
    /* The two sets of 12 per-lane twiddles (forward half's last group, inverse half's first group) share one set
     * of registers and are therefore requested per block.  They do not depend on the block, so the
     * compiler would hoist both sets (and their 24 lane offsets) out of the loop and spill; the opaque copy of
     * the thread id ties them -- and every other lane-dependent address of the two halves (six exchanges, two
     * prefetches, the stores): hoisted, those were spilled and reloaded from scratch behind the HBM prefetch --
     * to the iteration. */
    uint32_t tl = tid;
    asm volatile("" : "+v"(tl));
    /* (PTRS: the table entries this iteration needs -- b and c of this block, a of the next one -- are read here, scalar loads a
     * whole half ahead of their use.  a's entry read where the next block is requested, between the two halves, cost 31 spilled
     * VGPRs: the inverse half's per-lane twiddles went to scratch) */
    const uint64_t *pb_cur = nullptr;
    uint64_t *      pc_cur = nullptr;
    const uint64_t *pa_nxt = nullptr;
    /* (the slab forms gain nothing from the same hoist: 120 instead of 128 VGPRs, 1-2 % slower -- profiles/r06/ab_product_hoist.txt) */
    constexpr bool HOIST = PTRS;
    if constexpr(HOIST) {
      pb_cur = at_b(b);
      pc_cur = at_c(b);
      const bool more0 = SCMP ? below(b + stride, pf.nblocks) : b + stride < pf.nblocks;
      pa_nxt           = at_a(more0 ? b + stride : b);
    }
    typename A::ctw pre[4][kE / 2];
    preload_group_tw<A, LOGN, GL>(pre, tl, blk, pf);
    /* (52-bit class, BOTH at 2^14: b's words are requested behind a's first stage group instead of in front of it -- the
     * reduce-both-operands butterflies of that group need the registers: 2 spilled VGPRs otherwise) */
    constexpr bool LATE_B = BOTH && A::kWide52 && LOGN == 14;
    const auto forward = [&](typename A::val(&v)[kE], auto late) {
      run_group<A, LOGN, 0, false, MASKF>(v, tl, blk, pf);
      if constexpr(decltype(late)::value) {
        __builtin_amdgcn_sched_barrier(0);
        prefetch_first<LOGN>(raw, tl, HOIST ? pb_cur : at_b(b));
      }
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        exchange<A, LOGN, GI, GI + 1>(v, tl, lds_all);
        if constexpr(GI + 1 == GL) {
          run_group_preloaded<A, LOGN, GL, MASKF>(v, pre, pf);
        } else if constexpr(G::TBL(GI + 1) > 0) {
          run_group<A, LOGN, GI + 1, false, MASKF, true>(v, tl, blk, pf, ltw + G::TBL_OFF(GI + 1));
        } else {
          run_group<A, LOGN, GI + 1, false, MASKF>(v, tl, blk, pf);
        }
      });
    };
    typename A::val x[kE];
    typename A::val xa[BOTH ? kE : 1];
    if constexpr(BOTH) {
      convert_inputs<A, false>(xa, raw, false, pf.c);
      if constexpr(!LATE_B) prefetch_first<LOGN>(raw, tl, HOIST ? pb_cur : at_b(b)); /* b's words travel during a's forward stages */
      forward(xa, std::integral_constant<bool, LATE_B>{});
      /* (b's words are converted after a's last stage, not before: interleaved by the scheduler, x, xa and the raw words
       * lived side by side and spilled) */
      __builtin_amdgcn_sched_barrier(0);
      convert_inputs<A, false>(x, raw, false, pf.c);
    } else {
      convert_inputs<A, false>(x, raw, false, pf.c);
      /* a^ in the last group's layout: requested now, used after the 14 forward stages */
      prefetch_last<LOGN>(raw, tl, at_a(b));
    }
    forward(x, std::false_type{});
    /* the inverse's first group: its twiddles land while the product is computed */
    asm volatile("" : "+v"(tl));
    preload_group_tw<A, LOGN, GL>(pre, tl, blk, pi);
    if constexpr(BOTH) {
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::product_rr(x[decltype(ee)::value], xa[decltype(ee)::value], pf.c); });
    } else {
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::template product_in_domain<ALAZY>(x[decltype(ee)::value], raw[decltype(ee)::value], pf.c); });
    }
    /* the next block's loads reuse a^'s registers: not before the last product has read them (interleaved by
     * the scheduler, the two lived side by side and spilled) */
    __builtin_amdgcn_sched_barrier(0);
    {
      const bool     more = SCMP ? below(b + stride, pf.nblocks) : b + stride < pf.nblocks;
      const uint64_t nb   = more ? b + stride : b;
      prefetch_first<LOGN>(raw, tl, HOIST ? pa_nxt : (BOTH ? at_a(nb) : at_b(nb)), more);
    }
    run_group_preloaded<A, LOGN, GL, MASKI, true>(x, pre, pi);
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int GI = P::NG - 1 - decltype(gg)::value;
      exchange<A, LOGN, GI, GI - 1>(x, tl, lds_all);
      if constexpr(WHOLE && G::TBL(GI - 1) > 0) {
        run_group<A, LOGN, GI - 1, true, MASKI, true, true>(x, tl, blk, pi, ltw + G::TBL_OFF(GI - 1));
      } else {
        /* (per-lane twiddles from global memory in the !WHOLE form: keep their requests behind the exchange --
         * hoisted above it by the scheduler they occupied 30 registers during the previous group and spilled) */
        if constexpr(!WHOLE && G::TBL(GI - 1) > 0) __builtin_amdgcn_sched_barrier(0);
        run_group<A, LOGN, GI - 1, true, MASKI>(x, tl, blk, pi);
      }
    });
    uint64_t out[kE];
    static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = A::store_inv(x[decltype(ee)::value], pf.c); });
    buffer_store_first_raw<LOGN>(out, tl, HOIST ? pc_cur : at_c(b));
  }
}

/* ------------------------------------------------------------------ */
/* products at N = 2^15 .. 2^17: all passes as items of one launch      */
/* ------------------------------------------------------------------ */
/*
 * c = a * b with a^ = fwd(a) already in HBM: the remaining chain -- column stages of b, per block forward x a^ -> inverse,
 * inverse column stages of c -- as the three item kinds of ONE launch in team_kernel's scheme (per-XCD in-order queues,
 * per-polynomial hand-off counters, intermediates kept in the XCD's L2 / Infinity Cache): first-pass items of polynomial j,
 * second-pass items of polynomial j - lag, third-pass items of polynomial j - 2 lag.  An item only ever waits for items
 * handed out earlier in its queue, and first-pass items never wait: no deadlock whatever the residency.  Five launches per
 * 256 MiB chunk become one launch per batch; the fabric carries 40N bytes for this chain instead of 56N while the
 * intermediates stay on chip.  The blocks are 2^12 points at every size (2^17: five column stages in one item, where the
 * per-pass path needs 2^14-point blocks to get by with one column launch).
 * c may alias a or b exactly as in fused_product_kernel: a block's a^ and b words are read by the item that overwrites them.
 */
template <class A, int KSH, int LDAUX_B, int LDAUX_A, int STAUX>
__device__ __forceinline__ void team_product_item(uint64_t *bblk, const uint64_t *ablk, uint64_t *cblk, uint32_t blk, uint32_t tid0,
                                                  const Params<A> &pf, const Params<A> &pi, typename A::val *lds, typename A::ctw *tabl)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, 3>;
  constexpr uint32_t MASKF = fused_mask<A, LOGN, false, KSH>();
  constexpr uint32_t MASKI = fused_mask<A, LOGN, true, KSH>();
  constexpr int      GL    = P::NG - 1;
  static_assert(G::TBL(GL - 1) > 0 && G::TBL(GL) == 0 && !P::WAVE_LOCAL(0, 1), "twiddle placement / barrier this item assumes");
  const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  /* (an opaque copy of the thread id ties every lane-dependent address of the item to the item: hoisted out of the item
   * loop they live in registers -- or scratch -- for the whole launch; fused_product_kernel does the same) */
  uint32_t tl = tid0;
  asm volatile("" : "+v"(tl));
  const uint32_t tid = tl;
  uint64_t raw[kE];
  prefetch_first<LOGN, LDAUX_B>(raw, tid, bblk);
  typename A::ctw pre[4][kE / 2];
  preload_group_tw<A, LOGN, GL>(pre, tid, blk, pf);
  fill_lds_tables<A, LOGN, false, G>(tabl, pf, blk, tid); /* (published by the first exchange's barriers) */
  typename A::val x[kE];
  convert_inputs<A, false>(x, raw, false, pf.c);
  prefetch_last<LOGN, LDAUX_A>(raw, tid, ablk); /* a^ in the last group's layout: used after the twelve forward stages */
  run_group<A, LOGN, 0, false, MASKF>(x, tid, blk, pf);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = decltype(gg)::value;
    exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
    if constexpr(GI + 1 == GL) {
      run_group_preloaded<A, LOGN, GL, MASKF>(x, pre, pf);
    } else if constexpr(G::TBL(GI + 1) > 0) {
      run_group<A, LOGN, GI + 1, false, MASKF, true>(x, tid, blk, pf, ltw + G::TBL_OFF(GI + 1));
    } else {
      run_group<A, LOGN, GI + 1, false, MASKF>(x, tid, blk, pf);
    }
  });
  uint32_t t2 = tid;
  asm volatile("" : "+v"(t2));
  preload_group_tw<A, LOGN, GL>(pre, t2, blk, pi); /* the inverse's first group: lands while the product is computed */
  static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::template product_in_domain<true>(x[decltype(ee)::value], raw[decltype(ee)::value], pf.c); });
  run_group_preloaded<A, LOGN, GL, MASKI, true>(x, pre, pi);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    exchange<A, LOGN, GI, GI - 1>(x, tid, lds);
    if constexpr(G::TBL(GI - 1) > 0) __builtin_amdgcn_sched_barrier(0); /* (as in fused_product_kernel: keep the global twiddle requests behind the exchange) */
    run_group<A, LOGN, GI - 1, true, MASKI>(x, tid, blk, pi);
  });
  uint64_t out[kE];
  static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = A::store_inv(x[decltype(ee)::value], pf.c); });
  buffer_store_first_raw<LOGN, STAUX>(out, tid, cblk);
}

/* The same item with BOTH forward transforms inside (team_product_kernel<..., FOUR = true>): a's block comes in as the
 * intermediate of a's column pass, is taken through the twelve block stages first and waits in 32 VGPRs -- the registers
 * that hold the prefetched a^ words in the item above -- while b's block follows; a^ never exists in memory: 16N bytes
 * fewer across the fabric per product (no write-through store of a^, no read of it) and one launch less. */
/* (a block of a separately held polynomial, its table entry read where the block is first touched -- a pointer read at the top of the
 * item would sit in scalar registers through all 36 stages: two spilled VGPRs in the 128-VGPR kernels, whose scalar spills live in
 * vector lanes) */
struct TabBlock {
  const void *    tab;
  const uint64_t *limb;
  uint32_t        poly;
  uint32_t        off; /* words from the polynomial's start to the block */
};
__device__ __forceinline__ uint64_t *blk_ptr(uint64_t *p) { return p; }
__device__ __forceinline__ const uint64_t *blk_ptr(const uint64_t *p) { return p; }
__device__ __forceinline__ uint64_t *blk_ptr(const TabBlock &t) { return tab_poly<true>(t.tab, t.poly, t.limb) + t.off; }

template <class A, int KSH, int LDAUX, int STAUX, class BB = uint64_t *, class AB = const uint64_t *, class CB = uint64_t *>
__device__ __forceinline__ void team_product_item2(BB bblk, AB ablk, CB cblk, uint32_t blk, uint32_t tid0,
                                                   const Params<A> &pf, const Params<A> &pi, typename A::val *lds, typename A::ctw *tabl)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, 3>;
  constexpr uint32_t MASKF = fused_mask<A, LOGN, false, KSH>();
  constexpr uint32_t MASKI = fused_mask<A, LOGN, true, KSH>();
  constexpr int      GL    = P::NG - 1;
  static_assert(G::TBL(GL - 1) > 0 && G::TBL(GL) == 0 && !P::WAVE_LOCAL(0, 1), "twiddle placement / barrier this item assumes");
  const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  uint32_t tl = tid0;
  asm volatile("" : "+v"(tl));
  const uint32_t tid = tl;
  uint64_t raw[kE];
  prefetch_first<LOGN, LDAUX>(raw, tid, blk_ptr(ablk));
  typename A::ctw pre[4][kE / 2];
  preload_group_tw<A, LOGN, GL>(pre, tid, blk, pf);
  fill_lds_tables<A, LOGN, false, G>(tabl, pf, blk, tid); /* (published by the first exchange's barriers) */
  const auto forward = [&](typename A::val(&x)[kE]) {
    run_group<A, LOGN, 0, false, MASKF>(x, tid, blk, pf);
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int GI = decltype(gg)::value;
      exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
      if constexpr(GI + 1 == GL) {
        run_group_preloaded<A, LOGN, GL, MASKF>(x, pre, pf);
      } else if constexpr(G::TBL(GI + 1) > 0) {
        run_group<A, LOGN, GI + 1, false, MASKF, true>(x, tid, blk, pf, ltw + G::TBL_OFF(GI + 1));
      } else {
        run_group<A, LOGN, GI + 1, false, MASKF>(x, tid, blk, pf);
      }
    });
  };
  typename A::val xa[kE];
  convert_inputs<A, false>(xa, raw, false, pf.c);
  prefetch_first<LOGN, LDAUX>(raw, tid, blk_ptr(bblk)); /* b's words travel during a's twelve stages */
  forward(xa);
  typename A::val x[kE];
  convert_inputs<A, false>(x, raw, false, pf.c);
  forward(x);
  uint32_t t2 = tid;
  asm volatile("" : "+v"(t2));
  preload_group_tw<A, LOGN, GL>(pre, t2, blk, pi); /* the inverse's first group: lands while the product is computed */
  static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::product_rr(x[decltype(ee)::value], xa[decltype(ee)::value], pf.c); });
  run_group_preloaded<A, LOGN, GL, MASKI, true>(x, pre, pi);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    exchange<A, LOGN, GI, GI - 1>(x, tid, lds);
    if constexpr(G::TBL(GI - 1) > 0) __builtin_amdgcn_sched_barrier(0);
    run_group<A, LOGN, GI - 1, true, MASKI>(x, tid, blk, pi);
  });
  uint64_t out[kE];
  static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = A::store_inv(x[decltype(ee)::value], pf.c); });
  buffer_store_first_raw<LOGN, STAUX>(out, tid, blk_ptr(cblk));
}

struct TeamProdCtl {
  unsigned next[8][32];
  unsigned owner[8][32];
  unsigned done[1]; /* [2][polynomials]: first- and second-pass items finished */
};

template <class A> struct KTeamProd {
  KProd<A>     k;      /* f.a = b, ahat, out = c (limb 0's slabs; f.limb_stride apart); f.nblocks = polynomials PER LIMB */
  TeamProdCtl *ctl;    /* zeroed before the launch */
  uint32_t     lag;
  uint32_t     nlimbs; /* MULTI kernels: limbs of the launch */
  uint64_t     split_rcp;  /* as KTeam::split_rcp */
  uint32_t     poly_major; /* as KTeam::poly_major */
};

/* PTRS: kt.k.f.ptab / kt.k.ahat / kt.k.out are the device tables of b, a and c (tab_poly), kt.k.f.a = the first limb's offset; MULTI: the
 * item's limb kt.k.f.limb_stride words further behind every table entry (loff) */
template <class A, int LEAD, int KSH, bool FOUR = false, bool MULTI = false, bool PTRS = false>
__global__ void __launch_bounds__(256, 4) team_product_kernel(const KTeamProd<A> kt)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, 3>;
  static_assert(A::kCompact && P::T == kTeamCols && LEAD >= 3 && LEAD <= 5, "FP64 policies, N = 2^15..2^17");
  __shared__ typename A::val lds[P::LDS_ELEMS + G::LDS_TW];
  __shared__ unsigned        s_k, s_k2[2];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds + P::LDS_ELEMS);
  const uint32_t         tid  = threadIdx.x;
  uint32_t               bid_, gdim_;
  const ProdParams<A>    pp = limb_prod_params<A, false>(kt.k, bid_, gdim_);
  Params<A>              pf = pp.f;
  pf.s0   = LEAD;
  pf.wide = 0;
  pf.lazy = 0;
  Params<A> pi = pf;
  pi.tw        = pp.tw_i;
  pi.tw8       = pp.tw8_i;
  pi.lastinv   = 1;
  constexpr uint32_t CMASKF = column_mask<A, LEAD, false, KSH>();
  constexpr uint32_t CMASKI = column_mask<A, LEAD, true, KSH>();
  constexpr bool     MID_LAZY = !A::kTracksBounds;
  const uint32_t logn  = LOGN + LEAD;
  const uint32_t batch = (uint32_t)pf.nblocks; /* polynomials per limb */
  const uint32_t total = MULTI ? batch * kt.nlimbs : batch;
  constexpr uint32_t NCOL = 1u << (LOGN - 8), NROW = 1u << LEAD;
  /* FOUR: the first pass takes the column tiles of BOTH operands (b's, then a's) and the product item transforms both blocks */
  constexpr uint32_t NFIRST = FOUR ? 2u * NCOL : NCOL;
  TeamProdCtl *const ctl = kt.ctl;
  const uint32_t lag = kt.lag;
  const uint32_t my  = xcc_id();
  uint64_t       loff = 0; /* the item's limb: word offset of its slabs (MULTI) */
  for(uint32_t qq = 0; qq < 8; qq++) {
    const uint32_t q = (my + qq) & 7u;
    if(tid == 0) {
      const unsigned prev = atomicCAS(&ctl->owner[q][0], 0u, my + 1u);
      s_k                 = (prev == 0u || prev == my + 1u) ? 1u : 0u;
    }
    __syncthreads();
    const bool mine = s_k != 0;
    __syncthreads();
    if(!mine) continue;
    constexpr uint32_t kNoSignal = 0xffffffffu;
    uint32_t           sig       = kNoSignal; /* index into done[]: polynomial, or total + polynomial for the second pass */
    for(uint32_t it = 0;; it ^= 1u) {
      /* (lane-0 blocks are followed at once by a workgroup barrier: see team_kernel) */
      if(tid == 0) {
        if(sig != kNoSignal) __hip_atomic_fetch_add(&ctl->done[sig], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_k2[it] = atomicAdd(&ctl->next[q][0], 1u);
      }
      sig = kNoSignal;
      __syncthreads();
      const TeamItem ti = team_decode(uniform_u32(s_k2[it]), q, total, lag, NFIRST, NROW, NCOL);
      if(ti.stop) break;
      if(!ti.valid) continue;
      const uint32_t pass = ti.pass, item = ti.item, pidx = ti.v;
      uint32_t       pl   = pidx; /* the polynomial inside its limb */
      if constexpr(MULTI) {
        uint32_t limb;
        team_split(pidx, batch, kt.nlimbs, kt.poly_major, kt.split_rcp, limb, pl);
        loff                = (uint64_t)limb * kt.k.f.limb_stride;
        const LimbRec<A> &r = kt.k.f.limbs[limb];
        pf.tw  = r.tw_f;
        pf.tw8 = r.tw8_f;
        pf.c   = r.c;
        pi.tw  = r.tw_i;
        pi.tw8 = r.tw8_i;
        pi.c   = r.c;
      }
      if(pass > 0) {
        const uint32_t need = pass == 1 ? NFIRST : NROW;
        const uint32_t slot = pass == 1 ? pidx : total + pidx;
        if(tid == 0) {
          while(__hip_atomic_load(&ctl->done[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
      }
      const uint64_t  poff  = loff + (uint64_t)pl * pf.pstride;
      uint64_t *      bpoly = pf.a + poff;
      const uint64_t *apoly = pp.ahat + poff;
      uint64_t *      cpoly = pp.out + poff;
      const uint64_t *const plimb = pf.a + loff; /* PTRS: the item's limb as an offset from the null base (MULTI off: loff == 0) */
      if constexpr(PTRS) {
        /* (the column items' one pointer here; the block products read their entries where they touch the blocks: TabBlock) */
        if(pass == 0) {
          bpoly = tab_poly<true>(pf.ptab, pl, plimb);
          apoly = tab_poly<true>(pp.ahat, pl, plimb);
        }
        if(pass == 2) cpoly = tab_poly<true>(pp.out, pl, plimb);
      }
      /* NTT_TEAMPROD_ONLY (diagnostic builds, like NTT_STAMPS; never in the shipped library): which item types do their work -- bit 0
       * b's column items, 3 a's column items, 1 the block products, 2 c's inverse column items; the others only run the queue protocol.
       * Wrong results; one --pmc pass per build gives an item type's FETCH / WRITE bytes by themselves (profiles/r06/config5_bytes_by_item.txt) */
#ifndef NTT_TEAMPROD_ONLY
#  define NTT_TEAMPROD_ONLY 15
#endif
      constexpr uint32_t kOnly = NTT_TEAMPROD_ONLY;
      if(pass == 0) {
        uint64_t *const src = FOUR && item >= NCOL ? const_cast<uint64_t *>(apoly) : bpoly; /* (a is an operand buffer of the caller's: written here) */
        if((kOnly & 9u) == 9u || (kOnly & (FOUR && item >= NCOL ? 8u : 1u)) != 0)
          team_column_item<A, LEAD, false, CMASKF, kAuxSc0Sc1, 0>(src, (item & (NCOL - 1u)) * kTeamCols + tid, logn, pf, MID_LAZY);
      } else if((kOnly & (pass == 1 ? 2u : 4u)) == 0) {
        /* (switched off in this diagnostic build) */
      } else if(pass == 1) {
        if constexpr(FOUR) {
          /* (NTT_TEAMPROD_FAKEBLK, diagnostic builds: every block product reads block 0's twiddles -- wrong results, the same loads,
           * all of them L2-hot: what the twiddle reads of the product items cost at the fabric) */
#ifndef NTT_TEAMPROD_FAKEBLK
#  define NTT_TEAMPROD_FAKEBLK 0
#endif
          if constexpr(PTRS) {
            const uint32_t ioff = item << LOGN;
            team_product_item2<A, KSH, kAuxNt, 0>(TabBlock{pf.ptab, plimb, pl, ioff}, TabBlock{pp.ahat, plimb, pl, ioff}, TabBlock{pp.out, plimb, pl, ioff}, item,
                                                  tid, pf, pi, lds, tabl);
          } else {
            team_product_item2<A, KSH, kAuxNt, 0>(bpoly + ((uint64_t)item << LOGN), apoly + ((uint64_t)item << LOGN),
                                                  cpoly + ((uint64_t)item << LOGN), NTT_TEAMPROD_FAKEBLK ? 0u : item, tid, pf, pi, lds, tabl);
          }
        } else {
          team_product_item<A, KSH, kAuxNt, kAuxNt, 0>(bpoly + ((uint64_t)item << LOGN), apoly + ((uint64_t)item << LOGN),
                                                         cpoly + ((uint64_t)item << LOGN), item, tid, pf, pi, lds, tabl);
        }
      } else {
        team_column_item<A, LEAD, true, CMASKI, kAuxNt, kAuxSc1>(cpoly, item * kTeamCols + tid, logn, pi, false);
      }
      if(pass < 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sig = pass == 0 ? pidx : total + pidx;
      }
    }
  }
}

/* The same product for whole polynomials of 2^8 .. 2^11 points, where a 256-thread workgroup holds several blocks
 * (Geom::BPW) that share the LDS twiddle tables: the plain (non-persistent) loop of fused_kernel's small-block path with
 * the product and the inverse half appended.  Every per-lane group has its forward table in LDS at these sizes, and the
 * inverse half reads all of them mirrored, so the kernel issues no per-lane global twiddle load at all; a^ arrives in
 * the last group's layout as 16-byte loads.  40N bytes per product instead of 72N. */
template <class A, int LOGN, int KSH, bool MULTI = false, bool BOTH = false, bool PTRS = false>
__global__ void __launch_bounds__((Geom<LOGN, false, 3>::WG), (Geom<LOGN, false, 3>::WPS))
  fused_product_small_kernel(const KProd<A> kp)
{
  static_assert(!PTRS || BOTH, "pointer tables: coefficient-domain products (see fused_product_kernel)");
  uint32_t            bid, gdim;
  const ProdParams<A> pp = limb_prod_params<A, MULTI, PTRS>(kp, bid, gdim);
  using P = Plan<LOGN>;
  using G = Geom<LOGN, false, 3>;
  static_assert(A::kCompact && G::BPW > 1 && LOGN >= 8 && LOGN <= 11, "whole polynomials of 2^8..2^11 points");
  constexpr uint32_t MASKF = fused_mask<A, LOGN, false, KSH>();
  constexpr uint32_t MASKI = fused_mask<A, LOGN, true, KSH>() | kLastInvFlag;
  __shared__ typename A::val lds_all[G::BPW * P::LDS_ELEMS + G::LDS_TW];
  const uint32_t         tid = threadIdx.x;
  const uint32_t         sub = tid >> P::LT;
  const uint32_t         t   = tid & (P::T - 1);
  typename A::val *const lds = lds_all + sub * P::LDS_ELEMS;
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
  const lds_ctw_ptr<A>   gtw  = (lds_ctw_ptr<A>)tabl;
  Params<A> pf = pp.f;
  pf.s0        = 0;
  pf.logn      = LOGN;
  pf.wide      = 0;
  pf.lazy      = 0;
  Params<A> pi = pf;
  pi.tw        = pp.tw_i;
  pi.tw8       = pp.tw8_i;
  pi.lastinv   = 1;
  fill_lds_tables<A, LOGN, false, G>(tabl, pf, 0u, tid);
  __syncthreads();
  for(uint64_t b0 = (uint64_t)bid * G::BPW; b0 < pf.nblocks; b0 += (uint64_t)gdim * G::BPW) {
    uint64_t   b    = b0 + sub;
    const bool live = b < pf.nblocks;
    if(!live) b = pf.nblocks - 1; /* idle sub-blocks shadow a real polynomial (barriers are workgroup-wide), never store */
    /* (PTRS: the sub-block's polynomial differs between the waves of the workgroup -- a per-lane table read) */
    const uint64_t *ablk = PTRS ? tab_poly<false>(pp.ahat, b, pf.a) : pp.ahat + blk_off<LOGN>(pf, b);
    uint64_t *      bblk = PTRS ? tab_poly<false>(pf.ptab, b, pf.a) : pf.a + blk_off<LOGN>(pf, b);
    uint64_t *      cblk = PTRS ? tab_poly<false>(pp.out, b, pf.a) : pp.out + blk_off<LOGN>(pf, b);
    const auto forward = [&](typename A::val(&v)[kE]) {
      run_group<A, LOGN, 0, false, MASKF, (G::TBL(0) > 0)>(v, t, 0u, pf, gtw);
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        exchange<A, LOGN, GI, GI + 1>(v, t, lds);
        run_group<A, LOGN, GI + 1, false, MASKF, (G::TBL(GI + 1) > 0)>(v, t, 0u, pf, gtw + G::TBL_OFF(GI + 1));
      });
    };
    typename A::val x[kE];
    uint64_t        raw[kE];
    if constexpr(BOTH) {
      /* a's coefficients through the same forward stages first; a^ waits in registers (the ones a^'s words occupy otherwise) */
      typename A::val xa[kE];
      global_load_first<A, LOGN, false>(xa, t, ablk, false, pf.c);
      if constexpr(PTRS) {
        /* (per-lane addresses: plain loads -- a buffer descriptor must be wave-uniform) */
        static_for<0, kE>([&](auto ee) { raw[decltype(ee)::value] = stream_load(coef_at(bblk + ((uint32_t) decltype(ee)::value << P::LT), t)); });
      } else {
        prefetch_first<LOGN>(raw, t, bblk);
      }
      forward(xa);
      convert_inputs<A, false>(x, raw, false, pf.c);
      forward(x);
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::product_rr(x[decltype(ee)::value], xa[decltype(ee)::value], pf.c); });
    } else {
      global_load_first<A, LOGN, false>(x, t, bblk, false, pf.c);
      prefetch_last<LOGN>(raw, t, ablk);
      forward(x);
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::template product_in_domain<true>(x[decltype(ee)::value], raw[decltype(ee)::value], pf.c); });
    }
    constexpr int GL = P::NG - 1;
    run_group<A, LOGN, GL, true, MASKI, (G::TBL(GL) > 0), (G::TBL(GL) > 0)>(x, t, 0u, pi, gtw + G::TBL_OFF(GL));
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int GI = P::NG - 1 - decltype(gg)::value;
      exchange<A, LOGN, GI, GI - 1>(x, t, lds);
      run_group<A, LOGN, GI - 1, true, MASKI, (G::TBL(GI - 1) > 0), (G::TBL(GI - 1) > 0)>(x, t, 0u, pi, gtw + G::TBL_OFF(GI - 1));
    });
    if(live) global_store_first<A, LOGN, true>(x, t, cblk, pf.c, false);
  }
}

/* ------------------------------------------------------------------ */
/* products of operands that ARE in the NTT domain                      */
/* ------------------------------------------------------------------ */
/*
 * c = inv( sum_{i<k} a_i^ (.) b_i^ ): the other half of SURVEY 8(f) f1 ("fusing the multiply into the inverse's first load
 * saves 16N bytes").  Keys, ciphertexts and plaintexts of an FHE caller live in the NTT domain; what it issues is the
 * pointwise product of two transformed operands (k = 1) or the inner product of a digit-decomposed ciphertext with a
 * key (key switching, k = 2 .. tens) followed by ONE inverse transform.  This kernel is the inverse block kernel with its
 * input conversion replaced: where fused_kernel<.., INV> turns the 16 words of a thread into values, this one reads the 16
 * words of every a_i^ and b_i^ in the same layout (the product is element-wise, so the inverse's first group's layout
 * serves), forms the k products and their sum in registers (A::dot_term / dot_acc / dot_fold) and runs the inverse stages
 * on the sum.  HBM traffic: 16kN bytes in, 8N out -- 24N for a plain product instead of 40N for pointwise + inverse, no
 * intermediate ever written; with B_BCAST the b_i^ are ONE polynomial each, shared by the batch (a key: read from the
 * L2), and the traffic is 8kN + 8N.  Blocks of a larger transform (LASTINV = false: N > 2^14, the column stages of the
 * inverse follow as launches of their own) work the same way: the product rides in the first pass of the inverse.
 * Reference primitive this generalises: fast_mul_mod_q (include/internal/fast_mul_operators.h:56-60).
 */
constexpr int kMaxDot = 32; /* operand pairs of one launch (2 x 32 pointers in the kernel arguments) */

template <class A> struct KDot {
  KArgs<A>        k;             /* k.a = c (limb 0), nblocks / s0 / logn as for an inverse block pass */
  uint32_t        npairs;        /* 1 .. kMaxDot */
  uint32_t        lazy_in;       /* operand words may be lazy: anywhere in [0,4q) */
  uint32_t        b_bcast;       /* every b_i^ is one polynomial per limb, shared by the whole batch */
  uint64_t        b_limb_stride; /* words between consecutive limbs of a b operand */
  const uint64_t *a[kMaxDot];
  const uint64_t *b[kMaxDot];
};

/* The b operand of a pair goes through the caches (no nt hint): when ONE polynomial serves the whole batch (a key) every
 * block re-reads it and it must stay in the L2 -- with nt loads the broadcast form measured no faster than the
 * per-polynomial one (profiles/r04/domain_bench_first.txt).  The hint is an instruction bit, and a run-time branch between
 * two sets of loads makes the register allocator keep both sets apart (77 spilled VGPRs): one policy for both forms. */
constexpr int kDotAuxB = 0;
/* tuning knobs of the persistent loop (A/B builds: tools/build_tu_variant.sh) */
#ifndef NTT_DOT_AUX_A
#  define NTT_DOT_AUX_A 0 /* cache policy of the a operand's loads: plain, like b's (measured +3 % over nt at k = 1, +10 % with a broadcast key at k = 8: profiles/r04/ab_dot.txt) */
#endif
#ifndef NTT_DOT_A_AT
#  define NTT_DOT_A_AT 1 /* the next block's a words are requested behind the exchange into this group (1 = the last one) */
#endif
#ifndef NTT_DOT_LOOP_CHUNK
#  define NTT_DOT_LOOP_CHUNK 2 /* products in flight inside the pair loop */
#endif
template <int LOGN> __device__ __forceinline__ void prefetch_last_b(uint64_t (&raw)[kE], uint32_t t, const uint64_t *blk, bool live = true)
{
  prefetch_last<LOGN, kDotAuxB>(raw, t, blk, live);
}

/* PTRS (whole polynomials): kd.a[i], kd.b[i] (unless broadcast: then the polynomial itself, as ever) and kd.k.ptab are the tables of
 * a_i^, b_i^ and c; kd.k.a = the first limb's offset (tab_poly), the limbs of an RNS set kd.k.limb_stride words apart behind every
 * table entry (MULTI: limb_params adds them). */
template <class A, int LOGN, int KSH, bool LASTINV, bool MULTI = false, bool PTRS = false>
__global__ void __launch_bounds__((Geom<LOGN, true, flavor_of<A>()>::WG), (Geom<LOGN, true, flavor_of<A>()>::WPS)) dot_inv_kernel(const KDot<A> kd)
{
  static_assert(!PTRS || LASTINV, "pointer tables: whole polynomials");
  uint32_t        bid, gdim, limb;
  const Params<A> p = limb_params<A, true, MULTI>(kd.k, bid, gdim, limb);
  using P = Plan<LOGN>;
  using G = Geom<LOGN, true, flavor_of<A>()>;
  constexpr uint32_t MASK   = fused_mask<A, LOGN, true, KSH>() | (LASTINV ? kLastInvFlag : 0u);
  constexpr int      LDS_TW = G::LDS_TW;
  __shared__ typename A::val lds_all[G::BPW * P::LDS_ELEMS + LDS_TW];
  const uint32_t   tid   = threadIdx.x;
  const uint32_t   sub   = tid >> P::LT;
  const uint32_t   t     = tid & (P::T - 1);
  typename A::val *lds   = lds_all + sub * P::LDS_ELEMS;
  const uint32_t   bmask = (1u << p.s0) - 1u;
  const uint32_t   np    = kd.npairs;
  const bool       lazy  = kd.lazy_in != 0;
  const bool       bc    = kd.b_bcast != 0;
  const uint64_t   aoff  = (uint64_t)limb * kd.k.limb_stride; /* (MULTI off: limb == 0, both offsets fold away) */
  const uint64_t   boff  = (uint64_t)limb * kd.b_limb_stride;
  /* where block bb of operand pair i and of c starts (UNI: bb is the same for the whole wave) */
  const auto a_at = [&](uint32_t i, uint64_t bb, auto uni) -> const uint64_t * {
    if constexpr(PTRS) return tab_poly<decltype(uni)::value>(kd.a[i], bb, p.a);
    else return kd.a[i] + aoff + blk_off<LOGN>(p, bb);
  };
  const auto b_at = [&](uint32_t i, uint64_t bb, auto uni) -> const uint64_t * {
    if(bc) return kd.b[i] + boff + ((bb & bmask) << LOGN); /* a broadcast b_i^ is one dense polynomial */
    if constexpr(PTRS) return tab_poly<decltype(uni)::value>(kd.b[i], bb, p.a);
    else return kd.b[i] + boff + blk_off<LOGN>(p, bb);
  };
  const auto c_at = [&](uint64_t bb, auto uni) -> uint64_t * {
    if constexpr(PTRS) return tab_poly<decltype(uni)::value>(p.ptab, bb, p.a);
    else return p.a + blk_off<LOGN>(p, bb);
  };
  constexpr std::true_type  kUni{};
  constexpr std::false_type kLane{};

  /* (MULTI -- several limbs in one launch -- exists for batches that cannot fill the chip: a workgroup sees one or two blocks,
   * there is nothing to prefetch across, and the plain loop below needs fewer registers next to the run-time limb's constants) */
  if constexpr(G::PERSISTENT && A::kCompact && !MULTI) {
    static_assert(G::BPW == 1, "the persistent inverse loop owns one block per workgroup");
    constexpr int  GL  = P::NG - 1;
    constexpr bool LTW = LDS_TW > 0;
    const uint64_t stride = gdim;
    uint64_t       b      = bid;
    if(b >= p.nblocks) return;
    typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + P::LDS_ELEMS);
    const lds_ctw_ptr<A>   ltw  = (lds_ctw_ptr<A>)tabl;
    if constexpr(LTW) {
      fill_lds_tables<A, LOGN, true>(tabl, p, (uint32_t)b & bmask, tid);
      __syncthreads();
    }
    /* Register budget (2^14: 128 VGPRs at four waves per SIMD).  The transform kernel keeps the first executed group's
     * twelve per-lane twiddles resident (24 VGPRs) next to ONE prefetched block (32); here the next block's FIRST PAIR is
     * two blocks of words (64), so the twiddles are requested per block instead (from the L2, in front of the products that
     * hide their latency) and the prefetch is issued after the last exchange, when the LDS addresses and the per-lane
     * twiddles of the middle groups are dead: the words then have the last group, the stores and the next block's
     * twiddle request to arrive. */
    constexpr bool IPRE = stage_is_compact<A, LOGN, true>(GL, 0) && P::R(GL) < 4 && G::TBL(GL) == 0;
    uint64_t ra[kE], rb[kE];
    prefetch_last<LOGN, NTT_DOT_AUX_A>(ra, tid, a_at(0, b, kUni));
    prefetch_last_b<LOGN>(rb, tid, b_at(0, b, kUni));
    pin_raw(ra);
    pin_raw(rb);
    for(; b < p.nblocks; b += stride) {
      const uint32_t blk  = (uint32_t)b & bmask;
      uint64_t *     base = c_at(b, kUni);
      /* (an opaque copy of the thread id ties the per-block twiddle request and every lane-dependent address to the
       * iteration: hoisted, they would stay in registers -- or scratch -- for the whole launch; see fused_product_kernel) */
      uint32_t tl = tid;
      asm volatile("" : "+v"(tl));
      typename A::val x[kE];
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = typename A::val{}; });
      /* every pair but the last: add its products, request the next pair */
#pragma unroll 1
      for(uint32_t i = 0; i + 1 < np; i++) {
        if(i != 0 && i % (uint32_t)A::kDotEvery == 0) dot_fold_tile<A>(x, p.c);
        dot_tile<A, 0, kE, NTT_DOT_LOOP_CHUNK>(x, ra, rb, lazy, p.c);
        prefetch_last<LOGN, NTT_DOT_AUX_A>(ra, tl, a_at(i + 1, b, kUni));
        prefetch_last_b<LOGN>(rb, tl, b_at(i + 1, b, kUni));
        sched_fence();
      }
      if(np > 1 && (np - 1) % (uint32_t)A::kDotEvery == 0) dot_fold_tile<A>(x, p.c);
      /* the last pair.  The first executed group's twiddles are requested half way through its products -- which hide
       * most of their L2 latency --, when half of the 64 registers of words are free again (any earlier and they would
       * have to live next to all of them) */
      dot_tile<A, 0, kE / 2>(x, ra, rb, lazy, p.c);
      typename A::ctw pre[4][kE / 2];
      if constexpr(IPRE) preload_group_tw<A, LOGN, GL>(pre, tl, blk, p);
      dot_tile<A, kE / 2, kE>(x, ra, rb, lazy, p.c);
      if(np > 1) dot_fold_tile<A>(x, p.c);
      if constexpr(IPRE) {
        run_group_preloaded<A, LOGN, GL, MASK, true>(x, pre, p);
      } else if constexpr(G::TBL(GL) > 0) {
        run_group<A, LOGN, GL, true, MASK, true>(x, tl, blk, p, ltw + G::TBL_OFF(GL));
      } else {
        run_group<A, LOGN, GL, true, MASK>(x, tl, blk, p);
      }
      const bool     more = b + stride < p.nblocks;
      const uint64_t nb   = more ? b + stride : b;
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = P::NG - 1 - decltype(gg)::value;
        exchange<A, LOGN, GI, GI - 1>(x, tl, lds_all);
        if constexpr(GI == NTT_DOT_A_AT) {
          /* the next block's first pair, operand a: always issued (a dead descriptor moves no data past the end) */
          uint32_t t2 = tid;
          asm volatile("" : "+v"(t2));
          sched_fence();
          prefetch_last<LOGN, NTT_DOT_AUX_A>(ra, t2, a_at(0, nb, kUni), more);
          sched_fence();
        }
        if constexpr(G::TBL(GI - 1) > 0) {
          run_group<A, LOGN, GI - 1, true, MASK, true>(x, tl, blk, p, ltw + G::TBL_OFF(GI - 1));
        } else {
          run_group<A, LOGN, GI - 1, true, MASK>(x, tl, blk, p);
        }
      });
      {
        /* ... operand b: behind the last group's butterflies, whose temporaries do not fit next to 64 registers of words */
        uint32_t t3 = tid;
        asm volatile("" : "+v"(t3));
        sched_fence();
        prefetch_last_b<LOGN>(rb, t3, b_at(0, nb, kUni), more);
        sched_fence();
      }
      uint64_t out[kE];
      static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = out_word<A, true, false>(x[decltype(ee)::value], !LASTINV, p.c); });
      buffer_store_first_raw<LOGN>(out, tl, base);
    }
    return;
  } else {
    /* small blocks (several per workgroup, sharing the LDS tables), the integer policies, several limbs: the plain loop */
    const lds_ctw_ptr<A> gtw = (lds_ctw_ptr<A>)reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
    if constexpr(LDS_TW > 0) {
      /* blocks below 2^12 are whole polynomials; larger ones may be blocks of a bigger transform: a workgroup then always
       * sees the same block position (its stride is a multiple of the blocks per polynomial: launch_dot_blocks) */
      fill_lds_tables<A, LOGN, true>(reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS), p,
                                     G::BPW == 1 ? ((uint32_t)bid & bmask) : 0u, tid);
      __syncthreads();
    }
    for(uint64_t b0 = (uint64_t)bid * G::BPW; b0 < p.nblocks; b0 += (uint64_t)gdim * G::BPW) {
      uint64_t   b    = b0 + sub;
      const bool live = b < p.nblocks;
      if(!live) b = p.nblocks - 1; /* idle sub-blocks shadow a real block (barriers are workgroup-wide), never store */
      const uint32_t blk  = (uint32_t)b & bmask;
      const uint64_t offa = PTRS ? 0 : blk_off<LOGN>(p, b);
      const uint64_t offb = bc ? ((uint64_t)blk << LOGN) : offa;
      uint64_t *     base = PTRS ? c_at(b, kLane) : p.a + offa;
      typename A::val x[kE];
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = typename A::val{}; });
#pragma unroll 1
      for(uint32_t i = 0; i < np; i++) {
        if(i != 0 && i % (uint32_t)A::kDotEvery == 0) dot_fold_tile<A>(x, p.c);
        /* half a tile at a time: 32 registers of words next to the 32 running sums */
        static_for<0, 2>([&](auto hh) {
          constexpr int H = decltype(hh)::value;
          uint64_t      ra[kE], rb[kE];
          sched_fence();
          load_last_raw<LOGN, 8 * H, 8 * H + 8>(ra, t, PTRS ? a_at(i, b, kLane) : kd.a[i] + aoff + offa);
          load_last_raw<LOGN, 8 * H, 8 * H + 8>(rb, t, PTRS ? b_at(i, b, kLane) : kd.b[i] + boff + offb);
          dot_tile<A, 8 * H, 8 * H + 8>(x, ra, rb, lazy, p.c);
        });
      }
      if(np > 1) dot_fold_tile<A>(x, p.c);
      run_group<A, LOGN, P::NG - 1, true, MASK, (G::TBL(P::NG - 1) > 0)>(x, t, blk, p, gtw + G::TBL_OFF(P::NG - 1));
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = P::NG - 1 - decltype(gg)::value;
        exchange<A, LOGN, GI, GI - 1>(x, t, lds);
        run_group<A, LOGN, GI - 1, true, MASK, (G::TBL(GI - 1) > 0)>(x, t, blk, p, gtw + G::TBL_OFF(GI - 1));
      });
      /* (a pass that does not end the transform keeps the integer policies' lazy range, as fused_kernel's does) */
      if(live) global_store_first<A, LOGN, true>(x, t, base, p.c, !LASTINV);
    }
  }
}

/* ------------------------------------------------------------------ */
/* NTT-domain products at N = 2^15 .. 2^17 as items of ONE launch       */
/* ------------------------------------------------------------------ */
/*
 * c = inv( sum_i a_i^ (.) b_i^ ) for polynomials larger than a block: per 128 / 256 MiB chunk the library used to launch
 * dot_inv_kernel over the blocks (the products ride in the inverse's first pass) and then the inverse's column pass -- at
 * 4 GB per operand sixty launches of some 60 us, every one with its own ramp and tail.  Here both passes are the ITEMS of one
 * persistent launch in team_kernel's scheme (per-XCD in-order queues, a per-polynomial counter between the passes, a later
 * pass `lag` polynomials behind: see team_kernel for the protocol and its memory-order invariant):
 *   pass 0  row item   : one 2^12-point block -- the k operand pairs' words of the block, products and their sum in registers
 *                        (A::dot_term / dot_acc / dot_fold), the twelve inverse block stages, intermediate words to c;
 *   pass 1  column item: 256 adjacent columns of c through the LEAD leading inverse stages, N^-1 folded in, final stores.
 * The row item reads its operands' block and writes c's block at the same position, so c may alias an operand for k = 1
 * exactly as in dot_inv_kernel (the item that overwrites a line is the only one that ever read it).  A broadcast b_i^ (one
 * polynomial per limb shared by the batch) is read from the L2 by every item.
 * Reference primitive: fast_mul_mod_q (include/internal/fast_mul_operators.h:56-60) in front of inv_ntt_* (src/ntt_reference.c:33-66).
 */
template <class A> struct KTeamDot {
  KDot<A>  d;       /* d.k.a = c (limb 0), d.k.nblocks = polynomials PER LIMB; operand pointers, flags, strides as for dot_inv_kernel */
  TeamCtl *ctl;     /* zeroed before the launch */
  uint64_t split_rcp;
  uint32_t lag, nlimbs, poly_major;
};

/* row item: block `blk` of one polynomial; offa / offb = word offsets of the block inside the a-like operands (and c) / the b operands */
/* PTRS: kd.a[i] / kd.b[i] (unless broadcast) are device tables; pl = the polynomial, offa = the block's offset inside it */
template <class A, int KSH, bool PTRS = false>
__device__ __forceinline__ void team_dot_row_item(uint64_t *cblk, uint64_t offa, uint64_t offb, uint32_t blk, uint32_t tid0, const Params<A> &p,
                                                  const KDot<A> &kd, uint64_t aoff, uint64_t boff, typename A::val *lds, typename A::ctw *tabl,
                                                  uint32_t pl = 0)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, true, flavor_of<A>()>;
  constexpr int GL   = P::NG - 1;
  constexpr uint32_t MASK = fused_mask<A, LOGN, true, KSH>(); /* not the pass that ends the transform; inputs are products, not canonical words */
  constexpr bool     MID_LAZY = !A::kTracksBounds;
  /* (an opaque copy of the thread id ties every lane-dependent address to the item: see team_product_item) */
  uint32_t tl = tid0;
  asm volatile("" : "+v"(tl));
  const uint32_t tid  = tl;
  const uint32_t np   = kd.npairs;
  const bool     lazy = kd.lazy_in != 0;
  [[maybe_unused]] const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  constexpr bool IPRE = A::kCompact && stage_is_compact<A, LOGN, true>(GL, 0) && P::R(GL) < 4 && G::TBL(GL) == 0 && KSH != 1;
  [[maybe_unused]] typename A::ctw pre[4][kE / 2];
  if constexpr(A::kCompact) fill_lds_tables<A, LOGN, true>(tabl, p, blk, tid);
  typename A::val x[kE];
  static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = typename A::val{}; });
#pragma unroll 1
  for(uint32_t i = 0; i < np; i++) {
    if(i != 0 && i % (uint32_t)A::kDotEvery == 0) dot_fold_tile<A>(x, p.c);
    /* half a tile at a time: 32 registers of words next to the 32 running sums (dot_inv_kernel's plain loop) */
    static_for<0, 2>([&](auto hh) {
      constexpr int H = decltype(hh)::value;
      uint64_t      ra[kE], rb[kE];
      sched_fence();
      const uint64_t *ai = PTRS ? tab_poly<true>(kd.a[i], pl, p.a) + offa : kd.a[i] + aoff + offa;
      const uint64_t *bi = PTRS && kd.b_bcast == 0 ? tab_poly<true>(kd.b[i], pl, p.a) + offa : kd.b[i] + boff + offb;
      load_last_raw<LOGN, 8 * H, 8 * H + 8, false>(ra, tid, ai);
      load_last_raw<LOGN, 8 * H, 8 * H + 8>(rb, tid, bi);
      dot_tile<A, 8 * H, 8 * H + 8>(x, ra, rb, lazy, p.c);
    });
  }
  if(np > 1) dot_fold_tile<A>(x, p.c);
  /* (the first group's twiddles: requested behind the products, whose 64 registers of words are free again) */
  if constexpr(A::kCompact && IPRE) preload_group_tw<A, LOGN, GL>(pre, tid, blk, p);
  if constexpr(A::kCompact) __syncthreads(); /* the LDS table is read after a wave-local exchange: it needs a barrier of its own */
  if constexpr(A::kCompact && IPRE) {
    run_group_preloaded<A, LOGN, GL, MASK, true>(x, pre, p);
  } else {
    run_group<A, LOGN, GL, true, MASK>(x, tid, blk, p);
  }
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    exchange<A, LOGN, GI, GI - 1>(x, tid, lds);
    if constexpr(A::kCompact && G::TBL(GI - 1) > 0) {
      run_group<A, LOGN, GI - 1, true, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI - 1));
    } else {
      run_group<A, LOGN, GI - 1, true, MASK>(x, tid, blk, p);
    }
  });
  uint64_t out[kE];
  static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = out_word<A, true, false>(x[decltype(ee)::value], MID_LAZY, p.c); });
  buffer_store_first_raw<LOGN, 0>(out, tid, cblk);
}

/* PTRS: kt.d.k.ptab, kt.d.a[i], kt.d.b[i] (unless broadcast) are device tables, kt.d.k.a = the first limb's offset (MULTI: team_limb adds the
 * item's limb to it, so tab_poly(.., p.a) lands in that limb) */
template <class A, int LEAD, int KSH, bool MULTI = false, bool PTRS = false>
__global__ void __launch_bounds__(256, 4) team_dot_kernel(const KTeamDot<A> kt)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, true, flavor_of<A>()>;
  static_assert((A::kCompact || A::kIntWide) && P::T == kTeamCols && LEAD >= 3 && LEAD <= 5,
                "built for the FP64 policies and the wide integer policy on 2^12-point blocks, N = 2^15..2^17");
  __shared__ typename A::val lds[P::LDS_ELEMS + G::LDS_TW];
  __shared__ unsigned        s_k, s_k2[2];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds + P::LDS_ELEMS);
  const uint32_t         tid  = threadIdx.x;
  uint32_t               bid_, gdim_, limb_;
  Params<A>              p = limb_params<A, true, false>(kt.d.k, bid_, gdim_, limb_);
  p.s0                     = LEAD;
  constexpr uint32_t CMASK = column_mask<A, LEAD, true, KSH>();
  const uint32_t logn  = LOGN + LEAD;
  const uint32_t batch = (uint32_t)p.nblocks; /* polynomials per limb */
  const uint32_t total = MULTI ? batch * kt.nlimbs : batch;
  constexpr uint32_t NCOL = 1u << (LOGN - 8), NROW = 1u << LEAD;
  TeamCtl *const ctl = kt.ctl;
  const uint32_t lag = kt.lag;
  const uint32_t my  = xcc_id();
  const bool     bc  = kt.d.b_bcast != 0;
  uint64_t       aoff = 0, boff = 0; /* the item's limb: word offsets of its slabs (MULTI) */
  for(uint32_t qq = 0; qq < 8; qq++) {
    const uint32_t q = (my + qq) & 7u;
    if(tid == 0) {
      const unsigned prev = atomicCAS(&ctl->owner[q][0], 0u, my + 1u);
      s_k                 = (prev == 0u || prev == my + 1u) ? 1u : 0u;
    }
    __syncthreads();
    const bool mine = s_k != 0;
    __syncthreads();
    if(!mine) continue;
    constexpr uint32_t kNoSignal = 0xffffffffu;
    uint32_t           sig       = kNoSignal;
    for(uint32_t it = 0;; it ^= 1u) {
      /* (lane-0 blocks are followed at once by a workgroup barrier: see team_kernel) */
      if(tid == 0) {
        if(sig != kNoSignal) __hip_atomic_fetch_add(&ctl->done[sig], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_k2[it] = atomicAdd(&ctl->next[q][0], 1u);
      }
      sig = kNoSignal;
      __syncthreads();
      const TeamItem ti = team_decode(uniform_u32(s_k2[it]), q, total, lag, NROW, NCOL, 0u);
      if(ti.stop) break;
      if(!ti.valid) continue;
      const bool     second = ti.pass != 0;
      const uint32_t item = ti.item, pidx = ti.v;
      uint32_t       pl   = pidx;
      if constexpr(MULTI) {
        uint32_t limb;
        team_split(pidx, batch, kt.nlimbs, kt.poly_major, kt.split_rcp, limb, pl);
        team_limb<A, true>(p, kt.d.k, limb);
        aoff = (uint64_t)limb * kt.d.k.limb_stride;
        boff = (uint64_t)limb * kt.d.b_limb_stride;
      }
      const uint64_t poff = PTRS ? 0 : (uint64_t)pl * p.pstride; /* the polynomial inside its limb: c and every operand laid out like it */
      uint64_t *     poly = PTRS ? tab_poly<true>(p.ptab, pl, p.a) : p.a + poff;
      if(second) {
        if(tid == 0) {
          while(__hip_atomic_load(&ctl->done[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NROW) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
        team_column_item<A, LEAD, true, CMASK, kAuxNt, kAuxSc1>(poly, item * kTeamCols + tid, logn, p, false);
      } else {
        const uint64_t offa = poff + ((uint64_t)item << LOGN);
        const uint64_t offb = bc ? ((uint64_t)item << LOGN) : offa;
        team_dot_row_item<A, KSH, PTRS>(poly + ((uint64_t)item << LOGN), offa, offb, item, tid, p, kt.d, aoff, boff, lds, tabl, pl);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sig = pidx;
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* forward transform with the product at its output: c^ = fwd(a) (.) b^ (+ c^) */
/* ------------------------------------------------------------------ */
/*
 * The counterpart of dot_inv_kernel on the other side of the path: the operand comes in as coefficients, the result STAYS in
 * the NTT domain -- a plaintext or key factor b^ kept transformed is multiplied in where the forward block kernel would
 * reduce and store its outputs, optionally added to what c^ already holds (the multiply-accumulate of a key-switching inner
 * product, digit by digit).  24N bytes (16N with a broadcast b^) instead of 40N for forward transform + pointwise product;
 * with the accumulator 32N (24N) instead of 48N.  The forward transform ends in the layout the element-wise product needs
 * (runs of four consecutive coefficients per lane), so b^ and c^ are read and c^ written as 16-byte words by the lane that
 * owns them: c may alias a or b^.  Blocks of a larger transform (N > 2^14: the block pass is the forward transform's LAST
 * pass) work the same way.  Reference primitive: fast_mul_mod_q (include/internal/fast_mul_operators.h:56-60).
 * Registers (2^14: 128 VGPRs): the next block's words are in flight during the whole iteration (32), so the last group's
 * twiddles are requested per block instead of staying resident (24), b^ is fetched in two halves -- the first one in front of
 * the last group, the second one while the first half's products run -- and the accumulator words right where each half is
 * finished.
 */
template <class A> struct KMul {
  KArgs<A>        k;             /* k.a = a (coefficients, limb 0); nblocks / s0 / logn as for a forward block pass */
  const uint64_t *b;             /* b^ (limb 0) */
  uint64_t *      out;           /* c^ (limb 0) */
  uint64_t        b_limb_stride; /* words between consecutive limbs of b^ */
  uint32_t        lazy_in;       /* words of b^ may be lazy: anywhere in [0,4q) */
  uint32_t        b_bcast;       /* b^ is one polynomial per limb, shared by the whole batch */
  uint32_t        accumulate;    /* c^ += ... (c^ canonical on entry) */
};

/* live = false: a descriptor of zero records -- the loads return 0 and move no data (the accumulator words of a call that
 * does not accumulate: a run-time branch around the loads would make the register allocator keep two sets apart) */
template <int LOGN, int E0, int E1, int AUX = 0>
__device__ __forceinline__ void prefetch_last_range(uint64_t (&raw)[kE], uint32_t t, const uint64_t *blk, bool live = true)
{
  using P           = Plan<LOGN>;
  constexpr int G   = P::NG - 1;
  const uint32_t ib = P::IBASE(G, t);
  const __amdgpu_buffer_rsrc_t r = block_rsrc<LOGN>(blk, live);
  static_for<E0 / 2, E1 / 2>([&](auto hh) {
    constexpr int E = 2 * decltype(hh)::value;
    const u64x2   v = buffer_load_u64x2<AUX>(r, ib * 8u, P::IOFF(G, E) * 8u);
    raw[E]          = v.a;
    raw[E + 1]      = v.b;
  });
}
template <int LOGN, int E0, int E1>
__device__ __forceinline__ void buffer_store_last_range(const uint64_t (&u)[kE], uint32_t t, uint64_t *blk)
{
  using P           = Plan<LOGN>;
  constexpr int G   = P::NG - 1;
  const uint32_t ib = P::IBASE(G, t);
  const __amdgpu_buffer_rsrc_t r = block_rsrc<LOGN>(blk);
  typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
  static_for<E0 / 2, E1 / 2>([&](auto hh) {
    constexpr int E = 2 * decltype(hh)::value;
    v4u32         v;
    v.x = (unsigned)u[E];
    v.y = (unsigned)(u[E] >> 32);
    v.z = (unsigned)u[E + 1];
    v.w = (unsigned)(u[E + 1] >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(ib * 8u), (int)(P::IOFF(G, E) * 8u), 0);
  });
}

/* PTRS (whole polynomials): km.k.ptab, km.b (unless broadcast) and km.out are the tables of a, b^ and c^; km.k.a = the first limb's
 * offset (tab_poly), the limbs of an RNS set km.k.limb_stride words apart behind every table entry (MULTI: limb_params adds them). */
template <class A, int LOGN, int KSH, bool MULTI = false, bool PTRS = false>
__global__ void __launch_bounds__((Geom<LOGN, false, flavor_of<A>()>::WG), (Geom<LOGN, false, flavor_of<A>()>::WPS)) fwd_mul_kernel(const KMul<A> km)
{
  uint32_t        bid, gdim, limb;
  const Params<A> p = limb_params<A, false, MULTI>(km.k, bid, gdim, limb);
  using P = Plan<LOGN>;
  using G = Geom<LOGN, false, flavor_of<A>()>;
  constexpr uint32_t MASK   = fused_mask<A, LOGN, false, KSH>();
  constexpr int      LDS_TW = G::LDS_TW;
  __shared__ typename A::val lds_all[G::BPW * P::LDS_ELEMS + LDS_TW];
  const uint32_t   tid   = threadIdx.x;
  const uint32_t   sub   = tid >> P::LT;
  const uint32_t   t     = tid & (P::T - 1);
  typename A::val *lds   = lds_all + sub * P::LDS_ELEMS;
  const uint32_t   bmask = (1u << p.s0) - 1u;
  const bool       lazy  = km.lazy_in != 0;
  const bool       bc    = km.b_bcast != 0;
  const bool       acc   = km.accumulate != 0;
  const uint64_t * bptr  = km.b + (uint64_t)limb * km.b_limb_stride; /* (MULTI off: limb == 0) */
  uint64_t *       cptr  = km.out + (uint64_t)limb * km.k.limb_stride;
  /* where block bb of the three operands starts (UNI: bb is the same for the whole wave) */
  /* (the persistent loop's blocks hold at least one wave each: uniform -- with two blocks per workgroup the block index is derived
   * from the thread id and has to be said to be; the plain loop's blocks may be smaller than a wave) */
  const auto wave = [&](uint64_t bb, auto uni) {
    if constexpr(decltype(uni)::value && G::BPW != 1) return ((uint64_t)uniform_u32((uint32_t)(bb >> 32)) << 32) | uniform_u32((uint32_t)bb);
    else return bb;
  };
  const auto a_at = [&](uint64_t bb, auto uni) -> const uint64_t * {
    if constexpr(PTRS) return tab_poly<decltype(uni)::value>(p.ptab, wave(bb, uni), p.a);
    else return p.a + blk_off<LOGN>(p, bb);
  };
  const auto b_at = [&](uint64_t bb, auto uni) -> const uint64_t * {
    if(bc) return bptr + ((bb & bmask) << LOGN); /* a broadcast b^ is one dense polynomial */
    if constexpr(PTRS) return tab_poly<decltype(uni)::value>(km.b, wave(bb, uni), p.a);
    else return bptr + blk_off<LOGN>(p, bb);
  };
  const auto c_at = [&](uint64_t bb, auto uni) -> uint64_t * {
    if constexpr(PTRS) return tab_poly<decltype(uni)::value>(km.out, wave(bb, uni), p.a);
    else return cptr + blk_off<LOGN>(p, bb);
  };
  constexpr std::true_type  kUni{};
  constexpr std::false_type kLane{};

  if constexpr(G::PERSISTENT && A::kCompact && !MULTI) {
    constexpr int  GL  = P::NG - 1;
    constexpr bool PRE = stage_is_compact<A, LOGN, false>(GL, 0) && G::TBL(GL) == 0;
    constexpr bool LTW = LDS_TW > 0;
    static_assert(P::NG >= 3, "the persistent blocks have at least three stage groups");
    const uint32_t         tt     = G::BPW == 1 ? tid : t;
    typename A::val *const ll     = G::BPW == 1 ? lds_all : lds;
    const uint64_t         stride = (uint64_t)gdim * G::BPW;
    uint64_t               b0     = (uint64_t)bid * G::BPW;
    if(b0 >= p.nblocks) return;
    const uint64_t lastb = p.nblocks - 1;
    uint64_t       b     = G::BPW == 1 ? b0 : (b0 + sub < p.nblocks ? b0 + sub : lastb);
    typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
    const lds_ctw_ptr<A>   ltw  = (lds_ctw_ptr<A>)tabl;
    if constexpr(LTW) {
      fill_lds_tables<A, LOGN, false>(tabl, p, (uint32_t)b & bmask, tid);
      __syncthreads();
    }
    uint64_t raw[kE];
    prefetch_first<LOGN>(raw, tt, a_at(b, kUni));
    pin_raw(raw);
    for(; b0 < p.nblocks; b0 += stride) {
      const bool live = G::BPW == 1 || b0 + sub < p.nblocks;
      b               = live ? b0 + (G::BPW == 1 ? 0u : sub) : lastb;
      const uint32_t  blk   = (uint32_t)b & bmask;
      const uint64_t *bblk  = b_at(b, kUni);
      uint64_t *      cblk  = c_at(b, kUni);
      uint32_t        tl    = tt;
      asm volatile("" : "+v"(tl)); /* ties the per-block requests to the iteration (see dot_inv_kernel) */
      typename A::val x[kE];
      convert_inputs<A, false>(x, raw, false, p.c);
      {
        const bool     more = b0 + stride < p.nblocks;
        const uint64_t nb0  = more ? b0 + stride : b0;
        const uint64_t nb   = G::BPW == 1 ? nb0 : (nb0 + sub < p.nblocks ? nb0 + sub : lastb);
        prefetch_first<LOGN>(raw, tl, a_at(nb, kUni), more);
      }
      run_group<A, LOGN, 0, false, MASK>(x, tl, blk, p);
      typename A::ctw pre[4][kE / 2];
      uint64_t        rb[kE], rc[kE];
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        if constexpr(PRE && GI + 1 == GL) {
          /* the last group's twiddles: requested in front of the exchange into it (its LDS round trip hides part of the L2
           * latency; a whole group ahead they would live through the table group next to the prefetched block: spills) */
          sched_fence();
          preload_group_tw<A, LOGN, GL>(pre, tl, blk, p);
          sched_fence();
        }
        exchange<A, LOGN, GI, GI + 1>(x, tl, ll);
        if constexpr(PRE && GI + 1 == GL) {
          run_group_preloaded<A, LOGN, GL, MASK>(x, pre, p);
        } else if constexpr(G::TBL(GI + 1) > 0) {
          run_group<A, LOGN, GI + 1, false, MASK, true>(x, tl, blk, p, ltw + G::TBL_OFF(GI + 1));
        } else {
          run_group<A, LOGN, GI + 1, false, MASK>(x, tl, blk, p);
        }
      });
      /* the products, a quarter of the tile at a time, the next quarter's words in flight meanwhile: at most eight 16-byte
       * words of b^ and c^ per thread live next to the 32 values and the 32 prefetched words of the next block */
      uint64_t u[kE];
      uint32_t t2 = tt; /* (a fresh opaque copy: the lane offsets of this phase are computed here, not carried through the groups) */
      asm volatile("" : "+v"(t2));
      sched_fence();
      prefetch_last_range<LOGN, 0, 4>(rb, t2, bblk);
      static_for<0, 4>([&](auto qq) {
        constexpr int Q = decltype(qq)::value;
        sched_fence();
        /* this quarter's accumulator words (zeros, and no traffic, when the call does not accumulate) and the next
         * quarter's b^ words */
        prefetch_last_range<LOGN, 4 * Q, 4 * Q + 4>(rc, t2, cblk, acc);
        if constexpr(Q < 3) prefetch_last_range<LOGN, 4 * Q + 4, 4 * Q + 8>(rb, t2, bblk);
        sched_fence();
        mul_out_tile<A, 4 * Q, 4 * Q + 4, 1>(u, x, rb, rc, lazy, p.c); /* (one product at a time: two need 14 registers more than there are) */
        if(live) buffer_store_last_range<LOGN, 4 * Q, 4 * Q + 4>(u, t2, cblk);
        sched_fence();
      });
    }
    return;
  } else {
    /* small blocks (several per workgroup), the integer policies, several limbs: the plain loop */
    const lds_ctw_ptr<A> gtw = (lds_ctw_ptr<A>)reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
    if constexpr(LDS_TW > 0) {
      fill_lds_tables<A, LOGN, false>(reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS), p,
                                      G::BPW == 1 ? ((uint32_t)bid & bmask) : 0u, tid);
      __syncthreads();
    }
    for(uint64_t b0 = (uint64_t)bid * G::BPW; b0 < p.nblocks; b0 += (uint64_t)gdim * G::BPW) {
      uint64_t   b    = b0 + sub;
      const bool live = b < p.nblocks;
      if(!live) b = p.nblocks - 1;
      const uint32_t  blk  = (uint32_t)b & bmask;
      const uint64_t *bblk = b_at(b, kLane);
      uint64_t *      cblk = c_at(b, kLane);
      /* (an opaque copy of the thread id per block: the integer policy's per-lane twiddle addresses would otherwise be
       * computed once for the launch and sit in registers -- or scratch -- throughout) */
      uint32_t tg = t;
      asm volatile("" : "+v"(tg));
      typename A::val x[kE];
      global_load_first<A, LOGN, false>(x, tg, a_at(b, kLane), false, p.c);
      run_group<A, LOGN, 0, false, MASK, (G::TBL(0) > 0)>(x, tg, blk, p, gtw);
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        exchange<A, LOGN, GI, GI + 1>(x, tg, lds);
        run_group<A, LOGN, GI + 1, false, MASK, (G::TBL(GI + 1) > 0)>(x, tg, blk, p, gtw + G::TBL_OFF(GI + 1));
      });
      static_for<0, 4>([&](auto qq) {
        constexpr int Q = decltype(qq)::value;
        uint64_t      rb[kE], rc[kE], u[kE];
        sched_fence();
        load_last_raw<LOGN, 4 * Q, 4 * Q + 4>(rb, tg, bblk);
        /* (the accumulator words: c^ itself, or zeros when the call does not accumulate) */
        if(acc) load_last_raw<LOGN, 4 * Q, 4 * Q + 4>(rc, tg, cblk);
        else static_for<4 * Q, 4 * Q + 4>([&](auto ee) { rc[decltype(ee)::value] = 0; });
        mul_out_tile<A, 4 * Q, 4 * Q + 4, 1>(u, x, rb, rc, lazy, p.c);
        if(live) store_last_raw<LOGN, 4 * Q, 4 * Q + 4>(u, tg, cblk);
        sched_fence();
      });
    }
  }
}

/* c^ = fwd(a) (.) b^ (+ c^) at N = 2^15 in ONE pass (round 6): onepass_forward (ntt_kernels_block.h) with fwd_mul_kernel's epilogue where a
 * half would be reduced and stored -- b^ and the accumulator read in the last group's layout by the lane that owns the words, a
 * quarter of the tile at a time, c^ written; a itself is left as it was.  The one-pass transform is bound by its arithmetic and the
 * half of its loads it cannot prefetch, not by bytes: the 8N (16N accumulating) more that the product reads ride almost free --
 * measured 0.475 -> see profiles/r06/onepass_products_2p15.txt of the 24N roofline. */
/* PTRS (one limb per launch): km.k.ptab, km.b (unless broadcast) and km.out are the tables of a, b^ and c^, km.k.a = the limb's offset --
 * as fwd_mul_kernel's PTRS form. */
template <class A, int KSH, bool MULTI = false, bool PTRS = false>
__global__ void __launch_bounds__(1024, 4) onepass_mul_kernel(const KMul<A> km)
{
  static_assert(!PTRS || !MULTI, "pointer tables: one limb per launch");
  uint32_t  bid, gdim, limb;
  Params<A> p = limb_params<A, false, MULTI>(km.k, bid, gdim, limb);
  constexpr int LOGN = kFusedLarge;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, flavor_of<A>()>;
  static_assert(A::kCompact && A::kTracksBounds && G::BPW == 1 && P::T == 1024, "built for the FP64 policies on the 2^14 block");
  __shared__ typename A::val lds_all[P::LDS_ELEMS + G::LDS_TW];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + P::LDS_ELEMS);
  constexpr uint64_t HALF     = 1ull << LOGN;
  const bool       lazy = km.lazy_in != 0, bc = km.b_bcast != 0, acc = km.accumulate != 0;
  const uint64_t * bptr = km.b + (uint64_t)limb * km.b_limb_stride; /* (MULTI off: limb == 0) */
  uint64_t *       cptr = km.out + (uint64_t)limb * km.k.limb_stride;
  onepass_forward<A, KSH>(p, bid, gdim, threadIdx.x, lds_all, tabl, [&](typename A::val(&x)[kE], uint32_t tl, uint32_t h, uint64_t off, uint64_t poly) {
    const uint64_t *bblk = (PTRS && !bc ? tab_poly<true>(km.b, poly, p.a) : bptr + (bc ? 0 : off)) + (h ? HALF : 0);
    uint64_t *      cblk = (PTRS ? tab_poly<true>(km.out, poly, p.a) : cptr + off) + (h ? HALF : 0);
    /* the products, a quarter of the tile at a time, the next quarter's words in flight meanwhile (as fwd_mul_kernel) */
    uint64_t u[kE], rb[kE], rc[kE];
    uint32_t t2 = tl;
    asm volatile("" : "+v"(t2));
    sched_fence();
    prefetch_last_range<LOGN, 0, 4>(rb, t2, bblk);
    static_for<0, 4>([&](auto qq) {
      constexpr int Q = decltype(qq)::value;
      sched_fence();
      prefetch_last_range<LOGN, 4 * Q, 4 * Q + 4>(rc, t2, cblk, acc);
      if constexpr(Q < 3) prefetch_last_range<LOGN, 4 * Q + 4, 4 * Q + 8>(rb, t2, bblk);
      sched_fence();
      mul_out_tile<A, 4 * Q, 4 * Q + 4, 1>(u, x, rb, rc, lazy, p.c);
      buffer_store_last_range<LOGN, 4 * Q, 4 * Q + 4>(u, t2, cblk);
      sched_fence();
    });
  });
}

/* ------------------------------------------------------------------ */
/* c^ = fwd(a) (.) b^ (+ c^) at N = 2^15 .. 2^17 as ONE launch            */
/* ------------------------------------------------------------------ */
/*
 * team_kernel's forward scheme (column items of polynomial j, row items of polynomial j - lag in the same XCD's queue, a
 * per-polynomial counter between them) with fwd_mul_kernel's epilogue in the row items: where the forward block stages would
 * reduce and store their outputs, b^ (and the accumulator) are read by the lane that owns the words and c^ is written --
 * instead of a column launch and a block launch per 256 MiB chunk.  a is scratch (its column stages run in place), c^ may
 * alias a (a block's words are consumed before its products are stored; not when accumulating) or b^.
 */
template <class A> struct KTeamMul {
  KMul<A>  m;       /* m.k.a = a (limb 0), m.k.nblocks = polynomials PER LIMB; b^, c^, flags and strides as for fwd_mul_kernel */
  TeamCtl *ctl;     /* zeroed before the launch */
  uint64_t split_rcp;
  uint32_t lag, nlimbs, poly_major;
};

template <class A, int KSH, int LDAUX>
__device__ __forceinline__ void team_mul_row_item(const uint64_t *ablk, const uint64_t *bblk, uint64_t *cblk, uint32_t blk, uint32_t tid0,
                                                  const Params<A> &p, bool lazy, bool acc, typename A::val *lds, typename A::ctw *tabl)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, flavor_of<A>()>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, false, KSH>();
  /* (an opaque copy of the thread id ties every lane-dependent address to the item: see team_product_item) */
  uint32_t tl = tid0;
  asm volatile("" : "+v"(tl));
  const uint32_t tid = tl;
  uint64_t raw[kE];
  prefetch_first<LOGN, LDAUX>(raw, tid, ablk);
  typename A::val x[kE];
  if constexpr(A::kCompact) {
    constexpr int GL = P::NG - 1;
    static_assert(G::TBL(GL - 1) > 0 && G::TBL(GL) == 0 && !P::WAVE_LOCAL(0, 1), "twiddle placement / barrier this item assumes");
    const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
    typename A::ctw      pre[4][kE / 2];
    preload_group_tw<A, LOGN, GL>(pre, tid, blk, p);
    fill_lds_tables<A, LOGN, false>(tabl, p, blk, tid); /* (published by the first exchange's barriers) */
    convert_inputs<A, false>(x, raw, false, p.c);
    run_group<A, LOGN, 0, false, MASK>(x, tid, blk, p);
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int GI = decltype(gg)::value;
      exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
      if constexpr(GI + 1 == GL) {
        run_group_preloaded<A, LOGN, GL, MASK>(x, pre, p);
      } else if constexpr(G::TBL(GI + 1) > 0) {
        run_group<A, LOGN, GI + 1, false, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI + 1));
      } else {
        run_group<A, LOGN, GI + 1, false, MASK>(x, tid, blk, p);
      }
    });
  } else {
    (void)tabl;
    convert_inputs<A, false>(x, raw, false, p.c);
    run_group<A, LOGN, 0, false, MASK>(x, tid, blk, p);
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int GI = decltype(gg)::value;
      exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
      run_group<A, LOGN, GI + 1, false, MASK>(x, tid, blk, p);
    });
  }
  /* the products, a quarter of the tile at a time (fwd_mul_kernel's plain loop) */
  uint32_t t2 = tid0;
  asm volatile("" : "+v"(t2));
  static_for<0, 4>([&](auto qq) {
    constexpr int Q = decltype(qq)::value;
    uint64_t      rb[kE], rc[kE], u[kE];
    sched_fence();
    load_last_raw<LOGN, 4 * Q, 4 * Q + 4>(rb, t2, bblk);
    if(acc) load_last_raw<LOGN, 4 * Q, 4 * Q + 4>(rc, t2, cblk);
    else static_for<4 * Q, 4 * Q + 4>([&](auto ee) { rc[decltype(ee)::value] = 0; });
    mul_out_tile<A, 4 * Q, 4 * Q + 4, 1>(u, x, rb, rc, lazy, p.c);
    store_last_raw<LOGN, 4 * Q, 4 * Q + 4>(u, t2, cblk);
    sched_fence();
  });
}

/* PTRS: kt.m.k.ptab, kt.m.b (unless broadcast) and kt.m.out are the device tables of a, b^ and c^, kt.m.k.a = the first limb's offset (MULTI:
 * team_limb adds the item's limb to it) */
template <class A, int LEAD, int KSH, bool MULTI = false, bool PTRS = false>
__global__ void __launch_bounds__(256, 4) team_mul_kernel(const KTeamMul<A> kt)
{
  
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, flavor_of<A>()>;
  static_assert((A::kCompact || A::kIntWide) && P::T == kTeamCols && LEAD >= 3 && LEAD <= 5,
                "built for the FP64 policies and the wide integer policy on 2^12-point blocks, N = 2^15..2^17");
  __shared__ typename A::val lds[P::LDS_ELEMS + G::LDS_TW];
  __shared__ unsigned        s_k, s_k2[2];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds + P::LDS_ELEMS);
  const uint32_t         tid  = threadIdx.x;
  uint32_t               bid_, gdim_, limb_;
  Params<A>              p = limb_params<A, false, false>(kt.m.k, bid_, gdim_, limb_);
  p.s0                     = LEAD;
  constexpr uint32_t CMASK    = column_mask<A, LEAD, false, KSH>();
  constexpr bool     MID_LAZY = !A::kTracksBounds; /* words between the passes: canonical for the FP64 policies */
  const uint32_t logn  = LOGN + LEAD;
  const uint32_t batch = (uint32_t)p.nblocks; /* polynomials per limb */
  const uint32_t total = MULTI ? batch * kt.nlimbs : batch;
  constexpr uint32_t NCOL = 1u << (LOGN - 8), NROW = 1u << LEAD;
  TeamCtl *const ctl = kt.ctl;
  const uint32_t lag = kt.lag;
  const uint32_t my  = xcc_id();
  const bool     bc  = kt.m.b_bcast != 0, lazy = kt.m.lazy_in != 0, acc = kt.m.accumulate != 0;
  uint64_t       boff = 0, coff = 0; /* the item's limb: word offsets of its b^ and c^ slabs (MULTI) */
  for(uint32_t qq = 0; qq < 8; qq++) {
    const uint32_t q = (my + qq) & 7u;
    if(tid == 0) {
      const unsigned prev = atomicCAS(&ctl->owner[q][0], 0u, my + 1u);
      s_k                 = (prev == 0u || prev == my + 1u) ? 1u : 0u;
    }
    __syncthreads();
    const bool mine = s_k != 0;
    __syncthreads();
    if(!mine) continue;
    constexpr uint32_t kNoSignal = 0xffffffffu;
    uint32_t           sig       = kNoSignal;
    for(uint32_t it = 0;; it ^= 1u) {
      /* (lane-0 blocks are followed at once by a workgroup barrier: see team_kernel) */
      if(tid == 0) {
        if(sig != kNoSignal) __hip_atomic_fetch_add(&ctl->done[sig], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_k2[it] = atomicAdd(&ctl->next[q][0], 1u);
      }
      sig = kNoSignal;
      __syncthreads();
      const TeamItem ti = team_decode(uniform_u32(s_k2[it]), q, total, lag, NCOL, NROW, 0u);
      if(ti.stop) break;
      if(!ti.valid) continue;
      const bool     second = ti.pass != 0;
      const uint32_t item = ti.item, pidx = ti.v;
      uint32_t       pl   = pidx;
      if constexpr(MULTI) {
        uint32_t limb;
        team_split(pidx, batch, kt.nlimbs, kt.poly_major, kt.split_rcp, limb, pl);
        team_limb<A, false>(p, kt.m.k, limb);
        boff = (uint64_t)limb * kt.m.b_limb_stride;
        coff = (uint64_t)limb * kt.m.k.limb_stride;
      }
      const uint64_t poff = (uint64_t)pl * p.pstride; /* the polynomial inside its limb: a, c^ and a per-polynomial b^ alike */
      uint64_t *     poly = PTRS ? tab_poly<true>(p.ptab, pl, p.a) : p.a + poff;
      if(!second) {
        /* inputs -> intermediate (kept dirty in the L2), as in team_kernel */
        team_column_item<A, LEAD, false, CMASK, kAuxSc0Sc1, 0>(poly, item * kTeamCols + tid, logn, p, MID_LAZY);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sig = pidx;
      } else {
        if(tid == 0) {
          while(__hip_atomic_load(&ctl->done[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NCOL) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
        const uint64_t ioff = (uint64_t)item << LOGN;
        const uint64_t *bblk = bc ? kt.m.b + boff + ioff : (PTRS ? tab_poly<true>(kt.m.b, pl, p.a) + ioff : kt.m.b + boff + poff + ioff);
        uint64_t *      cblk = PTRS ? tab_poly<true>(kt.m.out, pl, p.a) + ioff : kt.m.out + coff + poff + ioff;
        team_mul_row_item<A, KSH, kAuxNt>(poly + ioff, bblk, cblk, item, tid, p, lazy, acc, lds, tabl);
      }
    }
  }
}

} /* namespace ntt */
