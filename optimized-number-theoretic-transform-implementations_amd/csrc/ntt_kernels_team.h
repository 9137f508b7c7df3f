/*
 * ntt_kernels_team.h -- team_kernel: both passes of a 2^15 .. 2^17 transform as items of ONE persistent launch -- per-XCD queues, per-polynomial hand-off
 * counters, the intermediate kept behind one L2 (the memory-order invariant of the hand-off is stated here).
 * Part of ntt_kernels.h (included from there, in this order: block, team, products, launch); not a header of its own.
 */
#pragma once

namespace ntt {

/* ------------------------------------------------------------------ */
/* N = 2^15 .. 2^17: both passes in one launch, intermediate kept in the XCD's L2 */
/* ------------------------------------------------------------------ */
/*
 * The two passes of a large transform -- LEAD = m - 12 strided column stages, then the 2^12-point blocks -- as ITEMS of
 * one persistent launch instead of two launches per 256 MiB chunk.  A column item is 256 adjacent columns of one
 * polynomial (column_pass_thread's work: 2^LEAD values per thread in registers, no exchange, every access a contiguous
 * 2 KiB row segment), a row item one 2^12-point block (the fused block kernel's work).  What makes it worth a kernel:
 *   - a workgroup reads which XCD it runs on (HW_REG_XCC_ID) and pulls items from THAT XCD's queue, so all items of a
 *     polynomial run on one XCD whatever the dispatcher does: the intermediate is written by plain stores into that
 *     XCD's 4 MiB L2 and read back from it -- measured FETCH_SIZE 1.0x the data instead of 2.0x (profiles/r03,
 *     skel_pmc_fabric_traffic.txt).  Final stores are write-through (sc1: the line leaves the L2 at once) and input loads
 *     sc0 sc1, so the streaming sides do not push the waiting intermediates out of the L2;
 *   - queue order per XCD: first-pass items of polynomial j, then second-pass items of polynomial j - LAG (forward:
 *     columns then rows; inverse: rows then columns).  A second-pass item waits on a per-polynomial counter that the
 *     first-pass items bump once their stores have completed (s_waitcnt vmcnt(0), workgroup barrier, agent-scope
 *     atomic).  First-pass items never wait and items are handed out in order, so every item somebody waits for is
 *     already in the hands of a running workgroup: no deadlock whatever the residency;
 *   - polynomials are dealt to the eight queues statically (p mod 8); a queue is processed only by the XCD that owns it
 *     (compare-and-swap on first touch: normally its namesake; an XCD that finds its own queue finished or foreign
 *     adopts queues nobody has claimed), so exactly one L2 sees all items of a polynomial even on a device that exposes
 *     fewer XCDs than eight.
 * Correctness never rests on placement assumptions: the XCD is read, and a hand-off only happens inside one XCD.
 * MEMORY-ORDER INVARIANT of the hand-off (team_kernel, team_product_kernel, team_dot kernels).  The signal is a relaxed
 * agent-scope atomic behind s_waitcnt vmcnt(0) + a workgroup barrier, the poll a relaxed agent-scope load in front of a
 * workgroup barrier; there is deliberately NO release/acquire fence (an agent-scope release is buffer_wbl2: it writes back
 * every dirty L2 line of the XCD, the neighbours' results included -- 8 us per hand-off) and no buffer_inv on the consumer.
 * That is sound because producer and consumer sit on ONE XCD, i.e. behind one L2, which is the point of coherence for
 * them (stores are complete in that L2 once vmcnt reaches 0; the per-CU vector cache is write-through), PROVIDED that no
 * CU's vector cache (TCP) can hold a stale copy of a line the consumer reads:
 *   (1) a line that a later pass of the same launch overwrites from ANOTHER CU is only ever read with sc0 sc1 loads,
 *       which do not allocate in the TCP (kAuxSc0Sc1: the inputs of the first pass);
 *   (2) every other load (nt: may allocate) of a line that is overwritten later in the launch is issued by the very item
 *       that overwrites it: items are block-aligned (a row item reads and writes exactly its own 2^12-point block, a
 *       column item its own 2 KiB row segments), so the only CU that may cache the old contents is the one whose own
 *       write-through stores replace them;
 *   (3) TCPs start a launch invalid, and the launch never reads a final output again.
 * Changing a cache policy or making items overlap in lines breaks this silently; tests/test_gpu_parity.py
 * (test_xcd_local_*) and tools/soak.py compare every polynomial with the per-pass path for that reason.
 * Reference precedent for finishing a sub-transform while its data is close: third_party/hexl/fwd-ntt-avx512.c:311-329.
 */
static_assert(kAuxSc0Sc1 == 17 && kAuxNt == 2 && kAuxSc1 == 16, "cache-policy encodings the hand-off invariant is written for");
struct TeamCtl {
  unsigned next[8][32];  /* per queue: next item; one 128-byte line each */
  unsigned owner[8][32]; /* per queue: 0 = unclaimed, else 1 + the XCD that processes it */
  unsigned done[1];      /* [polynomials] first-pass items finished (flexible) */
};

template <class A> struct KTeam {
  KArgs<A> k;      /* a, limbs[], limb_stride, logn; nblocks = polynomials PER LIMB */
  TeamCtl *ctl;    /* zeroed before the launch */
  uint32_t lag;    /* polynomials between a first-pass item and the second-pass items of the same queue */
  uint32_t nlimbs; /* MULTI kernels: limbs of the launch (polynomial v of the queues = limb * batch + polynomial) */
  uint64_t split_rcp; /* MULTI kernels: floor(2^64 / D) + 1 for the divisor D of team_split (batch, or nlimbs when poly_major): the
                       * quotient as a scalar multiply-high, exact for every v (v D < 2^64) -- a division would run in the VALU */
  uint32_t poly_major; /* MULTI kernels: v = polynomial * nlimbs + limb instead -- the numbering that follows the ADDRESSES when a
                        * polynomial's limbs lie side by side ([batch][limb][N]): the queues then walk memory as they do in the
                        * limb-major layout (which polynomials are in flight together decides the HBM channel mix) */
};
/* queue polynomial v -> (limb, polynomial inside the limb) */
__device__ __forceinline__ void team_split(uint32_t v, uint32_t batch, uint32_t nlimbs, uint32_t poly_major, uint64_t rcp, uint32_t &limb, uint32_t &pl)
{
  /* (v is wave-uniform: the multiply-high stays in scalar registers, see limb_params; rcp = floor(2^64 / D) + 1 gives the exact quotient) */
  const uint32_t quo = rcp ? (uint32_t)__umul64hi((uint64_t)v, rcp) : v; /* (rcp == 0: divisor 1) */
  if(poly_major) {
    pl   = quo;
    limb = v - pl * nlimbs;
  } else {
    limb = quo;
    pl   = v - limb * batch;
  }
}
inline uint64_t team_split_rcp(uint64_t divisor) { return divisor > 1 ? ~0ull / divisor + 1 : 0; /* (D = 1: the quotient is v itself, see callers) */ }

/* MULTI (several RNS limbs in one launch): the item's limb picks the tables, the constants and the slab.  The items of these
 * kernels fetch their tables per item anyway, so a limb that changes from item to item costs scalar loads only. */
template <class A, bool INV> __device__ __forceinline__ void team_limb(Params<A> &p, const KArgs<A> &k, uint32_t limb)
{
  const LimbRec<A> &r = k.limbs[limb];
  p.a   = k.a + (uint64_t)limb * k.limb_stride;
  p.tw  = INV ? r.tw_i : r.tw_f;
  p.tw8 = INV ? r.tw8_i : r.tw8_f;
  p.c   = r.c;
}

__device__ __forceinline__ uint32_t xcc_id()
{
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 7u;
}

constexpr int kTeamBlock = 12;  /* log2 of the row items (the block size below the column stages) */
constexpr int kTeamCols  = 256; /* adjacent columns of a column item = threads of a workgroup */

/* column item: leading stages [0, R) of a 2^logn-point polynomial on columns col of the 2^R x 2^(logn-R) view; the
 * thread's 2^R values sit 2^(logn-R) apart.  column_pass_thread (ntt_core.h) with S = 0, through a buffer descriptor so
 * that loads and stores carry a cache policy, twiddles through the scalar cache (their slots are compile-time here). */
template <class A, int R, bool INV, uint32_t MASK, int LDAUX, int STAUX>
__device__ __forceinline__ void team_column_item(uint64_t *poly, uint32_t col, uint32_t logn, const Params<A> &p, bool lazy_out)
{
  constexpr int  NE  = 1 << R;
  const uint32_t lsp = logn - R;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(poly, 0, (int)(8u << logn), 0x00020000);
  typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
  uint64_t raw[NE];
  static_for<0, NE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    raw[E]          = buffer_load_u64<LDAUX>(r, col * 8u, ((uint32_t)E << lsp) * 8u);
  });
  typename A::val x[NE];
  static_for<0, NE>([&](auto ee) { x[decltype(ee)::value] = A::template load<INV, false>(raw[decltype(ee)::value], p.c); });
  static_for<0, R>([&](auto jj) {
    constexpr int  J   = INV ? (R - 1 - decltype(jj)::value) : decltype(jj)::value;
    constexpr int  AB  = R - 1 - J;
    constexpr int  POS = INV ? (R - 1 - J) : J;
    constexpr bool RED = (MASK >> POS) & 1u;
    static_for<0, NE>([&](auto ee) {
      constexpr int E0 = decltype(ee)::value;
      if constexpr(((E0 >> AB) & 1) == 0) {
        constexpr int E1 = E0 | (1 << AB);
        if constexpr(INV && J == 0) {
          A::inv_bfly_last(x[E0], x[E1], p.c); /* global stage 0 ends the inverse transform: N^-1 folded in */
        } else {
          const typename A::tw w = load_tw<A, true>(p.tw, (1u << J) + (uint32_t)(E0 >> (R - J)));
          if constexpr(INV) {
            A::template inv_bfly<RED>(x[E0], x[E1], w, p.c);
          } else {
            A::template fwd_bfly<RED>(x[E0], x[E1], w, p.c);
          }
        }
      }
    });
  });
  static_for<0, NE>([&](auto ee) {
    constexpr int  E = decltype(ee)::value;
    const uint64_t u = out_word<A, INV, false>(x[E], lazy_out, p.c);
    v2u32          w2;
    w2.x = (unsigned)u;
    w2.y = (unsigned)(u >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(w2, r, (int)(col * 8u), (int)(((uint32_t)E << lsp) * 8u), STAUX);
  });
}

/* row item, forward: one 2^12-point block at position blk of its polynomial (the body of fused_kernel's persistent
 * loop without the prefetch: the table of the second-to-last group and the last group's twiddles are per position,
 * so they are fetched per item -- from the L2, the whole batch shares them) */
template <class A, int KSH, int LDAUX, int STAUX>
__device__ __forceinline__ void team_row_item_fwd(uint64_t *base, uint32_t blk, uint32_t tid, const Params<A> &p,
                                                  typename A::val *lds, typename A::ctw *tabl)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, flavor_of<A>()>;
  constexpr int GL   = P::NG - 1;
  constexpr uint32_t MASK = fused_mask<A, LOGN, false, KSH>();
  static_assert(G::TBL(GL - 1) > 0 && G::TBL(GL) == 0, "twiddle placement this item assumes");
  const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  uint64_t raw[kE];
  prefetch_first<LOGN, LDAUX>(raw, tid, base);
  typename A::ctw pre[4][kE / 2];
  preload_group_tw<A, LOGN, GL>(pre, tid, blk, p);
  fill_lds_tables<A, LOGN, false>(tabl, p, blk, tid);
  /* (the table is read in the second-to-last group; the cross-wave exchange in front of it has two workgroup barriers) */
  static_assert(!P::WAVE_LOCAL(0, 1), "the first exchange must cross waves: its barriers publish the LDS table");
  typename A::val x[kE];
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
  /* whole 128-byte lines per store instruction: a write-through store of half a line costs a full line's write */
  store_last_whole_lines<A, LOGN, false, STAUX>(x, tid, base, p.c, p.lazy != 0);
}

/* row item, inverse: the mirror image (the block pass comes first in the inverse transform) */
template <class A, int KSH, int LDAUX, int STAUX>
__device__ __forceinline__ void team_row_item_inv(uint64_t *base, uint32_t blk, uint32_t tid, const Params<A> &p,
                                                  typename A::val *lds, typename A::ctw *tabl, bool mid_lazy)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, true, flavor_of<A>()>;
  constexpr int GL   = P::NG - 1;
  constexpr uint32_t MASK = fused_mask<A, LOGN, true, KSH>() | (A::kWide52 ? kCanonInFlag : 0u); /* not the pass that ends the transform; canonical inputs */
  const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  uint64_t raw[kE];
  prefetch_last<LOGN, LDAUX>(raw, tid, base);
  constexpr bool IPRE = stage_is_compact<A, LOGN, true>(GL, 0) && P::R(GL) < 4 && G::TBL(GL) == 0 && KSH != 1;
  typename A::ctw pre[4][kE / 2];
  if constexpr(IPRE) preload_group_tw<A, LOGN, GL>(pre, tid, blk, p);
  fill_lds_tables<A, LOGN, true>(tabl, p, blk, tid);
  /* (read after the exchange between the last two groups -- wave-local -- so the table needs its own barrier here) */
  __syncthreads();
  typename A::val x[kE];
  convert_inputs<A, true>(x, raw, p.wide != 0, p.c);
  if constexpr(IPRE) {
    run_group_preloaded<A, LOGN, GL, MASK, true>(x, pre, p);
  } else {
    run_group<A, LOGN, GL, true, MASK>(x, tid, blk, p);
  }
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    exchange<A, LOGN, GI, GI - 1>(x, tid, lds);
    if constexpr(G::TBL(GI - 1) > 0) {
      run_group<A, LOGN, GI - 1, true, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI - 1));
    } else {
      run_group<A, LOGN, GI - 1, true, MASK>(x, tid, blk, p);
    }
  });
  uint64_t out[kE];
  static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = out_word<A, true, false>(x[decltype(ee)::value], mid_lazy, p.c); });
  buffer_store_first_raw<LOGN, STAUX>(out, tid, base);
}

/* the row items of a policy without compact twiddles and LDS tables (the wide integer policy): the block body of
 * fused_kernel's plain loop -- 16-byte records through the scalar cache and the L1/L2, which the whole batch shares -- with
 * the cache policies of the items above (the launch's memory-order invariant does not depend on the arithmetic) */
template <class A, int KSH, int LDAUX, int STAUX>
__device__ __forceinline__ void team_row_item_fwd_plain(uint64_t *base, uint32_t blk, uint32_t tid, const Params<A> &p, typename A::val *lds)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, false, KSH>();
  uint64_t raw[kE];
  prefetch_first<LOGN, LDAUX>(raw, tid, base);
  typename A::val x[kE];
  convert_inputs<A, false>(x, raw, false, p.c);
  run_group<A, LOGN, 0, false, MASK>(x, tid, blk, p);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = decltype(gg)::value;
    exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
    run_group<A, LOGN, GI + 1, false, MASK>(x, tid, blk, p);
  });
  store_last_whole_lines<A, LOGN, false, STAUX>(x, tid, base, p.c, p.lazy != 0);
}
template <class A, int KSH, int LDAUX, int STAUX>
__device__ __forceinline__ void team_row_item_inv_plain(uint64_t *base, uint32_t blk, uint32_t tid, const Params<A> &p, typename A::val *lds,
                                                        bool mid_lazy)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, true, KSH>(); /* not the pass that ends the transform */
  uint64_t raw[kE];
  prefetch_last<LOGN, LDAUX>(raw, tid, base);
  typename A::val x[kE];
  convert_inputs<A, true>(x, raw, p.wide != 0, p.c);
  run_group<A, LOGN, P::NG - 1, true, MASK>(x, tid, blk, p);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    exchange<A, LOGN, GI, GI - 1>(x, tid, lds);
    run_group<A, LOGN, GI - 1, true, MASK>(x, tid, blk, p);
  });
  uint64_t out[kE];
  static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = out_word<A, true, false>(x[decltype(ee)::value], mid_lazy, p.c); });
  buffer_store_first_raw<LOGN, STAUX>(out, tid, base);
}

template <class A, int LEAD, bool INV, int KSH, bool MULTI = false>
__global__ void __launch_bounds__(256, 4) team_kernel(const KTeam<A> kt)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, INV, flavor_of<A>()>;
  static_assert((A::kCompact || A::kIntWide) && P::T == kTeamCols && LEAD >= 3 && LEAD <= 5,
                "built for the FP64 policies and the wide integer policy on 2^12-point blocks, N = 2^15..2^17");
  __shared__ typename A::val lds[P::LDS_ELEMS + G::LDS_TW];
  __shared__ unsigned        s_k, s_k2[2];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds + P::LDS_ELEMS);
  const uint32_t         tid  = threadIdx.x;
  uint32_t               bid_, gdim_, limb_;
  Params<A>              p = limb_params<A, INV, false>(kt.k, bid_, gdim_, limb_);
  p.s0                     = LEAD;
  constexpr uint32_t CMASK = column_mask<A, LEAD, INV, KSH>();
  constexpr bool     MID_LAZY = !A::kTracksBounds; /* words between the passes: canonical for the FP64 policies */
  const uint32_t logn  = LOGN + LEAD;
  const uint32_t batch = (uint32_t)p.nblocks;      /* polynomials per limb */
  const uint32_t total = MULTI ? batch * kt.nlimbs : batch;
  constexpr uint32_t NCOL = 1u << (LOGN - 8);      /* column items per polynomial: 2^(m - LEAD) columns / 256 */
  constexpr uint32_t NROW = 1u << LEAD;            /* row items per polynomial */
  constexpr uint32_t NA   = INV ? NROW : NCOL;     /* first-pass items */
  constexpr uint32_t NB   = INV ? NCOL : NROW;
  TeamCtl *const ctl = kt.ctl;
  const uint32_t lag = kt.lag;
  const uint32_t my  = xcc_id();
  for(uint32_t qq = 0; qq < 8; qq++) {
    const uint32_t q = (my + qq) & 7u; /* own queue first, then whatever nobody claimed */
    if(tid == 0) {
      const unsigned prev = atomicCAS(&ctl->owner[q][0], 0u, my + 1u);
      s_k                 = (prev == 0u || prev == my + 1u) ? 1u : 0u;
    }
    __syncthreads();
    const bool mine = s_k != 0;
    __syncthreads();
    if(!mine) continue;
    /* Every lane-0 block of this loop is followed at once by a workgroup barrier.  A lane-0 block at the END of the body
     * (the completion signal used to sit there) ends up next to the loop's back edge, and the compiler then lets lane 0
     * leave the loop "early" while lanes 1-63 of its wave wait at the next iteration's barrier for the item only lane 0
     * can fetch: the first version of this kernel hung on its first items exactly like that.  So the signal of a finished
     * first-pass item is carried into the next iteration and issued by the same lane-0 block that fetches the next item. */
    constexpr uint32_t kNoSignal = 0xffffffffu;
    uint32_t           sig       = kNoSignal;
    for(uint32_t it = 0;; it ^= 1u) {
      /* the item index travels through one of two LDS words in turn, so that one barrier per fetch is enough (lane 0
       * writes the other word next time: nobody can still be reading it, everybody has passed this barrier since) */
      if(tid == 0) {
        if(sig != kNoSignal) __hip_atomic_fetch_add(&ctl->done[sig], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_k2[it] = atomicAdd(&ctl->next[q][0], 1u);
      }
      sig = kNoSignal;
      __syncthreads();
      /* queue entry -> (pass, item, polynomial): ntt_core.h team_decode, the function tests/test_team_protocol.py simulates */
      const TeamItem ti = team_decode(uniform_u32(s_k2[it]), q, total, lag, NA, NB, 0u);
      if(ti.stop) break;
      if(!ti.valid) continue;
      const bool     second = ti.pass != 0;
      const uint32_t item   = ti.item;
      const uint32_t pidx   = ti.v;
      uint32_t       pl     = pidx; /* the polynomial inside its limb */
      if constexpr(MULTI) {
        uint32_t limb;
        team_split(pidx, batch, kt.nlimbs, kt.poly_major, kt.split_rcp, limb, pl);
        team_limb<A, INV>(p, kt.k, limb);
      }
      uint64_t *poly = p.a + poly_offset<true>((uint64_t)pl, p.pstride, p.ptab);
      if(second) {
        if(tid == 0) {
#ifdef NTT_TEAM_WATCHDOG
          /* development builds: a wait that lasts longer than about a second is recorded (owner[q][1..3]) and abandoned,
           * so that a protocol error shows up as a wrong result with a diagnosis instead of a hung GPU */
          unsigned spins = 0;
          while(__hip_atomic_load(&ctl->done[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NA) {
            __builtin_amdgcn_s_sleep(8);
            if(++spins > (1u << 15)) {
              ctl->owner[q][1] = pidx + 1u;
              ctl->owner[q][2] = __hip_atomic_load(&ctl->done[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ctl->owner[q][3] = s_k2[it];
              break;
            }
          }
#else
          while(__hip_atomic_load(&ctl->done[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NA) __builtin_amdgcn_s_sleep(8);
#endif
        }
        __syncthreads(); /* (also keeps every later load of the workgroup behind the poll) */
      }
      const bool row = second != INV;
      if(!row) {
        /* forward: inputs -> intermediate (kept dirty in the L2); inverse: intermediate -> final (write-through) */
        if constexpr(INV) team_column_item<A, LEAD, true, CMASK, kAuxNt, kAuxSc1>(poly, item * kTeamCols + tid, logn, p, p.lazy != 0);
        else team_column_item<A, LEAD, false, CMASK, kAuxSc0Sc1, 0>(poly, item * kTeamCols + tid, logn, p, MID_LAZY);
      } else {
        uint64_t *base = poly + ((uint64_t)item << LOGN);
        /* (inverse inputs arrive as 16-byte loads in runs of four coefficients per lane: two instructions share every 128-byte
         * line, so these loads must be allowed to hit the L2 -- nt; with the cache-bypassing policy of the forward
         * inputs every line crossed the fabric twice) */
        if constexpr(!A::kCompact) {
          (void)tabl;
          if constexpr(INV) team_row_item_inv_plain<A, KSH, kAuxNt, 0>(base, item, tid, p, lds, MID_LAZY);
          else team_row_item_fwd_plain<A, KSH, kAuxNt, kAuxSc1>(base, item, tid, p, lds);
        } else if constexpr(INV) team_row_item_inv<A, KSH, kAuxNt, 0>(base, item, tid, p, lds, tabl, MID_LAZY);
        else team_row_item_fwd<A, KSH, kAuxNt, kAuxSc1>(base, item, tid, p, lds, tabl);
      }
      if(!second) {
        /* the item's stores have completed (every wave waits for its own, the barrier collects the waves) before the
         * counter moves -- at the top of the next iteration */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sig = pidx;
      }
    }
  }
}

} /* namespace ntt */
