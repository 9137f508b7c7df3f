/*
 * ntt_core.h -- the butterfly network shared by the gfx950 kernels and the CPU
 * emulation used by the unit tests (tests/emu).
 *
 * Transform semantics are the reference's: forward =
 * Cooley-Tukey, natural order in, bit-reversed order out, in place; stage s
 * (s = 0..m-1) pairs elements whose indices differ in bit m-1-s and uses twiddle
 * slot 2^s + (i >> (m-s)) of the bit-reversed power table
 * (reference src/ntt_reference.c:17-30).  Inverse = Gentleman-Sande over the
 * same stages in reverse with the inverse table, N^-1 folded into stage 0
 * (src/ntt_reference.c:41-65).
 *
 * GPU decomposition.  A "block" of 2^LOGN consecutive
 * coefficients is transformed by 2^(LOGN-4) threads that hold 16 coefficients
 * each in registers.  The LOGN stages are cut into groups of <= 4 stages; inside
 * a group every butterfly is register-to-register (the in-thread radix-16 tile
 * echoes reference src/ntt_radix4x4.c:54-78); between groups the block is
 * re-distributed through LDS in a "reader-linear" layout
 *     addr = e_r * (T + PAD) + t_r
 * (e_r = register slot, t_r = thread of the *next* group) so reads are always
 * conflict-free and PAD (found at compile time) makes the scattered
 * ds_write_b64 of the previous group conflict-free as well.
 *
 *   first group : top 4 index bits in-thread (R0 active + passengers below),
 *                 coalesced 8-byte global loads, block-uniform twiddles
 *   middle      : 4 active bits [lo,lo+4), wave-uniform twiddles while lo >= 6
 *   last group  : bits [0,RL) in-thread, RL = 2 (LOGN even) or 1; lanes sit on
 *                 consecutive pairs/quads so the final stores are 16 B/lane
 */
#pragma once
#include <stdint.h>
#include <type_traits>
#include <utility>

#include "ntt_arith.h"
#include "ntt_passplan.h" /* kFusedLarge: the block size the one-pass 2^15 schedule is written for */

namespace ntt {

constexpr int kLE = 4;       /* log2 coefficients per thread */
constexpr int kE  = 1 << kLE;

template <int I, int N, class F> NTT_HD void static_for(F &&f)
{
  if constexpr(I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

/* Lane-bit order per (LOGN, group): thread bit t < NL of group g addresses index
 * bit kLaneBits[LOGN-6][g][t].  Found by tools/lds_layout_search.py so that, with
 * one pad element per LDS row, every ds_write_b64 of every exchange (both
 * directions) hits 16 distinct 8-byte bank columns per 16-lane group; the reads
 * are linear and conflict-free by construction.  -1 = unused. */
constexpr int8_t kLaneBits[9][4][6] = {
  /* LOGN  6 */ {{0, 1, -1, -1, -1, -1}, {2, 3, -1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}},
  /* LOGN  7 */ {{0, 1, 2, -1, -1, -1}, {5, 6, 0, -1, -1, -1}, {1, 2, 3, -1, -1, -1}, {-1, -1, -1, -1, -1, -1}},
  /* LOGN  8 */ {{0, 1, 2, 3, -1, -1}, {6, 7, 0, 1, -1, -1}, {2, 3, 4, 5, -1, -1}, {-1, -1, -1, -1, -1, -1}},
  /* LOGN  9 */ {{0, 1, 2, 3, 4, -1}, {6, 7, 8, 0, 5, -1}, {1, 2, 3, 4, 5, -1}, {-1, -1, -1, -1, -1, -1}},
  /* LOGN 10 */ {{0, 1, 2, 3, 4, 5}, {8, 9, 0, 1, 6, 7}, {2, 3, 4, 5, 6, 7}, {-1, -1, -1, -1, -1, -1}},
  /* LOGN 11 */ {{0, 1, 2, 3, 4, 5}, {0, 1, 2, 3, 4, 9}, {6, 7, 8, 0, 5, 9}, {2, 3, 4, 6, 1, 5}},
  /* LOGN 12 */ {{0, 1, 2, 3, 4, 5}, {0, 1, 2, 3, 4, 5}, {8, 9, 0, 1, 6, 7}, {2, 3, 4, 5, 6, 7}},
  /* LOGN 13 */ {{0, 1, 2, 3, 4, 5}, {0, 1, 2, 3, 4, 9}, {6, 7, 8, 0, 5, 9}, {2, 3, 4, 6, 1, 5}},
  /* LOGN 14 */ {{0, 1, 2, 3, 4, 5}, {0, 1, 2, 3, 4, 5}, {8, 9, 0, 1, 6, 7}, {2, 3, 4, 5, 6, 7}},
};
constexpr int kLdsPad = 1;

/* ------------------------------------------------------------------ */
/* compile-time plan                                                   */
/* ------------------------------------------------------------------ */
template <int LOGN> struct Plan {
  static_assert(LOGN >= 6 && LOGN <= 14, "fused block size out of range");
  static constexpr int LT   = LOGN - kLE; /* log2 threads per block */
  static constexpr int T    = 1 << LT;
  static constexpr int RL   = (LOGN & 1) ? 1 : 2;
  static constexpr int REM  = LOGN - RL;
  static constexpr int R0   = (REM % 4 == 0) ? 4 : (REM % 4);
  static constexpr int NMID = (REM - R0) / 4;
  static constexpr int NG   = NMID + 2;
  static constexpr int NL   = LT < 6 ? LT : 6; /* lane bits of one block */

  static constexpr int R(int g) { return g == 0 ? R0 : (g == NG - 1 ? RL : 4); }
  /* first local stage of group g */
  static constexpr int S(int g)
  {
    int s = 0;
    for(int i = 0; i < g; i++) s += R(i);
    return s;
  }
  /* lowest active index bit of group g */
  static constexpr int LO(int g) { return LOGN - S(g) - R(g); }

  /* in-thread slot bit b (0..3) -> index bit */
  static constexpr int EB(int g, int b)
  {
    if(g == 0) return LOGN - 4 + b;
    if(g == NG - 1) return b < RL ? b : RL + NL + (b - RL);
    return LO(g) + b;
  }
  /* thread bit t (0..LT-1) -> index bit */
  static constexpr int TB(int g, int t)
  {
    if(t < NL) return kLaneBits[LOGN - 6][g][t]; /* lane bits: tuned order */
    return g == 0 ? t : t + 4;                   /* wave bits              */
  }
  /* the untuned assignment kLaneBits permutes (kept for the layout tests) */
  static constexpr int TB_DEFAULT(int g, int t)
  {
    if(g == 0) return t;
    if(g == NG - 1) return t < NL ? RL + t : t + 4;
    return t < LO(g) ? t : t + 4;
  }
  /* slot bit that local stage j of group g toggles */
  static constexpr int ABIT(int g, int j)
  {
    if(g == 0) return (4 - R0) + (R0 - 1 - j);
    if(g == NG - 1) return RL - 1 - j;
    return 3 - j;
  }
  static constexpr uint32_t IOFF(int g, int e)
  {
    uint32_t v = 0;
    for(int b = 0; b < 4; b++) v |= ((e >> b) & 1u) << EB(g, b);
    return v;
  }
  static constexpr uint32_t IBASE(int g, uint32_t t)
  {
    uint32_t v = 0;
    for(int b = 0; b < LT; b++) v |= ((t >> b) & 1u) << TB(g, b);
    return v;
  }
  /* mask of the index bits carried by the thread id in group g */
  static constexpr uint32_t THREAD_BITS(int g)
  {
    uint32_t v = 0;
    for(int b = 0; b < LT; b++) v |= 1u << TB(g, b);
    return v;
  }
  /* inverse maps: index -> (thread, slot) of group g */
  static constexpr uint32_t THREAD_OF(int g, uint32_t i)
  {
    uint32_t v = 0;
    for(int b = 0; b < LT; b++) v |= ((i >> TB(g, b)) & 1u) << b;
    return v;
  }
  static constexpr uint32_t SLOT_OF(int g, uint32_t i)
  {
    uint32_t v = 0;
    for(int b = 0; b < 4; b++) v |= ((i >> EB(g, b)) & 1u) << b;
    return v;
  }
  /* LDS layout read by group gr: addr(i) = slot*ROW + thread.  PAD makes the
   * writer (group gw) conflict-free: ds_write_b64 is serviced in 16-lane groups
   * over 32 four-byte banks, i.e. 16 eight-byte columns. */
  static constexpr bool pad_ok(int gw, int gr, int pad)
  {
    const int row = T + pad;
    for(uint32_t t0 = 0; t0 < (uint32_t)T; t0 += 16) {
      uint32_t seen = 0;
      const int n   = T < 16 ? T : 16;
      for(int l = 0; l < n; l++) {
        const uint32_t i = IBASE(gw, t0 + l);
        const uint32_t a = SLOT_OF(gr, i) * row + THREAD_OF(gr, i);
        const uint32_t c = a & 15u;
        if(seen & (1u << c)) return false;
        seen |= 1u << c;
      }
    }
    return true;
  }
  /* one pad for every exchange of the plan (both directions), so that all
   * layouts share the same rows and a wave-local exchange only ever touches
   * the columns owned by its own wave */
  static constexpr int PAD() { return kLdsPad; }
  static constexpr bool layout_conflict_free()
  {
    bool ok = true;
    for(int g = 0; g + 1 < NG; g++) ok = ok && pad_ok(g, g + 1, PAD()) && pad_ok(g + 1, g, PAD());
    return ok;
  }
  static constexpr int ROW       = T + PAD();
  static constexpr int LDS_ELEMS = kE * ROW; /* per block */
  /* an exchange stays inside one wave when the threads' wave bits address
   * the same index bits on both sides */
  static constexpr bool WAVE_LOCAL(int ga, int gb)
  {
    for(int t = 6; t < LT; t++) {
      if(TB(ga, t) != TB(gb, t)) return false;
    }
    return true;
  }
  /* group and local stage that own local stage sl of the block */
  static constexpr int GROUP_OF(int sl)
  {
    int g = 0;
    while(g + 1 < NG && S(g + 1) <= sl) g++;
    return g;
  }
  /* B-th butterfly of local stage j of group g: its lower slot, and the first
   * butterfly of that stage that uses the same twiddle */
  static constexpr int BFLY_E0(int g, int j, int b)
  {
    const int ab = ABIT(g, j);
    return ((b >> ab) << (ab + 1)) | (b & ((1 << ab) - 1));
  }
  static constexpr int BFLY_FIRST(int g, int j, int b)
  {
    const int      sh  = LOGN - (S(g) + j);
    const uint32_t off = IOFF(g, BFLY_E0(g, j, b)) >> sh;
    for(int k = 0; k < b; k++) {
      if((IOFF(g, BFLY_E0(g, j, k)) >> sh) == off) return k;
    }
    return b;
  }
  /* number of consecutive table slots [0,span) the 8 butterflies of local stage j
   * of group g use relative to the thread's base slot, or 0 if they are not a
   * dense range (then they are fetched one by one) */
  static constexpr int UNIFORM_SPAN(int g, int j)
  {
    const int sh  = LOGN - (S(g) + j);
    uint32_t  mx  = 0;
    uint32_t  set = 0;
    for(int b = 0; b < kE / 2; b++) {
      const uint32_t off = IOFF(g, BFLY_E0(g, j, b)) >> sh;
      if(off >= 32) return 0;
      set |= 1u << off;
      mx = off > mx ? off : mx;
    }
    return set == ((mx + 1 >= 32) ? 0xffffffffu : ((1u << (mx + 1)) - 1)) ? (int)(mx + 1) : 0;
  }
  /* twiddle slot of a stage is wave-uniform when no lane bit reaches the
   * shifted-in part */
  static constexpr bool TW_UNIFORM(int g, int j)
  {
    if(LT < 6) return false; /* several blocks share a wave */
    const int sh = LOGN - (S(g) + j);
    for(int t = 0; t < 6; t++) {
      if(TB(g, t) >= sh) return false;
    }
    return true;
  }
};

/* ------------------------------------------------------------------ */
/* kernel parameters                                                   */
/* ------------------------------------------------------------------ */
template <class A> struct Params {
  uint64_t *             a;       /* [batch][N] coefficients, in place            */
  const typename A::tw * tw;      /* N records, bit-reversed power order          */
  const typename A::ctw *tw8;     /* same slots, compact 8-byte form (forward FP64 only) */
  typename A::consts     c;
  uint32_t               logn;    /* log2 N of the whole transform                */
  uint32_t               s0;      /* global stages handled before (fwd) / after (inv) this pass */
  uint32_t               wide;    /* inputs may be lazy ([0,8q)) instead of [0,q) */
  uint32_t               lastinv; /* inverse: this pass ends with global stage 0  */
  uint32_t               lazy;    /* outputs of the pass that ends a transform stay in the reference's lazy range */
  uint64_t               nblocks; /* batch * 2^s0 blocks of 2^LOGN                */
  uint64_t               pstride; /* words between consecutive polynomials of this limb: N for the dense [batch][N] slab, more
                                   * for a caller-native layout ([polynomial][limb][N]: limbs * N) -- see block_offset */
  const uint64_t *       ptab;    /* pointer batches (ntt_transform_ptrs, ntt_transform_dev_ptrs): device table, one entry per polynomial =
                                   * the ADDRESS of its limb 0 (a plain array of device pointers; the launch passes a = the limb's offset
                                   * alone); null = the arithmetic progression above -- see poly_offset */
};

/* Where block b of a pass starts, in words from the limb's first coefficient.  A pass over `batch` polynomials of 2^logn points
 * works on batch * 2^s0 blocks of 2^LOGN points (s0 = logn - LOGN leading stages belong to column passes): block b is position
 * b mod 2^s0 of polynomial b >> s0, and polynomial p of the limb starts p * pstride words in.  The dense layout the reference's
 * batching precedent implies (fwd_ntt_ref_harvey_lazy_dbl's a1[], a2[] side by side, include/ntt_reference.h:44-49) has
 * pstride = N and this is b << LOGN; SURVEY 8(d)'s [batch][prime][N] has pstride = primes * N.  One function for the kernels,
 * the host's chunking and the CPU emulation (tests/emu). */
template <int LOGN> NTT_HD uint64_t block_offset(uint64_t b, uint32_t s0, uint64_t pstride)
{
  return (b >> s0) * pstride + ((b & ((1ull << s0) - 1ull)) << LOGN);
}

/* Where polynomial `poly` of the limb starts, in words from Params::a: the progression poly * pstride, or -- a batch of separately
 * held arrays, the reference's own batch form (fwd_ntt_ref_harvey_lazy_dbl(a1[], a2[], ...), include/ntt_reference.h:44-49,
 * src/ntt_reference.c:71-91) for any number of polynomials placed anywhere -- entry `poly` of a device table.  UNIFORM: `poly` is the
 * same for the whole wave (derived from the workgroup id and loop counters): the entry is read through the constant address space,
 * one s_load_dwordx2 into scalar registers like a wave-uniform twiddle, and everything derived from it (the block's buffer
 * descriptor) stays scalar.  The table is written before the launch and never by it. */
template <bool UNIFORM = true> NTT_HD uint64_t poly_offset(uint64_t poly, uint64_t pstride, const uint64_t *ptab)
{
  if(ptab) {
    /* entries are byte addresses of 8-byte aligned polynomials; Params::a is the limb's offset from a null base: words */
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr(UNIFORM) {
      typedef const uint64_t __attribute__((address_space(4))) * ctab_t;
      return ((ctab_t)(uintptr_t)ptab)[poly] >> 3;
    }
#endif
    return ptab[poly] >> 3;
  }
  return poly * pstride;
}
/* block_offset for the kernels that serve pointer batches (the transform kernels): block b mod 2^s0 of polynomial b >> s0 */
template <int LOGN> NTT_HD uint64_t block_offset(uint64_t b, uint32_t s0, uint64_t pstride, const uint64_t *ptab)
{
  return poly_offset<true>(b >> s0, pstride, ptab) + ((b & ((1ull << s0) - 1ull)) << LOGN);
}

NTT_HD uint32_t uniform_u32(uint32_t v)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
#else
  return v;
#endif
}

/* Twiddle fetch.  UNIFORM: the slot is the same for the whole wave (first group
 * and every group whose active bits lie above the lane bits): on the device the
 * record is read through the constant address space so the compiler emits one
 * s_load_dwordx4 into SGPRs instead of 64 identical vector-lane requests -- the
 * vector memory pipe (TA) was the bottleneck with per-lane loads (profiles/r01). */
template <class A, bool UNIFORM, int G = 0> NTT_HD typename A::tw load_tw(const typename A::tw *tab, uint32_t idx)
{
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr(UNIFORM) {
    typedef const typename A::tw __attribute__((address_space(4))) * ctab_t;
    ctab_t ct = (ctab_t)(uintptr_t)tab;
    return ct[idx];
  }
#endif
  return tab[idx];
}

/* ------------------------------------------------------------------ */
/* one register-resident group of stages                               */
/* ------------------------------------------------------------------ */
/*
 * x[16]  : the thread's coefficients (slot e <-> index IBASE(t)+IOFF(e))
 * t      : thread id inside the block;  blk: block id inside the polynomial
 * MASK   : ArithF64 reduction schedule, bit = processing position of the stage
 * MASK & kLastInvFlag (inverse only): local stage 0 is global stage 0 and folds N^-1
 *          (reference src/ntt_reference.c:55-65)
 */
/* Per-lane (non-uniform) twiddles of the FP64 policy are fetched in
 * the compact 8-byte form: half the bytes, half the landing registers, and the
 * two twiddles of a last-stage pair become one 16-byte load. */
template <class A, int LOGN, bool INV> constexpr bool stage_is_compact(int g, int j)
{
  (void)INV;
  return A::kCompact && !Plan<LOGN>::TW_UNIFORM(g, j);
}

/* table[idx] with a 32-bit byte offset: tables are at most 2^28 records, so the
 * offset fits and the load can use the scalar-base + 32-bit-lane-offset form
 * instead of a 64-bit per-lane address (two carry-chained VALU adds and a hazard nop) */
template <class T> NTT_HD const T &at32(const T *base, uint32_t idx)
{
  return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (uint32_t)(idx * (uint32_t)sizeof(T)));
}

/* pointer to the LDS-resident compact twiddle table: kept in the LDS address
 * space on the device so that reads are ds_read_b64 -- a generic pointer (e.g. a
 * run-time select between LDS and global) turns them into flat loads, which
 * also occupy the vector-memory pipe */
#if defined(__HIP_DEVICE_COMPILE__)
template <class A> using lds_ctw_ptr = const typename A::ctw __attribute__((address_space(3))) *;
#else
template <class A> using lds_ctw_ptr = const typename A::ctw *;
#endif

/* One stage's twiddles for the thread's 8 butterflies (ascending E0 order).
 * Compact stages keep the 8-byte value in registers and rebuild w/q at the
 * point of use; the others hold full records (SGPRs when wave-uniform).  Only
 * the first butterfly of every distinct slot actually loads. */
template <class A> struct StageTw {
  typename A::tw  f[kE / 2];
  typename A::ctw c[kE / 2];
};

/* MIRROR (with LTW, whole-polynomial blocks): an inverse stage served from the FORWARD table.  In the bit-reversed
 * power layout w^-1 at slot 2^s + j equals -w at slot 2^(s+1) - 1 - j (root^N = -1, and complementing j
 * complements its bit reversal), i.e. the inverse twiddles of a stage are the forward ones in reverse order and
 * negated.  The sign is absorbed by the butterfly (A::inv_bfly_mirror multiplies y - x instead of x - y); the
 * reversed slot is position C - pos of the transposed LDS table, C = 3 * (2^J - 1) * 2^S(G) + 2^S(G) - 1.  This
 * lets the fused product kernel run its inverse half without a second 30 KB table in LDS. */
template <class A, int LOGN, int G, int J, bool INV, bool LTW = false, bool MIRROR = false>
NTT_HD void load_stage_tw(StageTw<A> &w, uint32_t ib, uint32_t blk, const Params<A> &p,
                          lds_ctw_ptr<A> ltw = nullptr)
{
  static_assert(!MIRROR || LTW, "mirrored twiddles come from the LDS table");
  using P           = Plan<LOGN>;
  constexpr int SL  = P::S(G) + J;
  constexpr int SH  = LOGN - SL;
  const uint32_t gs = p.s0 + SL;
  uint32_t       tb = (1u << gs) + (blk << SL) + (ib >> SH);
  if constexpr(P::TW_UNIFORM(G, J)) tb = uniform_u32(tb);
  /* LDS table (second-to-last group): stored TRANSPOSED per stage, position
   *   (2^J - 1) * 2^S(G)  +  u * 2^S(G)  +  prefix
   * (u = slot offset inside the stage, prefix = the thread's index bits above the
   * group = ib >> (LOGN - S(G))), so that neighbouring lanes -- which differ in the
   * prefix -- read neighbouring 8-byte words: conflict-free ds_read_b64, where the
   * natural order (prefix * 2^J + u) put them 2^J words apart (2-way conflicts). */
  /* In general both the thread's base index ib and the slot offset contribute to prefix and
   * u (their bit sets are disjoint, so the fields simply add): middle groups have all of u in
   * the slot and all of the prefix in the thread, a last group of fewer than four stages
   * also carries prefix bits in its slots. */
  constexpr uint32_t UMASK   = (1u << J) - 1u;
  constexpr bool     IB_IN_U = ((P::THREAD_BITS(G) >> SH) & UMASK) != 0;
  uint32_t           tl      = (uint32_t)(UMASK << P::S(G)) + (ib >> (LOGN - P::S(G)));
  if constexpr(IB_IN_U) tl += ((ib >> SH) & UMASK) << P::S(G);
#if defined(__HIP_DEVICE_COMPILE__)
  /* wave-uniform stage: its 2^J records are consecutive slots -> fetch them as
   * ONE aggregate through the constant address space (s_load_dwordx4/8/16), so a
   * group issues 4-5 scalar loads up front instead of 15 load/wait pairs */
  if constexpr(P::TW_UNIFORM(G, J) && P::UNIFORM_SPAN(G, J) > 0) {
    constexpr int NREC = P::UNIFORM_SPAN(G, J);
    struct alignas(16) Block {
      typename A::tw r[NREC];
    };
    typedef const Block __attribute__((address_space(4))) * cblk_t;
    const Block blkrec = *(cblk_t)(uintptr_t)(p.tw + tb);
    static_for<0, kE / 2>([&](auto bb) {
      constexpr int B = decltype(bb)::value;
      if constexpr(P::BFLY_FIRST(G, J, B) == B) {
        constexpr uint32_t OFF = P::IOFF(G, P::BFLY_E0(G, J, B)) >> SH;
        w.f[B]                 = blkrec.r[OFF];
      }
    });
    return;
  }
#endif
  static_for<0, kE / 2>([&](auto bb) {
    constexpr int B = decltype(bb)::value;
    if constexpr(P::BFLY_FIRST(G, J, B) == B) {
      constexpr uint32_t OFF = P::IOFF(G, P::BFLY_E0(G, J, B)) >> SH;
      if constexpr(stage_is_compact<A, LOGN, INV>(G, J)) {
        /* ltw: this group's slice of the compact table, resident in LDS, laid out
         * by the slot the stage would use in a stand-alone 2^LOGN transform
         * (block prefix and leading stages removed), minus the group's first slot */
        if constexpr(LTW) {
          constexpr uint32_t OFFT = ((OFF & UMASK) << P::S(G)) + (OFF >> J);
          if constexpr(MIRROR) {
            constexpr uint32_t CJ = 3u * (UMASK << P::S(G)) + ((1u << P::S(G)) - 1u);
            w.c[B]                = ltw[CJ - (tl + OFFT)];
          } else {
            w.c[B] = ltw[tl + OFFT];
          }
        } else {
          w.c[B] = at32(p.tw8, tb + OFF);
        }
      } else {
        w.f[B] = load_tw<A, P::TW_UNIFORM(G, J), G>(p.tw, tb + OFF);
      }
    }
  });
}

/* full record of butterfly B (non-compact stages only) */
template <class A, int LOGN, int G, int J, bool INV, int B>
NTT_HD typename A::tw stage_tw(const StageTw<A> &w, const typename A::consts &c)
{
  constexpr int F = Plan<LOGN>::BFLY_FIRST(G, J, B);
  return w.f[F];
}

/* ------------------------------------------------------------------ */
/* which butterflies reduce (ArithF64)                                   */
/* ------------------------------------------------------------------ */
/*
 * Forward (Cooley-Tukey): both outputs of a butterfly carry the same bound, so a
 * stage either reduces its eight unmultiplied operands or none: MASK bit per
 * processing position (f64_schedule).
 *
 * Inverse (Gentleman-Sande): the sum output doubles the bound while the product
 * output comes back near q/2, and the next stage pairs sums with sums and
 * products with products (the partner differs only in a bit not yet processed).
 * Bounds are therefore tracked PER SLOT inside a group -- all slots of a thread
 * share the worst case at a group boundary, where the partner's history is not
 * known at compile time -- and a sum is reduced only when keeping it would push
 * the next sum past the exactness limit.  At q ~ 2^51 this reduces 73 of the 112
 * sums of a 2^14 block instead of all of them.  MASK = kRedPlanFlag | KSH selects
 * this plan; the very last stage never reduces (its outputs are canonicalised or
 * multiplied by N^-1, both exact below 2^53).
 */
constexpr uint32_t kRedPlanFlag = 0x80000000u;
/* MASK bit: the pass ends with global stage 0 of an inverse transform (N^-1 is folded in).
 * Compile-time rather than a kernel argument: with a run-time branch both variants of the
 * last group live in one kernel and the 2^14 inverse spills. */
constexpr uint32_t kLastInvFlag = 0x40000000u;
/* MASK bit (inverse, moduli up to 2^52): the first executed group's inputs are CANONICAL words of the caller (or of the pass
 * before): 0 <= x, y < q, so their difference is below q in magnitude like that of two reduced sums (bfly_reduces).  Set by the
 * transform kernels; not by the kernels whose inverse half starts from products (fused products, NTT-domain products). */
constexpr uint32_t kCanonInFlag = 0x20000000u;

struct RedPlan {
  uint8_t red[4][4]; /* [group][local stage]: bit b => butterfly b reduces its sum */
  bool    ok;        /* every sum stays below the limit                            */
  double  bout;      /* worst-case |value|/q leaving the block                     */
};

template <class A, int LOGN, int KSH> constexpr RedPlan make_inv_red_plan()
{
  using P = Plan<LOGN>;
  RedPlan rp{};
  rp.ok         = true;
  double theta2 = 0.25 * 1.001;
  double lim    = 4.0 / 1.001;
  for(int i = 0; i < KSH; i++) {
    theta2 *= 0.5;
    lim *= 2.0;
  }
  lim *= (1.0 - 1.0 / 64.0);
  double b_in = 1.0; /* inputs in [0,q) */
  for(int g = P::NG - 1; g >= 0; g--) {
    double bnd[kE] = {};
    for(int e = 0; e < kE; e++) bnd[e] = b_in;
    for(int jj = 0; jj < P::R(g); jj++) {
      const int    j    = P::R(g) - 1 - jj;
      const double t2   = stage_is_compact<A, LOGN, true>(g, j) ? 1.5 * theta2 : theta2;
      const bool   last = (g == 0 && j == 0);
      const int    ab   = P::ABIT(g, j);
      for(int b = 0; b < kE / 2; b++) {
        const int    e0 = P::BFLY_E0(g, j, b);
        const int    e1 = e0 | (1 << ab);
        const double sm = bnd[e0] + bnd[e1];
        if(sm > lim) rp.ok = false;
        const bool keep = last || (2.0 * sm <= lim);
        if(!keep) rp.red[g][j] = (uint8_t)(rp.red[g][j] | (1u << b));
        bnd[e0] = keep ? sm : 0.501;
        bnd[e1] = f64_rho(sm, t2);
      }
    }
    b_in = 0.0;
    for(int e = 0; e < kE; e++) b_in = bnd[e] > b_in ? bnd[e] : b_in;
  }
  rp.bout = b_in;
  return rp;
}

/* Inverse butterflies for moduli up to 2^52 (WideF64, ntt_arith.h): which of the two reductions a butterfly needs.
 * A Gentleman-Sande stage pairs two slots with the same history (they differ only in the bit the stage processes), so both
 * inputs of a butterfly are of one kind:
 *   REDUCED  |.| <= q/2 + 2    a sum the stage before reduced
 *   other    |.| <= q - N + 3  a product, a sum of two reduced values left unreduced, or anything a previous stage GROUP left
 *                              (across an LDS exchange the bit the last stage processed is a lane bit: the history differs from lane
 *                              to lane and is not a compile-time property of the slot)
 * Two REDUCED inputs: s = x + y and d = x - y are at most q + 4 in magnitude.  The sum is left as it is (kind "other"; the next
 * stage adds two of them: <= 2q + 8 < 2^53 for an NTT-friendly q <= 2^52 - 2N + 1) and the difference is multiplied unreduced
 * (|product| <= q - N + 3, WideF64::inv_bfly2): 8 (9 with a compact twiddle) instead of 14 instructions.  Any other pair: both are
 * reduced first (14).  CANONICAL inputs (0 <= x, y < q: the first stage of a transform kernel, kCanonInFlag): |d| < q, the
 * difference goes unreduced, the sum (up to 2q) is reduced.  Plan per group of four stages: first stage 8 full butterflies, then
 * 4 + 2 + 3 of 8 take the short form -- 12.3 instructions per butterfly on average (12.6 with compact twiddles) where rounds 3 and
 * 4 spent 14. */
struct W52Bfly {
  bool red_d, red_s;
};
template <int LOGN> constexpr bool w52_inputs_reduced(int g, int j, int e0)
{
  using P = Plan<LOGN>;
  if(j + 1 >= P::R(g)) return false;                              /* first executed stage of the group: unknown history */
  const int jp = j + 1;                                            /* the stage executed before */
  if((e0 >> P::ABIT(g, jp)) & 1) return false;                     /* a product of that stage */
  return !w52_inputs_reduced<LOGN>(g, jp, e0);                     /* a sum: reduced unless its butterfly took the short form */
}
template <int LOGN, uint32_t MASK> constexpr W52Bfly w52_inv_plan(int g, int j, int b)
{
  using P         = Plan<LOGN>;
  const int  e0   = P::BFLY_E0(g, j, b);
  const bool red  = w52_inputs_reduced<LOGN>(g, j, e0);
  const bool canon = (MASK & kCanonInFlag) != 0 && g == P::NG - 1 && j + 1 >= P::R(g);
  return W52Bfly{!(red || canon), !red};
}

template <class A, int LOGN, bool INV, uint32_t MASK> constexpr bool bfly_reduces(int g, int j, int b)
{
  if constexpr(INV && A::kWide52) {
    return w52_inv_plan<LOGN, MASK>(g, j, b).red_d; /* (moduli up to 2^52: the flag of inv_bfly<RED> = "reduce the difference") */
  } else if constexpr(INV && (MASK & kRedPlanFlag) != 0) {
    constexpr RedPlan rp = make_inv_red_plan<A, LOGN, (int)(MASK & 0xFFu)>();
    static_assert(rp.ok, "inverse reduction plan exceeds the FP64 exactness limit");
    return (rp.red[g][j] >> b) & 1u;
  } else {
    const int sl = Plan<LOGN>::S(g) + j;
    return (MASK >> (INV ? (LOGN - 1 - sl) : sl)) & 1u;
  }
}

/* butterfly B of local stage J of group G of an inverse block pass, with the reductions the policy's plan gives it */
template <class A, int LOGN, uint32_t MASK, int G, int J, int B, class TW>
NTT_HD void inv_butterfly(typename A::val &x, typename A::val &y, const TW &w, const typename A::consts &c)
{
  if constexpr(A::kWide52) {
    constexpr W52Bfly pl = w52_inv_plan<LOGN, MASK>(G, J, B);
    A::template inv_bfly2<pl.red_d, pl.red_s>(x, y, w, c);
  } else {
    A::template inv_bfly<bfly_reduces<A, LOGN, true, MASK>(G, J, B)>(x, y, w, c);
  }
}

/* every stage of group g has a wave-uniform twiddle slot */
template <int LOGN> constexpr bool group_all_uniform(int g)
{
  using P = Plan<LOGN>;
  for(int j = 0; j < P::R(g); j++)
    if(!P::TW_UNIFORM(g, j)) return false;
  return true;
}
/* inverse, group 0: slot bits toggled by the stages executed before local stage j (j+1 .. R-1) */
template <int LOGN> constexpr uint32_t group0_done_mask(int j)
{
  using P    = Plan<LOGN>;
  uint32_t m = 0;
  for(int k = j + 1; k < P::R(0); k++) m |= 1u << P::ABIT(0, k);
  return m;
}

/* Final group of a whole inverse transform, N^-1 folded into the twiddles.  A value is scaled
 * at the first product it passes inside this group (twiddle N^-1 * w, 16 records kept behind
 * the inverse table, wave-uniform like the group's plain twiddles); a butterfly whose inputs are both scaled
 * already -- its slot has a set bit among the bits this group processed before -- runs
 * unchanged; only slot 0, a sum on every stage of the group, needs the explicit product by
 * N^-1, in the very last butterfly.  (The reference scales all eight sums of its last stage,
 * src/ntt_reference.c:55-65: seven of those products are saved per thread.) */
template <class A, int LOGN, uint32_t MASK>
NTT_HD void run_group0_folded(typename A::val (&x)[kE], uint32_t ib, uint32_t blk, const Params<A> &p)
{
  using P         = Plan<LOGN>;
  constexpr int R = P::R(0);
  static_for<0, R>([&](auto jj) {
    constexpr int      J    = R - 1 - decltype(jj)::value;
    constexpr int      SL   = J; /* S(0) == 0 */
    constexpr int      AB   = P::ABIT(0, J);
    constexpr uint32_t DONE = group0_done_mask<LOGN>(J);
    StageTw<A>         w;
    load_stage_tw<A, LOGN, 0, J, true, false>(w, ib, blk, p, nullptr);
    static_for<0, kE / 2>([&](auto bb) {
      constexpr int      B   = decltype(bb)::value;
      constexpr int      E0  = P::BFLY_E0(0, J, B);
      constexpr int      E1  = E0 | (1 << AB);
      constexpr uint32_t OFF = P::IOFF(0, E0) >> (LOGN - SL);
      if constexpr((E0 & DONE) != 0) {
        inv_butterfly<A, LOGN, MASK, 0, J, B>(x[E0], x[E1], stage_tw<A, LOGN, 0, J, true, B>(w, p.c), p.c);
      } else if constexpr(SL == 0) {
        A::inv_bfly_last(x[E0], x[E1], p.c); /* slot 0: explicit N^-1 on the sum */
      } else {
        /* record N + slot of the inverse table (ntt_host.hip).  blk is 0 here (a whole
         * transform: s0 == 0); keeping it in the index ties the load to the block loop, so
         * these 7 records are re-read from the scalar cache per block instead of being
         * hoisted into 28 more SGPRs for the whole launch (which spilled) */
        const typename A::tw wn = load_tw<A, true, 0>(p.tw, uniform_u32((1u << p.logn) + (blk << SL) + (1u << SL) + OFF));
        inv_butterfly<A, LOGN, MASK, 0, J, B>(x[E0], x[E1], wn, p.c);
      }
    });
  });
}

/* ------------------------------------------------------------------ */
/* radix-4 stage pairs (ArithU64R4)                                      */
/* ------------------------------------------------------------------ */
/*
 * The reference's radix-4 transform (src/ntt_radix4.c:27-114) inside a register-resident stage
 * group: local stages (J, J+1) of the group form one radix-4 level.  With s = radix-2 table slot of
 * stage J for a quad of slots, collect_roots (src/ntt_radix4.c:7-25) reads expanded records 2s
 * (W1) and 4s..4s+3 (W2, W1W2, W3, -W1W3): one 16-byte and one 64-byte fetch, through the scalar cache
 * when the slot is wave-uniform.  A group with an odd number of stages exists only as the last group
 * of an odd-sized block: it is the reference's extra radix-2 stage (:56-61 forward, :85-93 inverse).
 */
template <class A, int LOGN, int G, int J> NTT_HD typename A::pack load_r4_pack(uint32_t ib, uint32_t blk, uint32_t e00off,
                                                                                 const Params<A> &p)
{
  using P           = Plan<LOGN>;
  constexpr int SL  = P::S(G) + J;
  constexpr int SH  = LOGN - SL;
  uint32_t      s   = (1u << (p.s0 + SL)) + (blk << SL) + (ib >> SH) + e00off;
  constexpr bool U  = P::TW_UNIFORM(G, J);
  if constexpr(U) s = uniform_u32(s);
  typename A::pack w;
  struct alignas(16) Four {
    typename A::tw r[4];
  };
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr(U) {
    typedef const typename A::tw __attribute__((address_space(4))) * ctab_t;
    typedef const Four __attribute__((address_space(4))) *           cfour_t;
    w.w1         = ((ctab_t)(uintptr_t)p.tw)[2u * s];
    const Four f = *(cfour_t)(uintptr_t)(p.tw + 4u * s);
    w.w2         = f.r[0];
    w.w12        = f.r[1];
    w.w3         = f.r[2];
    w.nw13       = f.r[3];
    return w;
  }
#endif
  w.w1         = at32(p.tw, 2u * s);
  const Four f = at32(reinterpret_cast<const Four *>(p.tw), s);
  w.w2         = f.r[0];
  w.w12        = f.r[1];
  w.w3         = f.r[2];
  w.nw13       = f.r[3];
  return w;
}

template <class A, int LOGN, int G, bool INV>
NTT_HD void run_group_r4(typename A::val (&x)[kE], uint32_t t, uint32_t blk, const Params<A> &p)
{
  using P           = Plan<LOGN>;
  constexpr int R   = P::R(G);
  const uint32_t ib = P::IBASE(G, t);
  if constexpr(R % 2 == 1) {
    /* single radix-2 stage with the even slots of the expanded table: e[2k] = w[k] */
    static_assert(R == 1 && G == P::NG - 1, "odd stage count only in the last group of an odd-sized block");
    constexpr int  SL = P::S(G);
    constexpr int  SH = LOGN - SL;
    constexpr int  AB = P::ABIT(G, 0);
    const uint32_t tb = (1u << (p.s0 + SL)) + (blk << SL) + (ib >> SH);
    static_for<0, kE / 2>([&](auto bb) {
      constexpr int      B   = decltype(bb)::value;
      constexpr int      E0  = P::BFLY_E0(G, 0, B);
      constexpr uint32_t OFF = P::IOFF(G, E0) >> SH;
      const typename A::tw w = at32(p.tw, 2u * (tb + OFF));
      if constexpr(INV) {
        A::r2_inv_head(x[E0], x[E0 | (1 << AB)], w, p.c);
      } else {
        A::r2_fwd_tail(x[E0], x[E0 | (1 << AB)], w, p.c);
      }
    });
  } else {
    static_for<0, R / 2>([&](auto ll) {
      constexpr int L  = INV ? (R / 2 - 1 - decltype(ll)::value) : decltype(ll)::value;
      constexpr int J  = 2 * L;
      constexpr int BA = P::ABIT(G, J);     /* slot bit of the upper stage: a[i] <-> a[i+2t] */
      constexpr int BB = P::ABIT(G, J + 1); /* slot bit of the lower stage: a[i] <-> a[i+t]  */
      constexpr int SH = LOGN - (P::S(G) + J);
      static_for<0, kE>([&](auto ee) {
        constexpr int E = decltype(ee)::value;
        if constexpr(((E >> BA) & 1) == 0 && ((E >> BB) & 1) == 0) {
          constexpr uint32_t OFF = P::IOFF(G, E) >> SH;
          const typename A::pack w = load_r4_pack<A, LOGN, G, J>(ib, blk, OFF, p);
          if constexpr(INV) {
            A::r4_inv(x[E], x[E | (1 << BB)], x[E | (1 << BA)], x[E | (1 << BA) | (1 << BB)], w, p.c);
          } else {
            A::r4_fwd(x[E], x[E | (1 << BB)], x[E | (1 << BA)], x[E | (1 << BA) | (1 << BB)], w, p.c);
          }
        }
      });
    });
  }
}

template <class A, int LOGN, int G, bool INV, uint32_t MASK, bool LTW = false, bool MIRROR = false>
NTT_HD void run_group(typename A::val (&x)[kE], uint32_t t, uint32_t blk,
                      const Params<A> &p, lds_ctw_ptr<A> ltw = nullptr)
{
  static_assert(!MIRROR || (INV && LTW), "MIRROR: an inverse group reading the forward LDS table");
  using P            = Plan<LOGN>;
  constexpr int R    = P::R(G);
  constexpr int SG   = P::S(G);
  const uint32_t ib  = P::IBASE(G, t);
  if constexpr(A::kRadix4) {
    run_group_r4<A, LOGN, G, INV>(x, t, blk, p);
    return;
  }
  /* Per-lane twiddles that come from global memory are software-pipelined over
   * the stages (the next stage's are requested before this stage's butterflies
   * issue); scalar-cache and LDS-resident twiddles are cheap enough to fetch
   * at the point of use, which keeps them out of the VGPR budget. */
  if constexpr(INV && G == 0 && (MASK & kLastInvFlag) != 0 && group_all_uniform<LOGN>(0)) {
    run_group0_folded<A, LOGN, MASK>(x, ib, blk, p);
    return;
  }
  StageTw<A> wcur, wnxt;
  constexpr int JFIRST = INV ? R - 1 : 0;
  if constexpr(stage_is_compact<A, LOGN, INV>(G, JFIRST) && !LTW) {
    load_stage_tw<A, LOGN, G, JFIRST, INV, false>(wcur, ib, blk, p, nullptr);
  }
  static_for<0, R>([&](auto jj) {
    /* forward walks local stages upward, inverse downward */
    constexpr int J  = INV ? (R - 1 - decltype(jj)::value) : decltype(jj)::value;
    constexpr int SL = SG + J;                 /* local stage              */
    constexpr int AB = P::ABIT(G, J);
    constexpr int  JN   = INV ? J - 1 : J + 1;   /* stage processed next     */
    constexpr bool PIPE = stage_is_compact<A, LOGN, INV>(G, J) && !LTW;
    constexpr bool PIPN = JN >= 0 && JN < R && stage_is_compact<A, LOGN, INV>(G, JN < 0 ? 0 : (JN < R ? JN : 0)) && !LTW;
    if constexpr(!PIPE) load_stage_tw<A, LOGN, G, J, INV, LTW, MIRROR>(wcur, ib, blk, p, ltw);
    if constexpr(PIPN) load_stage_tw<A, LOGN, G, (PIPN ? JN : J), INV, false>(wnxt, ib, blk, p, nullptr);
    constexpr bool FOLDED = INV && SL == 0 && (MASK & kLastInvFlag) != 0;
    if constexpr(FOLDED) {
      /* per-lane twiddles in the final group (blocks below 2^10): all eight sums of the last
       * stage are scaled, as in the reference */
      static_for<0, kE / 2>([&](auto bb) {
        constexpr int B  = decltype(bb)::value;
        constexpr int E0 = P::BFLY_E0(G, J, B);
        A::inv_bfly_last(x[E0], x[E0 | (1 << AB)], p.c);
      });
    }
    if constexpr(!FOLDED) static_for<0, kE / 2>([&](auto bb) {
      constexpr int B  = decltype(bb)::value;
      constexpr int E0 = P::BFLY_E0(G, J, B);
      constexpr int E1 = E0 | (1 << AB);
      constexpr bool RED = bfly_reduces<A, LOGN, INV, MASK>(G, J, B);
      if constexpr(stage_is_compact<A, LOGN, INV>(G, J)) {
        /* compact twiddle used as is (policy overload taking A::ctw) */
        constexpr int F = P::BFLY_FIRST(G, J, B);
        if constexpr(MIRROR) {
          A::template inv_bfly_mirror<RED>(x[E0], x[E1], wcur.c[F], p.c);
        } else if constexpr(INV) {
          inv_butterfly<A, LOGN, MASK, G, J, B>(x[E0], x[E1], wcur.c[F], p.c);
        } else {
          A::template fwd_bfly<RED>(x[E0], x[E1], wcur.c[F], p.c);
        }
      } else if constexpr(INV) {
        inv_butterfly<A, LOGN, MASK, G, J, B>(x[E0], x[E1], stage_tw<A, LOGN, G, J, INV, B>(wcur, p.c), p.c);
      } else {
        A::template fwd_bfly<RED>(x[E0], x[E1], stage_tw<A, LOGN, G, J, INV, B>(wcur, p.c), p.c);
      }
    });
    if constexpr(PIPN) wcur = wnxt;
  });
}

/* Forward, compact-capable policies: fetch ALL twiddles of group G early (8-byte
 * form, R x 8 registers-pairs) so they can be requested ahead of the next
 * block's coefficient prefetch -- vmcnt retires in order, and a twiddle load
 * queued behind 16 HBM loads would otherwise wait for all of them. */
template <class A, int LOGN, int G, uint32_t STAGES = 0xFu>
NTT_HD void preload_group_tw(typename A::ctw (&pre)[4][kE / 2], uint32_t t, uint32_t blk, const Params<A> &p)
{
  using P           = Plan<LOGN>;
  constexpr int R   = P::R(G);
  const uint32_t ib = P::IBASE(G, t);
  static_for<0, R>([&](auto jj) {
    constexpr int J   = decltype(jj)::value;
    if constexpr(((STAGES >> J) & 1u) == 0) return;
    constexpr int SL  = P::S(G) + J;
    constexpr int SH  = LOGN - SL;
    constexpr int AB  = P::ABIT(G, J);
    const uint32_t tb = (1u << (p.s0 + SL)) + (blk << SL) + (ib >> SH);
    static_for<0, kE / 2>([&](auto bb) {
      constexpr int      B   = decltype(bb)::value;
      constexpr int      E0  = ((B >> AB) << (AB + 1)) | (B & ((1 << AB) - 1));
      constexpr uint32_t OFF = P::IOFF(G, E0) >> SH;
      /* two butterflies whose slots are (even, even+1) share one 16-byte load */
      constexpr bool PAIR_LO = (B + 1 < kE / 2) && (OFF % 2 == 0) && P::BFLY_FIRST(G, J, B) == B &&
                               P::BFLY_FIRST(G, J, B + 1 < kE / 2 ? B + 1 : B) == B + 1 &&
                               (P::IOFF(G, P::BFLY_E0(G, J, B + 1 < kE / 2 ? B + 1 : B)) >> SH) == OFF + 1;
      constexpr bool PAIR_HI = (B > 0) && (OFF % 2 == 1) && P::BFLY_FIRST(G, J, B) == B &&
                               P::BFLY_FIRST(G, J, B > 0 ? B - 1 : 0) == B - 1 &&
                               (P::IOFF(G, P::BFLY_E0(G, J, B > 0 ? B - 1 : 0)) >> SH) + 1 == OFF;
      if constexpr(PAIR_LO && sizeof(typename A::ctw) == 8) {
        struct alignas(16) Pair {
          typename A::ctw a, b;
        };
        const Pair v  = at32(reinterpret_cast<const Pair *>(p.tw8), (tb + OFF) >> 1); /* tb+OFF is even: slots of a pair stage */
        pre[J][B]     = v.a;
        pre[J][B + 1] = v.b;
      } else if constexpr(PAIR_HI && sizeof(typename A::ctw) == 8) {
        /* loaded together with its even partner */
      } else if constexpr(P::BFLY_FIRST(G, J, B) == B) {
        pre[J][B] = at32(p.tw8, tb + OFF);
      }
    });
  });
}

template <class A, int LOGN, int G, uint32_t MASK, bool INV = false>
NTT_HD void run_group_preloaded(typename A::val (&x)[kE], const typename A::ctw (&pre)[4][kE / 2],
                                const Params<A> &p)
{
  using P         = Plan<LOGN>;
  constexpr int R = P::R(G);
  static_for<0, R>([&](auto jj) {
    constexpr int  J   = INV ? (R - 1 - decltype(jj)::value) : decltype(jj)::value;
    constexpr int  AB  = P::ABIT(G, J);
    static_for<0, kE / 2>([&](auto bb) {
      constexpr int  B   = decltype(bb)::value;
      constexpr int  E0  = ((B >> AB) << (AB + 1)) | (B & ((1 << AB) - 1));
      constexpr bool RED = bfly_reduces<A, LOGN, INV, MASK>(G, J, B);
      const typename A::ctw w = pre[J][P::BFLY_FIRST(G, J, B)];
      if constexpr(INV) {
        inv_butterfly<A, LOGN, MASK, G, J, B>(x[E0], x[E0 | (1 << AB)], w, p.c);
      } else {
        A::template fwd_bfly<RED>(x[E0], x[E0 | (1 << AB)], w, p.c);
      }
    });
  });
}

/* ------------------------------------------------------------------ */
/* LDS exchange: writer side of group GW into the layout read by GR    */
/* ------------------------------------------------------------------ */
template <class A, int LOGN, int GW, int GR>
NTT_HD void lds_scatter(const typename A::val (&x)[kE], uint32_t t, typename A::val *lds)
{
  using P           = Plan<LOGN>;
  constexpr int ROW = P::ROW;
  const uint32_t i  = P::IBASE(GW, t);
  const uint32_t b  = P::SLOT_OF(GR, i) * ROW + P::THREAD_OF(GR, i);
  static_for<0, kE>([&](auto ee) {
    constexpr int      E  = decltype(ee)::value;
    constexpr uint32_t IO = P::IOFF(GW, E);
    constexpr uint32_t D  = P::SLOT_OF(GR, IO) * ROW + P::THREAD_OF(GR, IO);
    lds[b + D]            = x[E];
  });
}

template <class A, int LOGN, int GW, int GR>
NTT_HD void lds_gather(typename A::val (&x)[kE], uint32_t t, const typename A::val *lds)
{
  using P           = Plan<LOGN>;
  constexpr int ROW = P::ROW;
  /* two base addresses (rows 0-7 / 8-15) so that every ds_read_b64 uses an
   * immediate offset below 64 KiB instead of a VGPR address per row */
  const typename A::val *lo = lds + t;
  const typename A::val *hi = lds + 8 * ROW + t;
  static_for<0, kE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    x[E]            = (E < 8 ? lo : hi)[(E & 7) * ROW];
  });
}

/* ------------------------------------------------------------------ */
/* global memory ends                                                  */
/* ------------------------------------------------------------------ */
struct alignas(16) u64x2 {
  uint64_t a, b;
};

/* Coefficient traffic is read once and written once per launch.  Cache-policy hints were
 * measured on the 2^14 kernel (profiles/r01/ablations.txt): non-temporal STORES lose 4 %,
 * sc1 stores 36 %; nt on the LOADS gains 0.6-1 % and is applied where the persistent loops
 * issue them (ntt_kernels.h, buffer loads).  These generic helpers stay plain. */
NTT_HD uint64_t stream_load(const uint64_t *p) { return *p; }
NTT_HD u64x2    stream_load2(const uint64_t *p) { return *reinterpret_cast<const u64x2 *>(p); }
/* the same marked non-temporal (global_load_dwordx4 ... nt: the line is the first the L2 gives up) */
NTT_HD u64x2 stream_load2_nt(const uint64_t *p)
{
#if defined(__HIP_DEVICE_COMPILE__)
  typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
  const v2u64 v = __builtin_nontemporal_load(reinterpret_cast<const v2u64 *>(p));
  return u64x2{v.x, v.y};
#else
  return stream_load2(p);
#endif
}
NTT_HD void     stream_store(uint64_t *p, uint64_t v) { *p = v; }
NTT_HD void     stream_store2(uint64_t *p, u64x2 v) { *reinterpret_cast<u64x2 *>(p) = v; }

/* coefficient addresses as wave-uniform row pointer + 32-bit lane byte offset (a
 * block is at most 2^14 coefficients), so that loads/stores take the
 * scalar-base + lane-offset form and need no per-lane 64-bit address adds */
NTT_HD const uint64_t *coef_at(const uint64_t *row, uint32_t idx)
{
  return reinterpret_cast<const uint64_t *>(reinterpret_cast<const char *>(row) + (uint32_t)(idx * 8u));
}
NTT_HD uint64_t *coef_at(uint64_t *row, uint32_t idx)
{
  return reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(row) + (uint32_t)(idx * 8u));
}

/* raw u64 -> policy representation for all 16 slots; the lazy-input ("wide")
 * variant is selected by ONE wave-uniform branch around the whole tile */
template <class A, bool INV>
NTT_HD void convert_inputs(typename A::val (&x)[kE], const uint64_t (&raw)[kE], bool wide,
                           const typename A::consts &c)
{
  if(wide) {
    static_for<0, kE>([&](auto ee) {
      constexpr int E = decltype(ee)::value;
      x[E]            = A::template load<INV, true>(raw[E], c);
    });
  } else {
    static_for<0, kE>([&](auto ee) {
      constexpr int E = decltype(ee)::value;
      x[E]            = A::template load<INV, false>(raw[E], c);
    });
  }
}

/* Inner product in the NTT domain (dot_inv_kernel): the thread's 16 slots of one operand pair a_i^, b_i^ as raw words
 * (last-kind layout: what the inverse transform's first group consumes) become the first term of the running sums
 * (FIRST) or are added to them.  lazy: the words may be anywhere in [0,4q) -- ONE wave-uniform branch around the tile. */
NTT_HD void sched_fence()
{
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0); /* the scheduler moves nothing across this point */
#endif
}
/* slots [E0, E1) of the tile */
template <class A, int E0, int E1>
NTT_HD void dot_slots(typename A::val (&x)[kE], const uint64_t (&ra)[kE], const uint64_t (&rb)[kE], bool lazy,
                      const typename A::consts &c)
{
  if(lazy) {
    static_for<E0, E1>([&](auto ee) {
      constexpr int E = decltype(ee)::value;
      x[E]            = A::dot_acc(x[E], A::template dot_term<true>(ra[E], rb[E], c), c);
    });
  } else {
    static_for<E0, E1>([&](auto ee) {
      constexpr int E = decltype(ee)::value;
      x[E]            = A::dot_acc(x[E], A::template dot_term<false>(ra[E], rb[E], c), c);
    });
  }
}
/* slots [E0, E1) of the tile, CH at a time: on the device the scheduler is fenced between the chunks, so that at most CH
 * products are in flight (each holds about ten temporaries next to the 64 registers of words and the 32 running sums) */
template <class A, int E0 = 0, int E1 = kE, int CH = A::kDotChunk>
NTT_HD void dot_tile(typename A::val (&x)[kE], const uint64_t (&ra)[kE], const uint64_t (&rb)[kE], bool lazy,
                     const typename A::consts &c)
{
  static_for<0, (E1 - E0) / CH>([&](auto cc) {
    constexpr int C = decltype(cc)::value;
    sched_fence();
    dot_slots<A, E0 + C * CH, E0 + C * CH + CH>(x, ra, rb, lazy, c);
    sched_fence();
  });
}
/* Product at the output of a forward transform (fwd_mul_kernel), slots [E0, E1) of the last group's layout:
 * u = canonical(x * b + rc); rc holds the accumulator words, or zeros when the call does not accumulate.  lazy: a
 * wave-uniform branch around the chunk. */
template <class A, int E0, int E1>
NTT_HD void mul_out_slots(uint64_t (&u)[kE], const typename A::val (&x)[kE], const uint64_t (&rb)[kE], const uint64_t (&rc)[kE],
                          bool lazy, const typename A::consts &c)
{
  if(lazy) {
    static_for<E0, E1>([&](auto ee) {
      constexpr int E = decltype(ee)::value;
      u[E]            = A::mul_store_acc(A::template mul_out<true>(x[E], rb[E], c), rc[E], c);
    });
  } else {
    static_for<E0, E1>([&](auto ee) {
      constexpr int E = decltype(ee)::value;
      u[E]            = A::mul_store_acc(A::template mul_out<false>(x[E], rb[E], c), rc[E], c);
    });
  }
}
/* CH slots at a time with the scheduler fenced in between (register budget, as dot_tile) */
template <class A, int E0, int E1, int CH = 2>
NTT_HD void mul_out_tile(uint64_t (&u)[kE], const typename A::val (&x)[kE], const uint64_t (&rb)[kE], const uint64_t (&rc)[kE],
                         bool lazy, const typename A::consts &c)
{
  static_for<0, (E1 - E0) / CH>([&](auto cc) {
    constexpr int C = decltype(cc)::value;
    sched_fence();
    mul_out_slots<A, E0 + C * CH, E0 + C * CH + CH>(u, x, rb, rc, lazy, c);
    sched_fence();
  });
}
/* store slots [E0, E1) of the last-kind layout as 16-byte words */
template <int LOGN, int E0 = 0, int E1 = kE> NTT_HD void store_last_raw(const uint64_t (&u)[kE], uint32_t t, uint64_t *blk)
{
  using P           = Plan<LOGN>;
  constexpr int G   = P::NG - 1;
  const uint32_t ib = P::IBASE(G, t);
  static_for<E0 / 2, E1 / 2>([&](auto hh) {
    constexpr int E = 2 * decltype(hh)::value;
    stream_store2(coef_at(blk + P::IOFF(G, E), ib), u64x2{u[E], u[E + 1]});
  });
}

template <class A> NTT_HD void dot_fold_tile(typename A::val (&x)[kE], const typename A::consts &c)
{
  static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::dot_fold(x[decltype(ee)::value], c); });
}
/* raw words of a block in the last-kind layout (runs of 2^RL consecutive indices per lane, 16-byte loads), slots [E0, E1) */
template <int LOGN, int E0 = 0, int E1 = kE, bool NT = false> NTT_HD void load_last_raw(uint64_t (&raw)[kE], uint32_t t, const uint64_t *blk)
{
  using P           = Plan<LOGN>;
  constexpr int G   = P::NG - 1;
  const uint32_t ib = P::IBASE(G, t);
  static_assert(E0 % 2 == 0 && E1 % 2 == 0, "slots come in pairs");
  static_for<E0 / 2, E1 / 2>([&](auto hh) {
    constexpr int E = 2 * decltype(hh)::value;
    const u64x2   v = NT ? stream_load2_nt(coef_at(blk + P::IOFF(G, E), ib)) : stream_load2(coef_at(blk + P::IOFF(G, E), ib));
    raw[E]          = v.a;
    raw[E + 1]      = v.b;
  });
}

/* first-kind group: slot e <-> index (e << LT) + t : 8-byte coalesced */
template <class A, int LOGN, bool INV>
NTT_HD void global_load_first(typename A::val (&x)[kE], uint32_t t, const uint64_t *blk,
                              bool wide, const typename A::consts &c)
{
  using P = Plan<LOGN>;
  uint64_t raw[kE];
  static_for<0, kE>([&](auto ee) {
    constexpr int   E   = decltype(ee)::value;
    const uint64_t *row = blk + ((uint32_t)E << P::LT); /* wave-uniform base, one lane offset */
    raw[E]              = stream_load(coef_at(row, t));
  });
  convert_inputs<A, INV>(x, raw, wide, c);
}

/* last-kind group: slots come in runs of 2^RL consecutive indices */
template <class A, int LOGN, bool INV>
NTT_HD void global_load_last(typename A::val (&x)[kE], uint32_t t, const uint64_t *blk,
                             bool wide, const typename A::consts &c)
{
  using P              = Plan<LOGN>;
  constexpr int G      = P::NG - 1;
  const uint32_t ib    = P::IBASE(G, t);
  uint64_t raw[kE];
  static_for<0, kE / 2>([&](auto hh) {
    constexpr int E = 2 * decltype(hh)::value;
    const u64x2   v = stream_load2(coef_at(blk + P::IOFF(G, E), ib));
    raw[E]          = v.a;
    raw[E + 1]      = v.b;
  });
  convert_inputs<A, INV>(x, raw, wide, c);
}

/* policy value -> output word.  Lazy outputs (the reference's *_lazy contract, include/ntt_reference.h:13-17)
 * skip the final reduction: for the FP64 policy that is a property of the kernel (LAZYT: its reduction
 * schedule bounds the last stage), for the integer policies a run-time flag of the launch. */
template <class A, bool INV, bool LAZYT> NTT_HD uint64_t out_word(typename A::val v, bool lazy_rt, const typename A::consts &c)
{
  if constexpr(A::kTracksBounds) {
    (void)lazy_rt;
    if constexpr(INV) return A::store_inv(v, c);
    else return LAZYT ? A::store_fwd_lazy(v, c) : A::store_fwd(v, c);
  } else {
    const uint64_t keep = lazy_rt ? 0ull : ~0ull; /* launch-uniform: scalar */
    return INV ? A::store_inv_sel(v, c, keep) : A::store_fwd_sel(v, c, keep);
  }
}

template <class A, int LOGN, bool INV, bool LAZYT = false>
NTT_HD void global_store_first(const typename A::val (&x)[kE], uint32_t t, uint64_t *blk,
                               const typename A::consts &c, bool lazy_rt = false)
{
  using P = Plan<LOGN>;
  static_for<0, kE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    stream_store(coef_at(blk + ((uint32_t)E << P::LT), t), out_word<A, INV, LAZYT>(x[E], lazy_rt, c));
  });
}

template <class A, int LOGN, bool INV, bool LAZYT = false>
NTT_HD void global_store_last(const typename A::val (&x)[kE], uint32_t t, uint64_t *blk,
                              const typename A::consts &c, bool lazy_rt = false)
{
  using P           = Plan<LOGN>;
  constexpr int G   = P::NG - 1;
  const uint32_t ib = P::IBASE(G, t);
  static_for<0, kE / 2>([&](auto hh) {
    constexpr int E = 2 * decltype(hh)::value;
    u64x2         v;
    v.a = out_word<A, INV, LAZYT>(x[E], lazy_rt, c);
    v.b = out_word<A, INV, LAZYT>(x[E + 1], lazy_rt, c);
    stream_store2(coef_at(blk + P::IOFF(G, E), ib), v);
  });
}

/* ------------------------------------------------------------------ */
/* reduction schedules (ArithF64) for a fused block pass                */
/* ------------------------------------------------------------------ */
/* |value|/q a lazy forward output may have: v + 2q must stay in [0,4q) */
constexpr double kLazyBound = 1.99;

/* forward stages whose twiddles are compact (bit = local stage) */
template <class A, int LOGN> constexpr uint32_t fused_cmask()
{
  using P        = Plan<LOGN>;
  uint32_t cmask = 0;
  for(int g = 0; g < P::NG; g++) {
    for(int j = 0; j < P::R(g); j++) {
      if(stage_is_compact<A, LOGN, false>(g, j)) cmask |= 1u << (P::S(g) + j);
    }
  }
  return cmask;
}

template <class A, int LOGN, bool INV, int KSH, bool LAZY = false> constexpr uint32_t fused_mask()
{
  if constexpr(A::kIntWide) {
    return u64x_schedule(INV, LOGN, A::kHead); /* ArithU64X: the stages that fold the growing operand */
  } else if constexpr(!A::kTracksBounds) {
    return 0;
  } else {
    /* the schedule is causal, so when the last inverse stage is the folded
     * N^-1 butterfly its (unused) bit does not disturb the earlier ones */
    /* stages whose twiddles are compact estimate the quotient from the rounded
     * product (ArithF64::mulmod_c): 1.5x the error term at those positions */
    if constexpr(A::kWide52) {
      /* moduli up to 2^52: forward = which stages also reduce the multiplied operand (ntt_arith.h f64w_fwd_schedule); the
       * inverse butterflies of that policy reduce everything and ignore the mask */
      if constexpr(INV) return 0;
      else return f64w_fwd_schedule(LOGN, 1.0, 0u).mask; /* (compact stages use mulmod_c2: a full record's bounds) */
    }
    if constexpr(INV) return kRedPlanFlag | (uint32_t)KSH; /* per-butterfly plan (bfly_reduces) */
    constexpr F64Sched sc = f64_schedule(false, LOGN, KSH, 1.0, fused_cmask<A, LOGN>(), LAZY ? kLazyBound : 1e30);
    /* store_fwd_lazy builds the bit pattern of v + 2q with one fma: only correct while |v| <= kLazyBound * q.  The
     * schedule forces reductions from the last stage backwards until the bound holds; if a future plan or twiddle
     * layout made that impossible the loop would simply end -- fail the build instead of emitting corrupt words */
    static_assert(!LAZY || sc.bout <= kLazyBound, "lazy forward output: the reduction schedule cannot bound the last stage");
    return sc.mask;
  }
}

/* ------------------------------------------------------------------ */
/* strided ("column") pass: R stages on elements far apart in memory    */
/* ------------------------------------------------------------------ */
/*
 * Used for the leading stages of transforms larger than one fused block and for
 * sizes below the fused range.  Global stages [S, S+R) of an N=2^m transform:
 * i = hi*(2^R*span) + e*span + lo, span = N >> (S+R); one thread per (hi,lo).
 * Twiddle slot at local stage j: 2^(S+j) + (hi<<j) + (e>>(R-j)).
 */
template <class A, int R, bool INV, uint32_t MASK>
NTT_HD void column_pass_thread(uint64_t *poly, uint32_t col, uint32_t logn, uint32_t S,
                               bool wide, bool lastinv, const typename A::tw *tab,
                               const typename A::consts &c, bool lazy_out = false)
{
  constexpr int  NE   = 1 << R;
  const uint32_t lsp  = logn - S - R; /* log2 span */
  const uint32_t lo   = col & ((1u << lsp) - 1);
  const uint32_t hi   = col >> lsp;
  uint64_t *     base = poly + ((uint64_t)hi << (lsp + R)) + lo;
  typename A::val x[NE];
  static_for<0, NE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    const uint64_t r = base[(uint64_t)E << lsp];
    x[E]             = wide ? A::template load<INV, true>(r, c) : A::template load<INV, false>(r, c);
  });
  static_for<0, R>([&](auto jj) {
    constexpr int  J   = INV ? (R - 1 - decltype(jj)::value) : decltype(jj)::value;
    constexpr int  AB  = R - 1 - J;
    constexpr int  POS = INV ? (R - 1 - J) : J;
    constexpr bool RED = (MASK >> POS) & 1u;
    const uint32_t tb  = (1u << (S + J)) + (hi << J);
    static_for<0, NE>([&](auto ee) {
      constexpr int E0 = decltype(ee)::value;
      if constexpr(((E0 >> AB) & 1) == 0) {
        constexpr int E1 = E0 | (1 << AB);
        if(INV && J == 0 && lastinv) {
          A::inv_bfly_last(x[E0], x[E1], c); /* global stage 0: S == 0 */
        } else {
          const typename A::tw w = tab[tb + (E0 >> (R - J))];
          if constexpr(INV) {
            A::template inv_bfly<RED>(x[E0], x[E1], w, c);
          } else {
            A::template fwd_bfly<RED>(x[E0], x[E1], w, c);
          }
        }
      }
    });
  });
  static_for<0, NE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    base[(uint64_t)E << lsp] = out_word<A, INV, false>(x[E], lazy_out, c);
  });
}

/* The same pass in the reference's radix-4 formulation: local stages (J, J+1) form one radix-4 level whose five-twiddle
 * pack is collect_roots' (src/ntt_radix4.c:7-25) for radix-2 slot s of stage S + J: expanded records 2s and 4s .. 4s+3.
 * Forward (the FIRST pass of a transform larger than a block): values stay in the butterfly's lazy range [0,8q) from load
 * to store, as they do in the reference's array between its levels (src/ntt_radix4.c:33-48).  Inverse (the LAST pass,
 * S = 0): words in [0,2q) from the block pass, the levels in the opposite order (:95-109), and the reference's final
 * N^-1 pass (:111-113) fused into the store. */
template <class A, int R, bool INV>
NTT_HD void column_pass_thread_r4(uint64_t *poly, uint32_t col, uint32_t logn, uint32_t S, const typename A::tw *tab,
                                  const typename A::consts &c, bool lazy_out)
{
  static_assert(A::kRadix4 && R % 2 == 0, "radix-4 levels come in stage pairs");
  constexpr int  NE   = 1 << R;
  const uint32_t lsp  = logn - S - R; /* log2 span */
  const uint32_t lo   = col & ((1u << lsp) - 1);
  const uint32_t hi   = col >> lsp;
  uint64_t *     base = poly + ((uint64_t)hi << (lsp + R)) + lo;
  typename A::val x[NE];
  static_for<0, NE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    x[E]            = A::template load<INV, false>(base[(uint64_t)E << lsp], c);
  });
  static_for<0, R / 2>([&](auto ll) {
    constexpr int  L  = INV ? (R / 2 - 1 - decltype(ll)::value) : decltype(ll)::value;
    constexpr int  J  = 2 * L;
    constexpr int  BA = R - 1 - J; /* slot bit of the upper stage: a[i] <-> a[i+2t] */
    constexpr int  BB = R - 2 - J; /* slot bit of the lower stage: a[i] <-> a[i+t]  */
    const uint32_t tb = (1u << (S + J)) + (hi << J);
    static_for<0, NE>([&](auto ee) {
      constexpr int E = decltype(ee)::value;
      if constexpr(((E >> BA) & 1) == 0 && ((E >> BB) & 1) == 0) {
        const uint32_t   s = tb + (uint32_t)(E >> (R - J));
        typename A::pack w;
        w.w1   = tab[2u * s];
        w.w2   = tab[4u * s];
        w.w12  = tab[4u * s + 1u];
        w.w3   = tab[4u * s + 2u];
        w.nw13 = tab[4u * s + 3u];
        if constexpr(INV) {
          A::r4_inv(x[E], x[E | (1 << BB)], x[E | (1 << BA)], x[E | (1 << BA) | (1 << BB)], w, c);
        } else {
          A::r4_fwd(x[E], x[E | (1 << BB)], x[E | (1 << BA)], x[E | (1 << BA) | (1 << BB)], w, c);
        }
      }
    });
  });
  static_for<0, NE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    if constexpr(INV) {
      base[(uint64_t)E << lsp] = lazy_out ? A::store_inv_lazy(x[E], c) : A::store_inv(x[E], c);
    } else {
      base[(uint64_t)E << lsp] = A::store_fwd_lazy(x[E], c);
    }
  });
}

/* ------------------------------------------------------------------ */
/* in-launch work queues of the XCD-local kernels (team_kernel, team_product_kernel) */
/* ------------------------------------------------------------------ */
/*
 * A launch has eight in-order queues (one per XCD).  The polynomials of the launch -- `total` of them: with several RNS
 * limbs in one launch, limb l's polynomial p is number l * batch + p -- are dealt to the queues statically: queue q owns
 * q, q + 8, q + 16, ...  Entry k of a queue (k = 0, 1, 2, ... as handed out by an atomic counter) decodes to one ITEM:
 *   step = k / PER, r = k % PER, PER = n0 + n1 + n2 items of the (up to three) passes of one polynomial;
 *   pass 0 items of the queue's polynomial number j = step, pass 1 items of j = step - lag, pass 2 items of
 *   j = step - 2 lag: a later pass runs `lag` polynomials behind the one before it.
 * An item whose j falls outside [0, J) is a hole (the queue's head and tail): nothing to do.  Items of pass P > 0 wait until
 * all items of pass P - 1 of their polynomial have signalled completion; those were handed out EARLIER in the same queue
 * (step - lag < step, and inside a step the passes are in order), so whoever waits, waits for an item that is already in
 * the hands of a running workgroup: no deadlock whatever the number of resident workgroups.  Pass-0 items never wait.
 * tests/test_team_protocol.py simulates the protocol on this very function.
 */
struct TeamItem {
  uint32_t stop;  /* the queue is exhausted: leave it */
  uint32_t valid; /* 0: a hole, fetch the next entry */
  uint32_t pass, item;
  uint32_t v;     /* the polynomial, 0 .. total-1 (limb * batch + polynomial) */
};

NTT_HD uint32_t team_queue_polys(uint32_t total, uint32_t q) { return total > q ? (total - q + 7u) / 8u : 0u; }

NTT_HD TeamItem team_decode(uint32_t k, uint32_t q, uint32_t total, uint32_t lag, uint32_t n0, uint32_t n1, uint32_t n2)
{
  const uint32_t J      = team_queue_polys(total, q);
  const uint32_t npass  = n2 ? 3u : 2u;
  const uint32_t steps  = J + (npass - 1u) * lag;
  const uint32_t per    = n0 + n1 + n2;
  const uint32_t step   = k / per, r = k % per;
  TeamItem       it;
  it.stop  = step >= steps ? 1u : 0u;
  it.pass  = r < n0 ? 0u : (r < n0 + n1 ? 1u : 2u);
  it.item  = it.pass == 0 ? r : (it.pass == 1 ? r - n0 : r - n0 - n1);
  const int64_t j = (int64_t)step - (int64_t)it.pass * (int64_t)lag;
  it.valid = (!it.stop && j >= 0 && j < (int64_t)J) ? 1u : 0u;
  it.v     = q + 8u * (uint32_t)(j < 0 ? 0 : j);
  return it;
}

template <class A, int R, bool INV, int KSH> constexpr uint32_t column_mask()
{
  if constexpr(A::kIntWide) {
    return u64x_schedule(INV, R, A::kHead);
  } else if constexpr(!A::kTracksBounds) {
    return 0;
  } else if constexpr(A::kWide52) {
    /* forward: the stages that also reduce the multiplied operand; inverse: every stage reduces its difference (the passes are
     * memory-bound: the per-slot plan of the block kernels, bfly_reduces, is not worth a second form here) */
    return INV ? ((1u << R) - 1u) : f64w_fwd_schedule(R, 1.0, 0u).mask; /* (column passes read full records) */
  } else {
    return f64_schedule(INV, R, KSH, 1.0).mask;
  }
}

/* ------------------------------------------------------------------ */
/* N = 2^15 in one pass (onepass_kernel): the stage on pairs 2^14 apart  */
/* ------------------------------------------------------------------ */
/* The reduction schedule of the FP64 policies over all FIFTEEN forward stages: bit 0 = the pair stage (global stage 0, a full twiddle
 * record), bits 1..14 = the block stages (local stage s of either half = global stage s + 1, compact twiddles where the 2^14 block
 * has them).  One schedule instead of "column pass + block pass with canonical words between": nothing is reduced just because a
 * pass ends. */
template <class A, int KSH> constexpr uint32_t onepass_fwd_mask()
{
  if constexpr(A::kWide52) {
    return f64w_fwd_schedule(kFusedLarge + 1, 1.0, 0u).mask;
  } else {
    return f64_schedule(false, kFusedLarge + 1, KSH, 1.0, fused_cmask<A, kFusedLarge>() << 1).mask;
  }
}
/* forward: x[i] +- w x[i + 2^14] for the thread's sixteen pairs (slot e of half 0 with slot e of half 1), the single twiddle of
 * slot 1 -- the reference's first stage, src/ntt_reference.c:17-30 with m = 1 */
template <class A, bool RED0> NTT_HD void onepass_pairs_fwd(typename A::val (&xa)[kE], typename A::val (&xb)[kE], const Params<A> &p)
{
  const typename A::tw w = load_tw<A, true>(p.tw, 1u);
  static_for<0, kE>([&](auto ee) { A::template fwd_bfly<RED0>(xa[decltype(ee)::value], xb[decltype(ee)::value], w, p.c); });
}
/* inverse: the last Gentleman-Sande stage with N^-1 folded in (src/ntt_reference.c:55-65) on one pair.  Its inputs are what the
 * halves' block stages left in registers -- unreduced, bounded by the per-slot plan's `bout`, which the last butterfly's exactness
 * argument (sum and difference of CANONICAL words) does not cover: both are reduced first, three exact instructions each. */
template <class A> NTT_HD void onepass_pair_inv(typename A::val &va, typename A::val &vb, const typename A::consts &c)
{
  va = A::reduce(va, c);
  vb = A::reduce(vb, c);
  A::inv_bfly_last(va, vb, c);
}

/* ------------------------------------------------------------------ */
/* fwd_ntt_radix4x4_lazy when log2 N = 4k+3, one layer at a time          */
/* ------------------------------------------------------------------ */
/*
 * The reference's radix-16 formulation (src/ntt_radix4x4.c:41-114) runs the butterflies of fwd_ntt_radix4_lazy in
 * another order -- no value changes -- EXCEPT for its remainder when log2 N = 4k+3 (:91-111): 2k radix-4 layers,
 * then a radix-2 stage on distance-4 pairs, then the last radix-4 layer, where ntt_radix4.c runs 2k+1 radix-4 layers
 * and ends on the radix-2 stage.  Same residues, other lazy words.  The reference-signature entry point serves those
 * sizes (2^7, 2^11, 2^15: one polynomial per call, a drop-in path whose job is identical words, not throughput) with one
 * launch per layer over global memory; `id` is the butterfly's index within the layer.  Host/device so that the CPU
 * emulator (tests/emu) runs the very same functions.
 */
/* layer of `blocks` groups of 4*span coefficients, pack blocks + b (collect_roots, src/ntt_radix4x4.c:7-25; e = records
 * {e[k], e_con[k]} of the caller's expanded table) */
NTT_HD void r4x4_layer_r4(uint64_t *a, const TwU64 *e, uint64_t blocks, uint64_t span, uint64_t id, const ArithU64::consts &c)
{
  const uint64_t   b = id / span, i = id - b * span, k = blocks + b;
  ArithU64R4::pack w;
  w.w1        = e[2 * k];
  w.w2        = e[4 * k];
  w.w12       = e[4 * k + 1];
  w.w3        = e[4 * k + 2];
  w.nw13      = e[4 * k + 3];
  uint64_t *x = a + 4 * span * b + i;
  ArithU64R4::r4_fwd(x[0], x[span], x[2 * span], x[3 * span], w, c);
}
/* the radix-2 stage in front of the last radix-4 layer (:93-104): N/8 groups of 8, pairs (j, j+4), twiddle e[2(N/8+g)].
 * The reference's loop applies reduce_8q_to_4q to a[i] for its GROUP counter i: coefficient 0 before its butterfly,
 * coefficients 1 .. N/8-1 after theirs (each belongs to an earlier group i/8 < i) -- reproduced, or the words entering
 * the last layer's Shoup products would differ */
NTT_HD void r4x4_layer_r2(uint64_t *a, const TwU64 *e, uint64_t N, uint64_t id, const ArithU64::consts &c)
{
  const uint64_t groups = N >> 3, g = id >> 2, j = 8 * g + (id & 3);
  const uint64_t q4     = 2 * c.q2;
  uint64_t       x = a[j], y = a[j + 4];
  if(j == 0) x = ArithU64::csub(x, q4);
  ArithU64::fwd_bfly<false>(x, y, e[2 * (groups + g)], c); /* harvey_fwd_butterfly, fast_mul_operators.h:72-81 */
  if(j >= 1 && j < groups) x = ArithU64::csub(x, q4);
  if(j + 4 < groups) y = ArithU64::csub(y, q4);
  a[j]     = x;
  a[j + 4] = y;
}

} /* namespace ntt */
