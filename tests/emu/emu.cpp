/*
 * emu.cpp -- CPU emulation of the gfx950 kernel logic (TEST INFRASTRUCTURE).
 *
 * Compiles the very same templates the HIP kernels are built from
 * (csrc/ntt_core.h, ntt_arith.h, ntt_passplan.h, ntt_tables.h) with g++ and runs
 * them thread-by-thread, phase-by-phase, so that index maps, LDS exchange
 * layouts, twiddle addressing, the FP64 exactness argument and its reduction
 * schedule can be checked against the oracle without a GPU.  Not part of the
 * product library; built by tests/emu/Makefile, driven by tests/test_emu.py.
 */
#include <array>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ntt_core.h"
#include "ntt_passplan.h"
#include "ntt_tables.h"

using namespace ntt;

/* The file is compiled several times (tests/emu/Makefile, -DEMU_PART=k, in parallel): part 0 holds the C interface,
 * parts 1..6 each a share of the emu_run / emu_fused_product instantiations (explicit instantiation definitions);
 * without EMU_PART everything lands in one translation unit (the sanitizer build). */
#ifndef EMU_PART
#  define EMU_PART (-1)
#endif
#define EMU_HAS(k) (EMU_PART == -1 || EMU_PART == (k))

/* ------------------------------------------------------------------ */
/* ArithF64Chk: the FP64 policy with every exactness claim of DESIGN.md */
/* section 4 asserted at run time against 128-bit integer arithmetic.   */
/* ------------------------------------------------------------------ */
/* words between consecutive polynomials for the next calls (emu_set_poly_stride; 0 = dense: N) -- the caller-native layouts of
 * the library's *_strided entry points, through the same block_offset the kernels use */
inline uint64_t g_pstride = 0;
static inline uint64_t emu_pstride(int m) { return g_pstride ? g_pstride : (1ull << m); }
/* pointer batches (emu_set_poly_table): host table of the polynomials' ADDRESSES for the next emu_transform -- the kernels' own
 * poly_offset / block_offset (csrc/ntt_core.h) then take every polynomial's start from it, with a null data pointer */
inline const uint64_t *g_ptab = nullptr;
inline int g_one_pass = -1; /* emu_set_one_pass: 1 = 2^15 transforms of the FP64 policies take the one-pass route, 0 = the two-pass route, -1 = as the library (on) */
inline uint64_t g_opstride = 0; /* emu_inv_dot: words between consecutive operands of the a / b arrays (0 = batch * N: dense) */

inline uint64_t g_chk_fail  = 0;   /* number of violated claims          (inline: one copy for all parts) */
inline double   g_chk_maxb  = 0;   /* largest |value|/q seen             */
inline double   g_chk_maxr  = 0;   /* largest |product|/q seen           */

struct ArithF64Chk : ArithF64 {
  static __int128 as_int(double v)
  {
    if(v != __builtin_rint(v) || __builtin_fabs(v) >= 9007199254740992.0) g_chk_fail++; /* integer, < 2^53 */
    return (__int128)v;
  }
  static void see(double v, const consts &c)
  {
    (void)as_int(v);
    const double b = __builtin_fabs(v) / c.q;
    if(b > g_chk_maxb) g_chk_maxb = b;
  }
  static double mulmod(const tw &t, double y, const consts &c)
  {
    const double   r  = ArithF64::mulmod(t, y, c);
    const __int128 q  = (__int128)c.qi;
    const __int128 ex = as_int(y) * as_int(t.w);
    __int128       d  = (ex - as_int(r)) % q;
    if(d != 0) g_chk_fail++;                              /* r == y*w (mod q), exactly */
    const double rb = __builtin_fabs(r) / c.q;
    if(rb > g_chk_maxr) g_chk_maxr = rb;
    return r;
  }
  static double mulmod_c(ctw w, double y, const consts &c)
  {
    const double   r  = ArithF64::mulmod_c(w, y, c);
    const __int128 q  = (__int128)c.qi;
    const __int128 ex = as_int(y) * as_int(w);
    if((ex - as_int(r)) % q != 0) g_chk_fail++;            /* r == y*w (mod q), exactly */
    const double rb = __builtin_fabs(r) / c.q;
    if(rb > g_chk_maxr) g_chk_maxr = rb;
    return r;
  }
  static double mulmod_c2(ctw w, double y, const consts &c)
  {
    const double   r  = ArithF64::mulmod_c2(w, y, c);
    const __int128 ex = as_int(y) * as_int(w);
    if((ex - as_int(r)) % (__int128)c.qi != 0) g_chk_fail++;       /* r == y*w (mod q), exactly */
    /* the estimate is as good as a stored quotient: |r| <= (1/2 + |y|/q * theta2 (1 + slack)) q */
    const double by = __builtin_fabs(y) / c.q, th2 = c.q / 9007199254740992.0;
    if(__builtin_fabs(r) > (0.5 + by * th2 * 1.01 + 0.001) * c.q) g_chk_fail++;
    const double rb = __builtin_fabs(r) / c.q;
    if(rb > g_chk_maxr) g_chk_maxr = rb;
    return r;
  }
  static double reduce(double v, const consts &c)
  {
    const double   r = ArithF64::reduce(v, c);
    if((as_int(v) - as_int(r)) % (__int128)c.qi != 0) g_chk_fail++;
    if(__builtin_fabs(r) > 0.5 * c.q + 2.0) g_chk_fail++;
    return r;
  }
  template <bool RED> static void fwd_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    see(x, c);
    see(y, c);
    const double xr = RED ? reduce(x, c) : x;
    const double m  = mulmod(t, y, c);
    x               = xr + m;
    y               = xr - m;
    see(x, c);
    see(y, c);
  }
  template <bool RED> static void inv_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    see(x, c);
    see(y, c);
    const double s = x + y;
    const double d = x - y;
    see(s, c);
    see(d, c);
    x = RED ? reduce(s, c) : s;
    y = mulmod(t, d, c);
  }
  template <bool RED> static void fwd_bfly(val &x, val &y, ctw w, const consts &c)
  {
    see(x, c);
    see(y, c);
    const double xr = RED ? reduce(x, c) : x;
    const double m  = mulmod_c(w, y, c);
    x               = xr + m;
    y               = xr - m;
    see(x, c);
    see(y, c);
  }
  template <bool RED> static void inv_bfly(val &x, val &y, ctw w, const consts &c)
  {
    see(x, c);
    see(y, c);
    const double s = x + y;
    const double d = x - y;
    see(s, c);
    see(d, c);
    x = RED ? reduce(s, c) : s;
    y = mulmod_c(w, d, c);
  }
  static void inv_bfly_last(val &x, val &y, const consts &c)
  {
    const double s = x + y;
    const double d = x - y;
    see(s, c);
    see(d, c);
    x = mulmod(c.ninv, s, c);
    y = mulmod(c.wninv, d, c);
  }
  template <bool RED> static void inv_bfly_mirror(val &x, val &y, ctw wneg, const consts &c)
  {
    see(x, c);
    see(y, c);
    const double s = x + y;
    const double d = y - x;
    see(s, c);
    see(d, c);
    x = RED ? reduce(s, c) : s;
    y = mulmod_c(wneg, d, c);
  }
  static val product_rr(val x, val y, const consts &c)
  {
    see(x, c);
    see(y, c);
    const double   r  = ArithF64::mulmod_c(reduce(y, c), reduce(x, c), c);
    const __int128 ex = as_int(x) * as_int(y);
    if((ex - as_int(r)) % (__int128)c.qi != 0) g_chk_fail++;      /* r == x * y (mod q), exactly */
    if(__builtin_fabs(r) > 0.75 * c.q) g_chk_fail++;              /* the bound the inverse plan relies on */
    return r;
  }
  template <bool LAZY> static val product_in_domain(val x, uint64_t a, const consts &c)
  {
    see(x, c);
    const double   r  = ArithF64::product_in_domain<LAZY>(x, a, c);
    const __int128 ex = as_int(x) * (__int128)(a % c.qi);
    if((ex - as_int(r)) % (__int128)c.qi != 0) g_chk_fail++;      /* r == x * a (mod q), exactly */
    if(__builtin_fabs(r) > 0.75 * c.q) g_chk_fail++;              /* the bound the inverse plan relies on */
    return r;
  }
  /* inner-product terms (dot_inv_kernel): exact product of the two stored words, bound of ArithF64::dot_term */
  template <bool LAZY> static val dot_term(uint64_t a, uint64_t b, const consts &c)
  {
    if(LAZY ? (a >= 4 * c.qi || b >= 4 * c.qi) : (a >= c.qi || b >= c.qi)) g_chk_fail++; /* operand range of the contract */
    const double   r  = ArithF64::dot_term<LAZY>(a, b, c);
    const __int128 ex = (__int128)(a % c.qi) * (__int128)(b % c.qi);
    if((ex - as_int(r)) % (__int128)c.qi != 0) g_chk_fail++;       /* r == a * b (mod q), exactly */
    if(__builtin_fabs(r) > 0.8756 * c.q) g_chk_fail++;            /* the bound the folding schedule relies on */
    return r;
  }
  static val dot_acc(val acc, val t, const consts &c)
  {
    const double s = acc + t;
    see(s, c);
    if((as_int(acc) + as_int(t)) != as_int(s)) g_chk_fail++;      /* the running sum is exact */
    return s;
  }
  static val dot_fold(val acc, const consts &c) { return reduce(acc, c); }
  /* product at the output of a forward transform (fwd_mul_kernel): exact, bounded, and the sum with an accumulator word
   * stays an integer below 2^53 */
  template <bool LAZY> static val mul_out(val x, uint64_t b, const consts &c)
  {
    see(x, c);
    if(LAZY ? b >= 4 * c.qi : b >= c.qi) g_chk_fail++;
    const double   r  = ArithF64::mul_out<LAZY>(x, b, c);
    const __int128 ex = as_int(x) * (__int128)(b % c.qi);
    if((ex - as_int(r)) % (__int128)c.qi != 0) g_chk_fail++;
    if(__builtin_fabs(r) > 0.88 * c.q) g_chk_fail++;
    return r;
  }
  static uint64_t mul_store(val r, const consts &c) { return store_fwd(r, c); }
  static uint64_t mul_store_acc(val r, uint64_t acc, const consts &c)
  {
    if(acc >= c.qi) g_chk_fail++;
    const double s = r + u64_to_f64_lt52(acc);
    see(s, c);
    if(as_int(r) + (__int128)acc != as_int(s)) g_chk_fail++;
    return store_fwd(s, c);
  }
  static tw expand(ctw w, const consts &c)
  {
    const tw t = ArithF64::expand(w, c);
    /* rebuilt quotient within 2^-52 relative of w/q */
    const long double tq = (long double)w / (long double)c.q;
    if(tq != 0 && __builtin_fabsl(((long double)t.wq - tq) / tq) > 2.3e-16L) g_chk_fail++;
    return t;
  }
  static uint64_t store_fwd(val v, const consts &c)
  {
    const uint64_t u = ArithF64::to_canonical(v, c);
    __int128       m = as_int(v) % (__int128)c.qi;
    if(m < 0) m += c.qi;
    if(u >= c.qi || (__int128)u != m) g_chk_fail++;
    return u;
  }
  static uint64_t store_inv(val v, const consts &c) { return store_fwd(v, c); }
};

/* ------------------------------------------------------------------ */
/* ArithU64XChk<K>: the wide integer policy with its range claims        */
/* (ntt_arith.h, ArithU64X) asserted against 128-bit arithmetic.          */
/* ------------------------------------------------------------------ */
/* g_u64x_worst: every estimated product / reduce_any returns the LARGEST representative its claim allows (below 4q /
 * below 2.01 q) instead of the one the arithmetic happened to produce: still congruent, so results stay right, but
 * every value now grows as fast as u64x_schedule assumes -- a run without a flagged wrap-around then covers the
 * worst case of the schedule, not just the data at hand. */
inline int g_u64x_worst = 0;

template <int K> struct ArithU64XChk : ArithU64X<K> {
  using Base   = ArithU64X<K>;
  using val    = typename Base::val;
  using tw     = typename Base::tw;
  using consts = typename Base::consts;
  using u128   = unsigned __int128;
  static constexpr uint64_t kB = 8u << K;

  static void see(uint64_t v, const consts &c)
  {
    const double b = (double)v / (double)c.q;
    if(b > g_chk_maxb) g_chk_maxb = b;
  }
  static uint64_t shoup_est(const tw &t, uint64_t y, const consts &c)
  {
    uint64_t r = Base::shoup_est(t, y, c);
    if(r >= 4 * c.q || (uint64_t)(((u128)t.w * y) % c.q) != r % c.q) g_chk_fail++;
    if(g_u64x_worst) r += (4 * c.q - 1 - r) / c.q * c.q;
    return r;
  }
  static uint64_t reduce_any(uint64_t v, const consts &c)
  {
    uint64_t r = Base::reduce_any(v, c);
    if((u128)r * 100 >= (u128)c.q * 201 || r % c.q != v % c.q) g_chk_fail++;
    if(g_u64x_worst)
      while((u128)(r + c.q) * 100 < (u128)c.q * 201) r += c.q;
    return r;
  }
  static uint64_t fold_inv(uint64_t s, const consts &c)
  {
    if(K == 0 && s >= 8 * c.q) g_chk_fail++; /* one conditional subtraction of 4q must do */
    return K == 0 ? Base::csub(s, 4 * c.q) : reduce_any(s, c);
  }
  template <bool INV, bool WIDE> static val load(uint64_t raw, const consts &c)
  {
    if(raw >= (WIDE ? 8 : 4) * c.q) g_chk_fail++; /* the caller's contract */
    return Base::template load<INV, WIDE>(raw, c);
  }
  template <bool RED> static void fwd_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    see(x, c);
    if((u128)x >= (u128)kB * c.q) g_chk_fail++;
    const uint64_t x1 = RED ? Base::fold_fwd(x, c) : x;
    const uint64_t m  = shoup_est(t, y, c);
    if((u128)x1 + 4 * (u128)c.q >= ((u128)1 << 64) || (u128)x1 + m >= (u128)kB * c.q) g_chk_fail++; /* no wrap-around, below B q */
    x = x1 + m;
    y = x1 + 4 * c.q - m;
  }
  template <bool RED> static void inv_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    see(x, c);
    see(y, c);
    const uint64_t half = Base::half_range(c);
    if(x >= half || y >= half) g_chk_fail++; /* entry invariant 2b <= B: the sum fits, the offset covers y */
    const uint64_t s = x + y;
    const uint64_t d = x + half - y;
    x                = RED ? fold_inv(s, c) : s;
    y                = shoup_est(t, d, c);
  }
  static void inv_bfly_last(val &x, val &y, const consts &c)
  {
    const uint64_t half = Base::half_range(c);
    if(x >= half || y >= half) g_chk_fail++;
    const uint64_t s = x + y;
    const uint64_t d = x + half - y;
    x                = shoup_est(c.ninv, s, c);
    y                = shoup_est(c.wninv, d, c);
  }
  static uint64_t canon(uint64_t v, const consts &c) { return Base::csub(Base::csub(v, c.q2), c.q); }
  static uint64_t store_fwd(val v, const consts &c)
  {
    const uint64_t u = canon(reduce_any(v, c), c);
    if(u != v % c.q) g_chk_fail++;
    return u;
  }
  static uint64_t store_fwd_lazy(val v, const consts &c) { return reduce_any(v, c); }
  static uint64_t store_inv(val v, const consts &c)
  {
    if(v >= 4 * c.q) g_chk_fail++; /* the last stage of every pass is a folding one */
    return canon(v, c);
  }
  static uint64_t store_inv_lazy(val v, const consts &c)
  {
    if(v >= 4 * c.q) g_chk_fail++;
    return Base::csub(v, c.q2);
  }
  static uint64_t store_fwd_sel(val v, const consts &c, uint64_t keep) { return keep ? store_fwd(v, c) : store_fwd_lazy(v, c); }
  static uint64_t store_inv_sel(val v, const consts &c, uint64_t keep) { return keep ? store_inv(v, c) : store_inv_lazy(v, c); }
  static val      scale_ninv(val v, const consts &c) { return shoup_est(c.ninv, v, c); }
  /* products of stored words: the Barrett form's preconditions and range claim */
  static uint64_t barrett_chk(uint64_t a, uint64_t b, const consts &c)
  {
    const u128 x = (u128)a * b;
    if((x >> (64 + c.bsh)) != 0) g_chk_fail++; /* q1 must fit 64 bits */
    uint64_t hi, lo;
    Base::mul128(a, b, hi, lo);
    if(hi != (uint64_t)(x >> 64) || lo != (uint64_t)x) g_chk_fail++;
    uint64_t r = Base::barrett(hi, lo, c);
    if(r >= 3 * c.q || r % c.q != (uint64_t)(x % c.q)) g_chk_fail++;
    if(g_u64x_worst) r += (3 * c.q - 1 - r) / c.q * c.q;
    return r;
  }
  template <bool LAZY> static val dot_term(uint64_t a, uint64_t b, const consts &c)
  {
    if(a >= (LAZY ? 4 : 1) * c.q || b >= (LAZY ? 4 : 1) * c.q) g_chk_fail++; /* the caller's contract */
    if(LAZY && K <= 1) b = Base::csub(b, c.q2);
    if(LAZY && K == 0) a = Base::csub(a, c.q2);
    return barrett_chk(a, b, c);
  }
  static val dot_acc(val acc, val t, const consts &c)
  {
    if((u128)acc + t >= (u128)kB * c.q) g_chk_fail++;
    see(acc + t, c);
    return acc + t;
  }
  static val dot_fold(val acc, const consts &c) { return reduce_any(acc, c); }
  template <bool LAZY> static val mul_out(val x, uint64_t b, const consts &c)
  {
    if((u128)x >= (u128)kB * c.q || b >= (LAZY ? 4 : 1) * c.q) g_chk_fail++;
    if(LAZY && K <= 1) b = Base::csub(b, c.q2);
    if(LAZY && K == 0) b = Base::csub(b, c.q);
    return barrett_chk(reduce_any(x, c), b, c);
  }
  static uint64_t mul_store(val r, const consts &c)
  {
    if(r >= 4 * c.q) g_chk_fail++;
    return canon(r, c);
  }
  static uint64_t mul_store_acc(val r, uint64_t acc, const consts &c)
  {
    if(acc >= c.q || r + acc >= 4 * c.q) g_chk_fail++;
    return canon(r + acc, c);
  }
};

template <class A> struct Regs {
  typename A::val x[kE];
};

template <class A, int LOGN, bool INV, int KSH, bool LASTINV, bool LAZY = false> static void emu_fused(const Params<A> &p)
{
  using P                 = Plan<LOGN>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, INV, KSH, LAZY>() | (INV && LASTINV ? kLastInvFlag : 0u) | (INV && A::kWide52 ? kCanonInFlag : 0u); /* as fused_kernel */
  std::vector<typename A::val> lds(P::LDS_ELEMS);
  std::vector<Regs<A>>         regs(P::T);
  for(uint64_t b = 0; b < p.nblocks; b++) {
    const uint32_t blk  = (uint32_t)(b & ((1ull << p.s0) - 1));
    uint64_t *     base = p.a + block_offset<LOGN>(b, p.s0, p.pstride, p.ptab); /* as the transform kernels: csrc/ntt_kernels.h blk_off_t */
    if constexpr(!INV) {
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        global_load_first<A, LOGN, false>(regs[t].x, t, base, p.wide, p.c);
        run_group<A, LOGN, 0, false, MASK>(regs[t].x, t, blk, p);
      }
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int G = decltype(gg)::value;
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G + 1>(regs[t].x, t, lds.data());
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
          lds_gather<A, LOGN, G, G + 1>(regs[t].x, t, lds.data());
          if constexpr(A::kCompact && G + 1 == P::NG - 1) {
            /* the device kernel's early-preload path for the last group */
            typename A::ctw pre[4][kE / 2];
            preload_group_tw<A, LOGN, G + 1>(pre, t, blk, p);
            run_group_preloaded<A, LOGN, G + 1, MASK>(regs[t].x, pre, p);
          } else {
            run_group<A, LOGN, G + 1, false, MASK>(regs[t].x, t, blk, p);
          }
        }
      });
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) global_store_last<A, LOGN, false, LAZY>(regs[t].x, t, base, p.c, p.lazy != 0);
    } else {
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        global_load_last<A, LOGN, true>(regs[t].x, t, base, p.wide, p.c);
        run_group<A, LOGN, P::NG - 1, true, MASK>(regs[t].x, t, blk, p);
      }
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int G = P::NG - 1 - decltype(gg)::value; /* writer */
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
          lds_gather<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
          run_group<A, LOGN, G - 1, true, MASK>(regs[t].x, t, blk, p);
        }
      });
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) global_store_first<A, LOGN, true>(regs[t].x, t, base, p.c, p.lazy != 0);
    }
  }
}

/* the fused product kernel (csrc/ntt_kernels.h fused_product_kernel) step by step: forward transform of b, product
 * with a^ in the last group's layout, inverse transform whose per-lane group reads the FORWARD twiddle table in the
 * LDS layout, mirrored (load_stage_tw MIRROR) */
/* BOTH forward transforms inside the product kernels (their BOTH variants): `ahat` then holds a's coefficients and is taken
 * through the same forward stages first (emu_set_product_both) */
inline bool g_prod_both = false;

template <class A, int LOGN, int KSH, bool ALAZY>
void emu_fused_product(uint64_t *out, const uint64_t *ahat, const uint64_t *b, uint64_t batch, const Params<A> &pf,
                              const Params<A> &pi)
{
  using P                  = Plan<LOGN>;
  constexpr uint32_t MASKF = fused_mask<A, LOGN, false, KSH>();
  constexpr uint32_t MASKI = fused_mask<A, LOGN, true, KSH>() | kLastInvFlag;
  constexpr int      GL    = P::NG - 1;
  constexpr int      GT    = GL - 1; /* the group whose twiddles live in the LDS table */
  /* the table as fill_lds_tables lays it out (stage J transposed) */
  constexpr int SG = P::S(GT);
  std::vector<typename A::ctw> table((size_t)(((1 << P::R(GT)) - 1) << SG));
  for(int jj = 0; jj < P::R(GT); jj++) {
    const int              slj = SG + jj;
    const typename A::ctw *src = pf.tw8 + ((size_t)1 << slj);
    for(uint32_t l = 0; l < (1u << slj); l++) {
      const uint32_t u = l & ((1u << jj) - 1u), prefix = l >> jj;
      table[(((1u << jj) - 1u) << SG) + (u << SG) + prefix] = src[l];
    }
  }
  std::vector<typename A::val> lds(P::LDS_ELEMS);
  std::vector<Regs<A>>         regs(P::T), regsa(P::T);
  const bool both = g_prod_both;
  const auto forward = [&](std::vector<Regs<A>> &rg, const uint64_t *src) {
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
      global_load_first<A, LOGN, false>(rg[t].x, t, src, false, pf.c);
      run_group<A, LOGN, 0, false, MASKF>(rg[t].x, t, 0u, pf);
    }
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int G = decltype(gg)::value;
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G + 1>(rg[t].x, t, lds.data());
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        lds_gather<A, LOGN, G, G + 1>(rg[t].x, t, lds.data());
        if constexpr(G + 1 == GL) {
          typename A::ctw pre[4][kE / 2];
          preload_group_tw<A, LOGN, GL>(pre, t, 0u, pf);
          run_group_preloaded<A, LOGN, GL, MASKF>(rg[t].x, pre, pf);
        } else if constexpr(G + 1 == GT) {
          run_group<A, LOGN, G + 1, false, MASKF, true>(rg[t].x, t, 0u, pf, table.data());
        } else {
          run_group<A, LOGN, G + 1, false, MASKF>(rg[t].x, t, 0u, pf);
        }
      }
    });
  };
  for(uint64_t pb = 0; pb < batch; pb++) {
    if(both) forward(regsa, ahat + (pb << LOGN));
    forward(regs, b + (pb << LOGN));
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
      const uint32_t ib = P::IBASE(GL, t);
      for(int e = 0; e < kE; e++) {
        if(both) {
          regs[t].x[e] = A::product_rr(regs[t].x[e], regsa[t].x[e], pf.c);
          continue;
        }
        const uint64_t aw = ahat[(pb << LOGN) + ib + P::IOFF(GL, e)];
        regs[t].x[e]      = A::template product_in_domain<ALAZY>(regs[t].x[e], aw, pf.c);
      }
      typename A::ctw pre[4][kE / 2];
      preload_group_tw<A, LOGN, GL>(pre, t, 0u, pi);
      run_group_preloaded<A, LOGN, GL, MASKI, true>(regs[t].x, pre, pi);
    }
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int G = P::NG - 1 - decltype(gg)::value;
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        lds_gather<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
        if constexpr(G - 1 == GT) {
          run_group<A, LOGN, G - 1, true, MASKI, true, true>(regs[t].x, t, 0u, pi, table.data());
        } else {
          run_group<A, LOGN, G - 1, true, MASKI>(regs[t].x, t, 0u, pi);
        }
      }
    });
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) global_store_first<A, LOGN, true>(regs[t].x, t, out + (pb << LOGN), pf.c);
  }
}

/* fused_product_small_kernel (2^8 .. 2^11): every per-lane group reads its forward table in the LDS layout -- the
 * inverse half mirrored -- and the last group is an ordinary table group too */
template <int LOGN> constexpr bool emu_group_per_lane(int g)
{
  using P = Plan<LOGN>;
  for(int j = 0; j < P::R(g); j++)
    if(P::TW_UNIFORM(g, j)) return false;
  return true;
}
template <int LOGN> constexpr int emu_tbl(int g)
{
  using P = Plan<LOGN>;
  return (g >= 0 && g < P::NG && emu_group_per_lane<LOGN>(g)) ? (((1 << P::R(g)) - 1) << P::S(g)) : 0;
}
template <int LOGN> constexpr int emu_tbl_off(int g)
{
  int o = 0;
  for(int h = 0; h < g; h++) o += emu_tbl<LOGN>(h);
  return o;
}
template <class A, int LOGN, int KSH>
void emu_fused_product_small(uint64_t *out, const uint64_t *ahat, const uint64_t *b, uint64_t batch, const Params<A> &pf,
                             const Params<A> &pi)
{
  using P                  = Plan<LOGN>;
  constexpr uint32_t MASKF = fused_mask<A, LOGN, false, KSH>();
  constexpr uint32_t MASKI = fused_mask<A, LOGN, true, KSH>() | kLastInvFlag;
  constexpr int      GL    = P::NG - 1;
  std::vector<typename A::ctw> table((size_t)emu_tbl_off<LOGN>(P::NG) + 1);
  static_for<0, P::NG>([&](auto gg) {
    constexpr int G = decltype(gg)::value;
    if constexpr(emu_tbl<LOGN>(G) > 0) {
      constexpr int SG = P::S(G);
      for(int jj = 0; jj < P::R(G); jj++) {
        const int              slj = SG + jj;
        const typename A::ctw *src = pf.tw8 + ((size_t)1 << slj);
        for(uint32_t l = 0; l < (1u << slj); l++) {
          const uint32_t u = l & ((1u << jj) - 1u), prefix = l >> jj;
          table[(size_t)emu_tbl_off<LOGN>(G) + (((1u << jj) - 1u) << SG) + (u << SG) + prefix] = src[l];
        }
      }
    }
  });
  std::vector<typename A::val> lds(P::LDS_ELEMS);
  std::vector<Regs<A>>         regs(P::T), regsa(P::T);
  const bool both = g_prod_both;
  const auto forward = [&](std::vector<Regs<A>> &rg, const uint64_t *src) {
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
      global_load_first<A, LOGN, false>(rg[t].x, t, src, false, pf.c);
      run_group<A, LOGN, 0, false, MASKF, (emu_tbl<LOGN>(0) > 0)>(rg[t].x, t, 0u, pf, table.data());
    }
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int G = decltype(gg)::value;
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G + 1>(rg[t].x, t, lds.data());
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        lds_gather<A, LOGN, G, G + 1>(rg[t].x, t, lds.data());
        run_group<A, LOGN, G + 1, false, MASKF, (emu_tbl<LOGN>(G + 1) > 0)>(rg[t].x, t, 0u, pf, table.data() + emu_tbl_off<LOGN>(G + 1));
      }
    });
  };
  for(uint64_t pb = 0; pb < batch; pb++) {
    if(both) forward(regsa, ahat + (pb << LOGN));
    forward(regs, b + (pb << LOGN));
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
      const uint32_t ib = P::IBASE(GL, t);
      for(int e = 0; e < kE; e++) {
        if(both) {
          regs[t].x[e] = A::product_rr(regs[t].x[e], regsa[t].x[e], pf.c);
          continue;
        }
        const uint64_t aw = ahat[(pb << LOGN) + ib + P::IOFF(GL, e)];
        regs[t].x[e]      = A::template product_in_domain<true>(regs[t].x[e], aw, pf.c);
      }
      run_group<A, LOGN, GL, true, MASKI, (emu_tbl<LOGN>(GL) > 0), (emu_tbl<LOGN>(GL) > 0)>(regs[t].x, t, 0u, pi, table.data() + emu_tbl_off<LOGN>(GL));
    }
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int G = P::NG - 1 - decltype(gg)::value;
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        lds_gather<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
        run_group<A, LOGN, G - 1, true, MASKI, (emu_tbl<LOGN>(G - 1) > 0), (emu_tbl<LOGN>(G - 1) > 0)>(regs[t].x, t, 0u, pi,
                                                                                                      table.data() + emu_tbl_off<LOGN>(G - 1));
      }
    });
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) global_store_first<A, LOGN, true>(regs[t].x, t, out + (pb << LOGN), pf.c);
  }
}

/* dot_inv_kernel (csrc/ntt_kernels.h) step by step: the inverse block pass whose inputs are the sums of the element-wise
 * products of k operand pairs given in the NTT domain (last-kind layout), with the running sum folded as the kernel folds it */
template <class A, int LOGN, int KSH, bool LASTINV>
void emu_dot_blocks(const Params<A> &p, int k, const uint64_t *const *a, const uint64_t *const *b, bool lazy, bool bcast)
{
  using P                 = Plan<LOGN>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, true, KSH>() | (LASTINV ? kLastInvFlag : 0u);
  constexpr int      GL   = P::NG - 1;
  std::vector<typename A::val> lds(P::LDS_ELEMS);
  std::vector<Regs<A>>         regs(P::T);
  const uint64_t bmask = (1ull << p.s0) - 1;
  for(uint64_t blkid = 0; blkid < p.nblocks; blkid++) {
    const uint32_t blk  = (uint32_t)(blkid & bmask);
    const uint64_t offa = block_offset<LOGN>(blkid, p.s0, p.pstride);
    const uint64_t offb = bcast ? ((uint64_t)blk << LOGN) : offa; /* a broadcast operand is one dense polynomial */
    uint64_t *     base = p.a + offa;
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
      typename A::val(&x)[kE] = regs[t].x;
      for(int e = 0; e < kE; e++) x[e] = typename A::val{};
      for(int i = 0; i < k; i++) {
        uint64_t ra[kE], rb[kE];
        load_last_raw<LOGN>(ra, t, a[i] + offa);
        load_last_raw<LOGN>(rb, t, b[i] + offb);
        if(i != 0 && i % A::kDotEvery == 0) dot_fold_tile<A>(x, p.c);
        dot_tile<A>(x, ra, rb, lazy, p.c);
      }
      if(k > 1) dot_fold_tile<A>(x, p.c);
    }
    /* (all products are formed before anything is stored: c may alias an operand) */
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
      if constexpr(A::kCompact && P::R(GL) < 4) {
        typename A::ctw pre[4][kE / 2];
        preload_group_tw<A, LOGN, GL>(pre, t, blk, p);
        run_group_preloaded<A, LOGN, GL, MASK, true>(regs[t].x, pre, p);
      } else {
        run_group<A, LOGN, GL, true, MASK>(regs[t].x, t, blk, p);
      }
    }
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int G = P::NG - 1 - decltype(gg)::value;
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        lds_gather<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
        run_group<A, LOGN, G - 1, true, MASK>(regs[t].x, t, blk, p);
      }
    });
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) global_store_first<A, LOGN, true>(regs[t].x, t, base, p.c, !LASTINV);
  }
}

template <class A, int R, bool INV, int KSH>
static void emu_column(uint64_t *a, uint64_t batch, uint32_t logn, uint32_t S, bool wide, bool lastinv,
                       const typename A::tw *tab, const typename A::consts &c, bool lazy_out);

/* fwd_mul_kernel step by step: the forward block pass with the product by b^ (and the accumulator) where the outputs would
 * be reduced and stored */
template <class A, int LOGN, int KSH>
void emu_fwd_mul_blocks(const Params<A> &p, const uint64_t *bhat, uint64_t *out, bool lazy, bool bcast, bool acc)
{
  using P                 = Plan<LOGN>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, false, KSH>();
  constexpr int      GL   = P::NG - 1;
  std::vector<typename A::val> lds(P::LDS_ELEMS);
  std::vector<Regs<A>>         regs(P::T);
  const uint64_t bmask = (1ull << p.s0) - 1;
  for(uint64_t blkid = 0; blkid < p.nblocks; blkid++) {
    const uint32_t  blk  = (uint32_t)(blkid & bmask);
    const uint64_t  offa = block_offset<LOGN>(blkid, p.s0, p.pstride);
    const uint64_t *bblk = bhat + (bcast ? ((uint64_t)blk << LOGN) : offa);
    uint64_t *      cblk = out + offa;
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
      global_load_first<A, LOGN, false>(regs[t].x, t, p.a + offa, false, p.c);
      run_group<A, LOGN, 0, false, MASK>(regs[t].x, t, blk, p);
    }
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int G = decltype(gg)::value;
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G + 1>(regs[t].x, t, lds.data());
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        lds_gather<A, LOGN, G, G + 1>(regs[t].x, t, lds.data());
        if constexpr(A::kCompact && G + 1 == GL) {
          typename A::ctw pre[4][kE / 2];
          preload_group_tw<A, LOGN, GL>(pre, t, blk, p);
          run_group_preloaded<A, LOGN, GL, MASK>(regs[t].x, pre, p);
        } else {
          run_group<A, LOGN, G + 1, false, MASK>(regs[t].x, t, blk, p);
        }
      }
    });
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
      uint64_t rb[kE], rc[kE], u[kE];
      load_last_raw<LOGN>(rb, t, bblk);
      if(acc) load_last_raw<LOGN>(rc, t, cblk);
      else for(int e = 0; e < kE; e++) rc[e] = 0;
      mul_out_tile<A, 0, kE, 1>(u, regs[t].x, rb, rc, lazy, p.c);
      store_last_raw<LOGN>(u, t, cblk);
    }
  }
}

/* onepass_mul_kernel (N = 2^15): the one-pass forward transform -- pair stage, both halves through the block stages with the
 * fifteen-stage schedule -- with fwd_mul_kernel's product where a half would be reduced and stored */
template <class A, int KSH>
void emu_onepass_mul(const Params<A> &pin, const uint64_t *bhat, uint64_t *out, bool lazy, bool bcast, bool acc)
{
  constexpr int LOGN = kFusedLarge;
  using P            = Plan<LOGN>;
  constexpr uint64_t HALF = 1ull << LOGN;
  constexpr uint32_t M15  = onepass_fwd_mask<A, KSH>();
  constexpr uint32_t MASK = M15 >> 1;
  constexpr bool     RED0 = (M15 & 1u) != 0;
  Params<A> p = pin;
  p.s0        = 1;
  std::vector<typename A::val> lds(P::LDS_ELEMS);
  std::vector<Regs<A>>         half[2] = {std::vector<Regs<A>>(P::T), std::vector<Regs<A>>(P::T)};
  for(uint64_t poly = 0; poly < p.nblocks; poly++) {
    const uint64_t off = poly * p.pstride;
    for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
      global_load_first<A, LOGN, false>(half[0][t].x, t, p.a + off, false, p.c);
      global_load_first<A, LOGN, false>(half[1][t].x, t, p.a + off + HALF, false, p.c);
      onepass_pairs_fwd<A, RED0>(half[0][t].x, half[1][t].x, p);
    }
    for(uint32_t h = 0; h < 2; h++) {
      std::vector<Regs<A>> &regs = half[h];
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) run_group<A, LOGN, 0, false, MASK>(regs[t].x, t, h, p);
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int G = decltype(gg)::value;
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G + 1>(regs[t].x, t, lds.data());
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
          lds_gather<A, LOGN, G, G + 1>(regs[t].x, t, lds.data());
          if constexpr(A::kCompact && G + 1 == P::NG - 1) {
            typename A::ctw pre[4][kE / 2];
            preload_group_tw<A, LOGN, G + 1>(pre, t, h, p);
            run_group_preloaded<A, LOGN, G + 1, MASK>(regs[t].x, pre, p);
          } else {
            run_group<A, LOGN, G + 1, false, MASK>(regs[t].x, t, h, p);
          }
        }
      });
      const uint64_t *bblk = bhat + (bcast ? 0 : off) + h * HALF;
      uint64_t *      cblk = out + off + h * HALF;
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        uint64_t rb[kE], rc[kE], u[kE];
        load_last_raw<LOGN>(rb, t, bblk);
        if(acc) load_last_raw<LOGN>(rc, t, cblk);
        else for(int e = 0; e < kE; e++) rc[e] = 0;
        mul_out_tile<A, 0, kE, 1>(u, regs[t].x, rb, rc, lazy, p.c);
        store_last_raw<LOGN>(u, t, cblk);
      }
    }
  }
}

/* the library's fwd_mul (ntt_host.hip): one block launch up to 2^14; 2^15 (FP64 policies): one pass; above, the forward column passes on a, then the blocks */
template <class A, int KSH>
int emu_fwd_mul_run(uint64_t *out, uint64_t *a, const uint64_t *bhat, uint64_t batch, int m, const typename A::tw *tab,
                    const typename A::ctw *tab8, const typename A::consts &c, bool lazy, bool bcast, bool acc)
{
  if(m < kFusedMin) return -1;
#ifdef EMU_SAN_BUILD
  if(m > kFusedMax) return -1;
#else
  if constexpr(A::kCompact && A::kTracksBounds) {
    if(m == kFusedMax + 1 && g_one_pass != 0) {
      Params<A> p{};
      p.a       = a;
      p.tw      = tab;
      p.tw8     = tab8;
      p.c       = c;
      p.logn    = (uint32_t)m;
      p.pstride = emu_pstride(m);
      p.nblocks = batch;
      emu_onepass_mul<A, KSH>(p, bhat, out, lazy, bcast, acc);
      return 0;
    }
  }
#endif
  const int      pblk = m > kFusedMax ? multi_pass_block(m, false, A::kTracksBounds) : m;
  const PassList L    = make_passes(m, false, pblk);
  for(int j = 0; j + 1 < L.n; j++) {
    const Pass &ps = L.p[j];
    switch(ps.r) {
      case 1: emu_column<A, 1, false, KSH>(a, batch, m, ps.s, false, false, tab, c, true); break;
      case 2: emu_column<A, 2, false, KSH>(a, batch, m, ps.s, false, false, tab, c, true); break;
      case 3: emu_column<A, 3, false, KSH>(a, batch, m, ps.s, false, false, tab, c, true); break;
      case 4: emu_column<A, 4, false, KSH>(a, batch, m, ps.s, false, false, tab, c, true); break;
      default: return -1;
    }
  }
  Params<A> p{};
  p.a       = a;
  p.tw      = tab;
  p.tw8     = tab8;
  p.c       = c;
  p.logn    = (uint32_t)m;
  p.pstride = emu_pstride(m);
  p.s0      = (uint32_t)(m - pblk);
  p.nblocks = batch << p.s0;
  switch(pblk) {
#define CASE(LN) \
  case LN: emu_fwd_mul_blocks<A, LN, KSH>(p, bhat, out, lazy, bcast, acc); return 0;
#ifdef EMU_SAN_BUILD
    CASE(6) CASE(8) CASE(12)
#else
    CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#endif
#undef CASE
    default: return -1;
  }
}

/* the library's inv_dot (ntt_host.hip): one block launch up to 2^14; above, the dot kernel over the blocks and the
 * inverse's column passes */
template <class A, int KSH>
int emu_dot_run(uint64_t *out, int k, const uint64_t *const *a, const uint64_t *const *b, uint64_t batch, int m, const typename A::tw *tab,
                const typename A::ctw *tab8, const typename A::consts &c, bool lazy, bool bcast)
{
  if(m < kFusedMin) return -1;
  const int      pblk = m > kFusedMax ? multi_pass_block(m, true, A::kTracksBounds) : m;
  const PassList L    = make_passes(m, false, pblk);
  Params<A>      p{};
  p.a       = out;
  p.tw      = tab;
  p.tw8     = tab8;
  p.c       = c;
  p.logn    = (uint32_t)m;
  p.pstride = emu_pstride(m);
  p.s0      = (uint32_t)(m - pblk);
  p.lastinv = m <= kFusedMax;
  p.nblocks = batch << p.s0;
#ifdef EMU_SAN_BUILD
  if(m > kFusedMax) return -1;
#endif
  if(m > kFusedMax) {
    if(pblk == kFusedSmallBlock) emu_dot_blocks<A, kFusedSmallBlock, KSH, false>(p, k, a, b, lazy, bcast);
    else emu_dot_blocks<A, kFusedLarge, KSH, false>(p, k, a, b, lazy, bcast);
    for(int j = L.n - 2; j >= 0; j--) {
      const Pass &ps = L.p[j];
      const bool  li = ps.s == 0;
      switch(ps.r) {
        case 1: emu_column<A, 1, true, KSH>(out, batch, m, ps.s, false, li, tab, c, j != 0); break;
        case 2: emu_column<A, 2, true, KSH>(out, batch, m, ps.s, false, li, tab, c, j != 0); break;
        case 3: emu_column<A, 3, true, KSH>(out, batch, m, ps.s, false, li, tab, c, j != 0); break;
        case 4: emu_column<A, 4, true, KSH>(out, batch, m, ps.s, false, li, tab, c, j != 0); break;
        default: return -1;
      }
    }
    return 0;
  }
  switch(m) {
#define CASE(LN) \
  case LN: emu_dot_blocks<A, LN, KSH, true>(p, k, a, b, lazy, bcast); return 0;
#ifdef EMU_SAN_BUILD /* sanitizer build: three block sizes keep the instrumented compile short */
    CASE(6) CASE(8) CASE(12)
#else
    CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#endif
#undef CASE
    default: return -1;
  }
}

template <class A, int R, bool INV, int KSH>
static void emu_column(uint64_t *a, uint64_t batch, uint32_t logn, uint32_t S, bool wide, bool lastinv,
                       const typename A::tw *tab, const typename A::consts &c, bool lazy_out)
{
  constexpr uint32_t MASK = column_mask<A, R, INV, KSH>();
  const uint64_t     cols = (1ull << logn) >> R;
  for(uint64_t pidx = 0; pidx < batch; pidx++) {
    for(uint64_t col = 0; col < cols; col++) {
      column_pass_thread<A, R, INV, MASK>((g_ptab ? (uint64_t *)nullptr : a) + poly_offset<false>(pidx, emu_pstride((int)logn), g_ptab), (uint32_t)col, logn, S,
                                          wide, lastinv, tab, c, lazy_out); /* as column_kernel */
    }
  }
}

template <class A, int R, bool INV>
static void emu_column_r4(uint64_t *a, uint64_t batch, uint32_t logn, uint32_t S, const typename A::tw *tab, const typename A::consts &c,
                          bool lazy_out)
{
  const uint64_t cols = (1ull << logn) >> R;
  for(uint64_t pidx = 0; pidx < batch; pidx++) {
    for(uint64_t col = 0; col < cols; col++)
      column_pass_thread_r4<A, R, INV>((g_ptab ? (uint64_t *)nullptr : a) + poly_offset<false>(pidx, emu_pstride((int)logn), g_ptab), (uint32_t)col, logn, S, tab, c,
                                       lazy_out);
  }
}

/* N = 2^15 in one pass (csrc/ntt_kernels.h onepass_kernel), thread by thread: both halves of a polynomial in "registers", the pair
 * stage thread-local (ntt_core.h onepass_pairs_fwd / onepass_pair_inv), each half through the 2^14-point block stages at block
 * position 0 / 1 -- forward with the fifteen-stage reduction schedule (onepass_fwd_mask), which the checked policies verify */
template <class A, bool INV, int KSH> static void emu_onepass(const Params<A> &pin)
{
  constexpr int LOGN = kFusedLarge;
  using P            = Plan<LOGN>;
  constexpr uint64_t HALF = 1ull << LOGN;
  Params<A> p = pin;
  p.s0        = 1;
  std::vector<typename A::val> lds(P::LDS_ELEMS);
  std::vector<Regs<A>>         half[2] = {std::vector<Regs<A>>(P::T), std::vector<Regs<A>>(P::T)};
  for(uint64_t poly = 0; poly < p.nblocks; poly++) {
    uint64_t *base = p.a + poly_offset<true>(poly, p.pstride, p.ptab);
    if constexpr(!INV) {
      constexpr uint32_t M15  = onepass_fwd_mask<A, KSH>();
      constexpr uint32_t MASK = M15 >> 1;
      constexpr bool     RED0 = (M15 & 1u) != 0;
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        global_load_first<A, LOGN, false>(half[0][t].x, t, base, p.wide, p.c);
        global_load_first<A, LOGN, false>(half[1][t].x, t, base + HALF, p.wide, p.c);
        onepass_pairs_fwd<A, RED0>(half[0][t].x, half[1][t].x, p);
      }
      for(uint32_t h = 0; h < 2; h++) {
        std::vector<Regs<A>> &regs = half[h];
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) run_group<A, LOGN, 0, false, MASK>(regs[t].x, t, h, p);
        static_for<0, P::NG - 1>([&](auto gg) {
          constexpr int G = decltype(gg)::value;
          for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G + 1>(regs[t].x, t, lds.data());
          for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
            lds_gather<A, LOGN, G, G + 1>(regs[t].x, t, lds.data());
            if constexpr(A::kCompact && G + 1 == P::NG - 1) {
              typename A::ctw pre[4][kE / 2];
              preload_group_tw<A, LOGN, G + 1>(pre, t, h, p);
              run_group_preloaded<A, LOGN, G + 1, MASK>(regs[t].x, pre, p);
            } else {
              run_group<A, LOGN, G + 1, false, MASK>(regs[t].x, t, h, p);
            }
          }
        });
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) global_store_last<A, LOGN, false, false>(regs[t].x, t, base + h * HALF, p.c, false);
      }
    } else {
      constexpr uint32_t MASK = fused_mask<A, LOGN, true, KSH>() | (A::kWide52 ? kCanonInFlag : 0u); /* as onepass_kernel: not the last pass */
      for(uint32_t h = 0; h < 2; h++) {
        std::vector<Regs<A>> &regs = half[h];
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
          global_load_last<A, LOGN, true>(regs[t].x, t, base + h * HALF, p.wide, p.c);
          run_group<A, LOGN, P::NG - 1, true, MASK>(regs[t].x, t, h, p);
        }
        static_for<0, P::NG - 1>([&](auto gg) {
          constexpr int G = P::NG - 1 - decltype(gg)::value;
          for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
          for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
            lds_gather<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
            run_group<A, LOGN, G - 1, true, MASK>(regs[t].x, t, h, p);
          }
        });
      }
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        static_for<0, kE>([&](auto ee) {
          constexpr int   E  = decltype(ee)::value;
          typename A::val va = half[0][t].x[E], vb = half[1][t].x[E];
          onepass_pair_inv<A>(va, vb, p.c);
          base[((uint32_t)E << P::LT) + t]        = A::store_inv(va, p.c);
          base[HALF + ((uint32_t)E << P::LT) + t] = A::store_inv(vb, p.c);
        });
      }
    }
  }
}

inline bool g_lazy = false; /* lazy outputs for the next emu_transform (set by emu_set_lazy) */

template <class A, bool INV, int KSH>
int emu_run(uint64_t *a, uint64_t batch, int m, const typename A::tw *tab,
            const typename A::consts &c, bool generic, bool wide, const typename A::ctw *tab8 = nullptr)
{
  /* as the library's run_transform */
  const PassList L = A::kRadix4 ? make_passes_r4(m) : make_passes(m, generic, multi_pass_block(m, INV, A::kTracksBounds));
  if(A::kRadix4 && (generic || m > kRadix4Max)) return -4; /* the library refuses these too */
  const bool lazy  = g_lazy;
  if constexpr(A::kCompact && A::kTracksBounds) {
    /* as the library's run_transform: N = 2^15, FP64 policies -> one pass */
#ifndef EMU_SAN_BUILD /* (the sanitizer build keeps its instrumented compile short: two-pass route only) */
    if(m == kFusedMax + 1 && !generic && g_one_pass != 0) { /* (lazy calls too: canonical words satisfy the lazy contract) */
      Params<A> p{};
      p.a       = g_ptab ? nullptr : a;
      p.ptab    = g_ptab;
      p.tw      = tab;
      p.tw8     = tab8;
      p.c       = c;
      p.logn    = (uint32_t)m;
      p.pstride = emu_pstride(m);
      p.wide    = wide;
      p.lastinv = INV;
      p.nblocks = batch;
      emu_onepass<A, INV, KSH>(p);
      return 0;
    }
#endif
  }
  for(int k = 0; k < L.n; k++) {
    const Pass &ps      = L.p[INV ? L.n - 1 - k : k];
    const bool  lastinv = INV && ps.s == 0;
    /* only the first pass of a transform sees caller data */
    const bool w = wide && k == 0;
    /* what the pass stores: as the library's pass_lazy() */
    const bool ends  = k == L.n - 1;
    const bool plazy = ends ? lazy : true;
    if(ps.fused) {
      Params<A> p{};
      p.a       = g_ptab ? nullptr : a;
      p.ptab    = g_ptab;
      p.tw      = tab;
      p.tw8     = tab8;
      p.c       = c;
      if constexpr(A::kRadix4) {
        /* as the library (ntt_host.hip, limbrec_mid): a block pass that does not end the inverse multiplies by 1 */
        if(INV && ps.s != 0) p.c.ninv = h_tw_u64(1, c.q);
      }
      p.logn    = (uint32_t)m;
      p.pstride = emu_pstride(m);
      p.s0      = (uint32_t)ps.s;
      p.wide    = w;
      p.lastinv = lastinv;
      p.lazy    = plazy;
      p.nblocks = batch << ps.s;
      switch(ps.r) {
#define CASE(LN)                                                                            \
  case LN:                                                                                  \
    if(lastinv) emu_fused<A, LN, INV, KSH, true>(p);                                        \
    else if constexpr(!INV && A::kTracksBounds) {                                           \
      if(ends && lazy) emu_fused<A, LN, INV, KSH, false, true>(p);                          \
      else emu_fused<A, LN, INV, KSH, false>(p);                                            \
    } else emu_fused<A, LN, INV, KSH, false>(p);                                            \
    break;
#ifdef EMU_SAN_BUILD /* sanitizer build: a subset of the block sizes keeps the instrumented compile short */
        CASE(6) CASE(7) CASE(8) CASE(9) CASE(12) CASE(14)
#else
        CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#endif
#undef CASE
        default: return -1;
      }
    } else if constexpr(A::kRadix4) {
      switch(ps.r) {
        case 2: emu_column_r4<A, 2, INV>(a, batch, m, ps.s, tab, c, plazy); break;
        case 4: emu_column_r4<A, 4, INV>(a, batch, m, ps.s, tab, c, plazy); break;
        default: return -4; /* (sizes below the block range: no radix-4 form) */
      }
    } else {
      switch(ps.r) {
        case 1: emu_column<A, 1, INV, KSH>(a, batch, m, ps.s, w, lastinv, tab, c, plazy); break;
        case 2: emu_column<A, 2, INV, KSH>(a, batch, m, ps.s, w, lastinv, tab, c, plazy); break;
        case 3: emu_column<A, 3, INV, KSH>(a, batch, m, ps.s, w, lastinv, tab, c, plazy); break;
        case 4: emu_column<A, 4, INV, KSH>(a, batch, m, ps.s, w, lastinv, tab, c, plazy); break;
        default: return -1;
      }
    }
  }
  return 0;
}

/* ---- the parts: explicit instantiations (definitions in parts 1..6, declarations in part 0) ---- */
#define EMU_RUN_ARGS(A) (uint64_t *, uint64_t, int, const typename A::tw *, const typename A::consts &, bool, bool, const typename A::ctw *)
#define EMU_PROD_ARGS(A) (uint64_t *, const uint64_t *, const uint64_t *, uint64_t, const Params<A> &, const Params<A> &)
#define EMU_RUN(KW, A, K)                               \
  KW template int emu_run<A, true, K> EMU_RUN_ARGS(A);  \
  KW template int emu_run<A, false, K> EMU_RUN_ARGS(A);
#define EMU_PROD(KW, A, K)                                               \
  KW template void emu_fused_product<A, 14, K, true> EMU_PROD_ARGS(A);   \
  KW template void emu_fused_product<A, 14, K, false> EMU_PROD_ARGS(A);
#define EMU_PROD_OTHER(KW, A, K)                                               \
  KW template void emu_fused_product<A, 13, K, true> EMU_PROD_ARGS(A);         \
  KW template void emu_fused_product<A, 12, K, true> EMU_PROD_ARGS(A);         \
  KW template void emu_fused_product_small<A, 11, K> EMU_PROD_ARGS(A);         \
  KW template void emu_fused_product_small<A, 10, K> EMU_PROD_ARGS(A);         \
  KW template void emu_fused_product_small<A, 9, K> EMU_PROD_ARGS(A);          \
  KW template void emu_fused_product_small<A, 8, K> EMU_PROD_ARGS(A);
#define EMU_DOT_ARGS(A) (uint64_t *, int, const uint64_t *const *, const uint64_t *const *, uint64_t, int, const typename A::tw *, const typename A::ctw *, const typename A::consts &, bool, bool)
#define EMU_DOT(KW, A, K) KW template int emu_dot_run<A, K> EMU_DOT_ARGS(A);
#define EMU_MUL_ARGS(A) (uint64_t *, uint64_t *, const uint64_t *, uint64_t, int, const typename A::tw *, const typename A::ctw *, const typename A::consts &, bool, bool, bool)
#define EMU_MUL(KW, A, K) KW template int emu_fwd_mul_run<A, K> EMU_MUL_ARGS(A);
using WideChk = WideF64<ArithF64Chk>;
#if EMU_PART >= 0
#  define P1(KW) EMU_RUN(KW, ArithU64, 0) EMU_RUN(KW, ArithU64R4, 0) EMU_RUN(KW, ArithF64W, 0)
#  define P2(KW) EMU_RUN(KW, ArithF64, 0) EMU_RUN(KW, ArithF64, 1)
#  define P3(KW) EMU_RUN(KW, ArithF64, 18) EMU_RUN(KW, ArithF64Chk, 18)
#  define P4(KW) EMU_RUN(KW, ArithF64Chk, 0)
#  define P5(KW) EMU_RUN(KW, ArithF64Chk, 1) EMU_RUN(KW, WideChk, 0)
#  define P6(KW) EMU_PROD(KW, ArithF64, 0) EMU_PROD(KW, ArithF64, 1) EMU_PROD(KW, ArithF64, 18) \
                 EMU_PROD(KW, ArithF64Chk, 0) EMU_PROD(KW, ArithF64Chk, 1) EMU_PROD(KW, ArithF64Chk, 18)
#  define P7(KW) EMU_PROD_OTHER(KW, ArithF64Chk, 0) EMU_PROD_OTHER(KW, ArithF64Chk, 1) EMU_PROD_OTHER(KW, WideChk, 0)
#  define P8(KW) EMU_DOT(KW, ArithU64, 0) EMU_DOT(KW, ArithF64Chk, 0)
#  define P9(KW) EMU_DOT(KW, ArithF64Chk, 1) EMU_DOT(KW, WideChk, 0)
#  define P10(KW) EMU_MUL(KW, ArithU64, 0) EMU_MUL(KW, ArithF64Chk, 0)
#  define P11(KW) EMU_MUL(KW, ArithF64Chk, 1) EMU_MUL(KW, WideChk, 0)
#  define P12(KW) EMU_RUN(KW, ArithU64XChk<0>, 0) EMU_RUN(KW, ArithU64XChk<1>, 1)
#  define P13(KW) EMU_RUN(KW, ArithU64XChk<3>, 3)
#  define P14(KW) EMU_DOT(KW, ArithU64XChk<3>, 3) EMU_DOT(KW, ArithU64XChk<0>, 0) EMU_MUL(KW, ArithU64XChk<3>, 3) EMU_MUL(KW, ArithU64XChk<0>, 0) \
                  EMU_DOT(KW, ArithU64XChk<1>, 1) EMU_MUL(KW, ArithU64XChk<1>, 1)
#  if EMU_PART == 0
P1(extern) P2(extern) P3(extern) P4(extern) P5(extern) P6(extern) P7(extern) P8(extern) P9(extern) P10(extern) P11(extern)
P12(extern) P13(extern) P14(extern)
#  elif EMU_PART == 1
P1()
#  elif EMU_PART == 2
P2()
#  elif EMU_PART == 3
P3()
#  elif EMU_PART == 4
P4()
#  elif EMU_PART == 5
P5()
#  elif EMU_PART == 6
P6()
#  elif EMU_PART == 7
P7()
#  elif EMU_PART == 8
P8()
#  elif EMU_PART == 9
P9()
#  elif EMU_PART == 10
P10()
#  elif EMU_PART == 11
P11()
#  elif EMU_PART == 12
P12()
#  elif EMU_PART == 13
P13()
#  elif EMU_PART == 14
P14()
#  endif
#endif

#if EMU_HAS(0)
/* plan introspection for the layout tests: returns LDS row pad, fills info[]:
 * {NG, R0, RL, T, ROW, LDS_ELEMS, wave_local bits, f64 fwd mask ksh0, f64 inv mask ksh0} */
template <int LOGN> static void plan_info(uint64_t *info)
{
  using P     = Plan<LOGN>;
  info[0]     = P::NG;
  info[1]     = P::R0;
  info[2]     = P::RL;
  info[3]     = P::T;
  info[4]     = P::ROW;
  info[5]     = P::LDS_ELEMS;
  uint64_t wl = 0;
  for(int g = 0; g + 1 < P::NG; g++) wl |= (uint64_t)P::WAVE_LOCAL(g, g + 1) << g;
  info[6] = wl;
  info[7] = fused_mask<ArithF64, LOGN, false, 0>();
  info[8] = fused_mask<ArithF64, LOGN, true, 0>();
  info[9] = P::layout_conflict_free();
}

extern "C" {

/* arith: 0 = U64, 1 = F64 (ksh: -1 = class of q, else forced class <= class of q)
 * returns 0 on success, <0 if the request is not representable */
int emu_transform(uint64_t *a, uint64_t batch, int m, uint64_t q, uint64_t root, int arith,
                  int inverse, int generic, int wide, int ksh_force)
{
  const uint64_t N    = 1ull << m;
  const uint64_t rinv = h_powmod(root, q - 2, q);
  const auto     w    = h_power_table(root, N, q);
  const auto     wi   = h_power_table(rinv, N, q);
  const auto     wix  = h_with_folded_ninv(wi, h_powmod(N % q, q - 2, q), q); /* as the plan uploads it */
  const auto &   src  = inverse ? wix : w;
  if(arith == 0) {
    std::vector<TwU64> tab(src.size());
    for(uint64_t i = 0; i < src.size(); i++) tab[i] = h_tw_u64(src[i], q);
    const auto c = h_consts_u64(q, N, wi);
    return inverse ? emu_run<ArithU64, true, 0>(a, batch, m, tab.data(), c, generic, wide)
                   : emu_run<ArithU64, false, 0>(a, batch, m, tab.data(), c, generic, wide);
  }
  if(arith == 6) { /* the wide integer policy (ArithU64X<K>, K = ksh_force), checked: same tables as arith 0 */
    std::vector<TwU64> tab(src.size());
    for(uint64_t i = 0; i < src.size(); i++) tab[i] = h_tw_u64(src[i], q);
    const auto c = h_consts_u64(q, N, wi);
    if(q < (1ull << 40) || q >= (1ull << (ksh_force == 3 ? 58 : (ksh_force == 1 ? 60 : 61)))) return -2;
#define EMU_U64X(KK)                                                                                            \
  return inverse ? emu_run<ArithU64XChk<KK>, true, KK>(a, batch, m, tab.data(), c, generic, wide)                \
                 : emu_run<ArithU64XChk<KK>, false, KK>(a, batch, m, tab.data(), c, generic, wide);
    if(ksh_force == 3) { EMU_U64X(3) }
#ifndef EMU_SAN_BUILD /* (sanitizer build: one class keeps the instrumented compile short) */
    if(ksh_force == 1) { EMU_U64X(1) }
    if(ksh_force == 0) { EMU_U64X(0) }
#endif
#undef EMU_U64X
    return -3;
  }
  if(arith == 3) { /* the reference's radix-4 formulation on the expanded table */
    if(m < kFusedMin || m > kRadix4Max || generic) return -4;
    const auto         e = h_expand_radix4(inverse ? wi : w, q);
    std::vector<TwU64> tab(e.size());
    for(uint64_t i = 0; i < e.size(); i++) tab[i] = h_tw_u64(e[i], q);
    const auto c = h_consts_u64(q, N, wi);
    return inverse ? emu_run<ArithU64R4, true, 0>(a, batch, m, tab.data(), c, false, wide)
                   : emu_run<ArithU64R4, false, 0>(a, batch, m, tab.data(), c, false, wide);
  }
  if(arith == 4 || arith == 5) { /* the FP64 policy for moduli up to 2^52 (ArithF64W); 5: checked */
    if(!h_f64w_eligible(q)) return -2;
    std::vector<TwF64>  tabw(src.size());
    std::vector<double> tabw8(src.size());
    for(uint64_t i = 0; i < src.size(); i++) {
      tabw[i]  = h_tw_f64(src[i], q);
      tabw8[i] = tabw[i].w;
    }
    const auto cw = h_consts_f64(q, N, wi);
#ifndef EMU_SAN_BUILD
    if(arith == 5)
      return inverse ? emu_run<WideF64<ArithF64Chk>, true, 0>(a, batch, m, tabw.data(), cw, generic, wide, tabw8.data())
                     : emu_run<WideF64<ArithF64Chk>, false, 0>(a, batch, m, tabw.data(), cw, generic, wide, tabw8.data());
#endif
    return inverse ? emu_run<ArithF64W, true, 0>(a, batch, m, tabw.data(), cw, generic, wide, tabw8.data())
                   : emu_run<ArithF64W, false, 0>(a, batch, m, tabw.data(), cw, generic, wide, tabw8.data());
  }
  if(!h_f64_eligible(q)) return -2;
#ifndef EMU_SAN_BUILD
  if(arith == 2) { /* checked FP64 policy */
    std::vector<TwF64>  tabc(src.size());
    std::vector<double> tabc8(src.size());
    for(uint64_t i = 0; i < src.size(); i++) {
      tabc[i]  = h_tw_f64(src[i], q);
      tabc8[i] = tabc[i].w;
    }
    const auto cc = h_consts_f64(q, N, wi);
    const int  kk = h_f64_ksh(q) >= 18 ? 18 : (h_f64_ksh(q) >= 1 ? 1 : 0);
    if(kk == 18) return inverse ? emu_run<ArithF64Chk, true, 18>(a, batch, m, tabc.data(), cc, generic, wide, tabc8.data())
                                : emu_run<ArithF64Chk, false, 18>(a, batch, m, tabc.data(), cc, generic, wide, tabc8.data());
    if(kk == 1) return inverse ? emu_run<ArithF64Chk, true, 1>(a, batch, m, tabc.data(), cc, generic, wide, tabc8.data())
                               : emu_run<ArithF64Chk, false, 1>(a, batch, m, tabc.data(), cc, generic, wide, tabc8.data());
    return inverse ? emu_run<ArithF64Chk, true, 0>(a, batch, m, tabc.data(), cc, generic, wide, tabc8.data())
                   : emu_run<ArithF64Chk, false, 0>(a, batch, m, tabc.data(), cc, generic, wide, tabc8.data());
  }
#endif
  std::vector<TwF64>  tab(src.size());
  std::vector<double> tab8(src.size());
  for(uint64_t i = 0; i < src.size(); i++) {
    tab[i]  = h_tw_f64(src[i], q);
    tab8[i] = tab[i].w;
  }
  const auto c   = h_consts_f64(q, N, wi);
  int        ksh = h_f64_ksh(q);
  if(ksh_force >= 0) {
    if(ksh_force > ksh) return -3;
    ksh = ksh_force;
  }
  const int cls = ksh >= 18 ? 18 : (ksh >= 1 ? 1 : 0);
#define RUN(K)                                                                              \
  return inverse ? emu_run<ArithF64, true, K>(a, batch, m, tab.data(), c, generic, wide, tab8.data())   \
                 : emu_run<ArithF64, false, K>(a, batch, m, tab.data(), c, generic, wide, tab8.data());
#ifndef EMU_SAN_BUILD /* (the sanitizer build runs every modulus through the class-0 schedule: always valid) */
  if(cls == 18) { RUN(18) }
  if(cls == 1) { RUN(1) }
#else
  (void)cls;
#endif
  RUN(0)
#undef RUN
}

void emu_set_lazy(int on) { g_lazy = on != 0; }
void emu_set_one_pass(int mode) { g_one_pass = mode; }
void emu_set_poly_stride(uint64_t words) { g_pstride = words; }
void emu_set_poly_table(const uint64_t *addresses) { g_ptab = addresses; }
void emu_set_operand_stride(uint64_t words) { g_opstride = words; }
void emu_set_u64x_worst(int on) { g_u64x_worst = on != 0; }
void emu_set_product_both(int on) { g_prod_both = on != 0; }

#ifndef EMU_SAN_BUILD
/* out = inv(fwd(b) * ahat) for polynomials of 2^14 points, as the fused product kernel computes it.  ahat: fwd(a),
 * canonical or (a_lazy) lazy in [0,4q).  chk != 0 runs the checked FP64 policy. */
int emu_fused_product14(uint64_t *out, const uint64_t *ahat, const uint64_t *b, uint64_t batch, uint64_t q, uint64_t root,
                        int a_lazy, int chk)
{
  constexpr int  m = 14;
  const uint64_t N = 1ull << m;
  if(!h_f64_eligible(q)) return -2;
  const uint64_t rinv = h_powmod(root, q - 2, q);
  const auto     w    = h_power_table(root, N, q);
  const auto     wi   = h_power_table(rinv, N, q);
  const auto     wix  = h_with_folded_ninv(wi, h_powmod(N % q, q - 2, q), q);
  std::vector<TwF64>  tf(w.size()), ti(wix.size());
  std::vector<double> tf8(w.size()), ti8(wi.size());
  for(size_t i = 0; i < w.size(); i++) {
    tf[i]  = h_tw_f64(w[i], q);
    tf8[i] = tf[i].w;
  }
  for(size_t i = 0; i < wix.size(); i++) ti[i] = h_tw_f64(wix[i], q);
  for(size_t i = 0; i < wi.size(); i++) ti8[i] = h_tw_f64(wi[i], q).w;
  const auto c   = h_consts_f64(q, N, wi);
  const int  ksh = h_f64_ksh(q);
  const int  cls = ksh >= 18 ? 18 : (ksh >= 1 ? 1 : 0);
#define RUNP(POL, K)                                                                                        \
  {                                                                                                         \
    Params<POL> pf{}, pi{};                                                                                 \
    pf.tw = tf.data(); pf.tw8 = tf8.data(); pf.c = c; pf.logn = m;                                           \
    pi = pf; pi.tw = ti.data(); pi.tw8 = ti8.data(); pi.lastinv = 1;                                         \
    if(a_lazy) emu_fused_product<POL, m, K, true>(out, ahat, b, batch, pf, pi);                              \
    else emu_fused_product<POL, m, K, false>(out, ahat, b, batch, pf, pi);                                   \
    return 0;                                                                                               \
  }
  if(chk) {
    if(cls == 18) RUNP(ArithF64Chk, 18)
    if(cls == 1) RUNP(ArithF64Chk, 1)
    RUNP(ArithF64Chk, 0)
  }
  if(cls == 18) RUNP(ArithF64, 18)
  if(cls == 1) RUNP(ArithF64, 1)
  RUNP(ArithF64, 0)
#undef RUNP
}
#endif

#ifndef EMU_SAN_BUILD
/* the same for the product kernels' other sizes (m = 8 .. 13), always with the CHECKED policy and a lazy-or-canonical a^
 * as ntt_fwd_batch_lazy leaves it: fused_product_small_kernel (8..11), fused_product_kernel (12, 13).  Moduli above
 * 2^51(1+2^-10) run the reduce-both-operands policy. */
int emu_fused_product_chk(uint64_t *out, const uint64_t *ahat, const uint64_t *b, uint64_t batch, int m, uint64_t q, uint64_t root)
{
  if(m < 8 || m > 13) return -1;
  const uint64_t N    = 1ull << m;
  const bool     wide = !h_f64_eligible(q);
  if(wide && !h_f64w_eligible(q)) return -2;
  const uint64_t rinv = h_powmod(root, q - 2, q);
  const auto     w    = h_power_table(root, N, q);
  const auto     wi   = h_power_table(rinv, N, q);
  const auto     wix  = h_with_folded_ninv(wi, h_powmod(N % q, q - 2, q), q);
  std::vector<TwF64>  tf(w.size()), ti(wix.size());
  std::vector<double> tf8(w.size()), ti8(wi.size());
  for(size_t i = 0; i < w.size(); i++) {
    tf[i]  = h_tw_f64(w[i], q);
    tf8[i] = tf[i].w;
  }
  for(size_t i = 0; i < wix.size(); i++) ti[i] = h_tw_f64(wix[i], q);
  for(size_t i = 0; i < wi.size(); i++) ti8[i] = h_tw_f64(wi[i], q).w;
  const auto c   = h_consts_f64(q, N, wi);
  const int  cls = wide ? 0 : (h_f64_ksh(q) >= 1 ? 1 : 0);
#define RUNQ(POL, K, FN, LN)                                                                                 \
  {                                                                                                         \
    Params<POL> pf{}, pi{};                                                                                 \
    pf.tw = tf.data(); pf.tw8 = tf8.data(); pf.c = c; pf.logn = LN;                                          \
    pi = pf; pi.tw = ti.data(); pi.tw8 = ti8.data(); pi.lastinv = 1;                                         \
    FN(out, ahat, b, batch, pf, pi);                                                                        \
    return 0;                                                                                               \
  }
#define RUNM(POL, K)                                                                                        \
  switch(m) {                                                                                               \
    case 13: RUNQ(POL, K, (emu_fused_product<POL, 13, K, true>), 13)                                         \
    case 12: RUNQ(POL, K, (emu_fused_product<POL, 12, K, true>), 12)                                         \
    case 11: RUNQ(POL, K, (emu_fused_product_small<POL, 11, K>), 11)                                         \
    case 10: RUNQ(POL, K, (emu_fused_product_small<POL, 10, K>), 10)                                         \
    case 9: RUNQ(POL, K, (emu_fused_product_small<POL, 9, K>), 9)                                            \
    default: RUNQ(POL, K, (emu_fused_product_small<POL, 8, K>), 8)                                           \
  }
  if(wide) { RUNM(WideChk, 0) }
  if(cls == 1) { RUNM(ArithF64Chk, 1) }
  RUNM(ArithF64Chk, 0)
#undef RUNM
#undef RUNQ
}
#endif

/* (sanitizer build: the integer policy and the checked class-0 policy at 2^6, 2^8, 2^12) */
/* out = inv(sum_i a_i (.) b_i) for operands in the NTT domain, as dot_inv_kernel (+ the inverse's column passes above
 * 2^14) computes it.  a: k x [batch][N], b: k x [batch][N] (bcast: k x [N]).  arith 0: integer radix-2; 1: the CHECKED
 * FP64 policy of q's class (the reduce-both-operands policy above 2^51(1+2^-10)). */
int emu_inv_dot(uint64_t *out, int k, const uint64_t *a, const uint64_t *b, uint64_t batch, int m, uint64_t q, uint64_t root, int arith,
                int lazy, int bcast)
{
  const uint64_t N    = 1ull << m;
  const uint64_t rinv = h_powmod(root, q - 2, q);
  const auto     wi   = h_power_table(rinv, N, q);
  const auto     wix  = h_with_folded_ninv(wi, h_powmod(N % q, q - 2, q), q);
  std::vector<const uint64_t *> pa(k), pb(k);
  for(int i = 0; i < k; i++) {
    pa[i] = a + (uint64_t)i * (g_opstride ? g_opstride : batch * N);
    pb[i] = b + (uint64_t)i * (bcast ? N : (g_opstride ? g_opstride : batch * N));
  }
  if(arith == 0) {
    std::vector<TwU64> tab(wix.size());
    for(size_t i = 0; i < wix.size(); i++) tab[i] = h_tw_u64(wix[i], q);
    const auto c = h_consts_u64(q, N, wi);
    return emu_dot_run<ArithU64, 0>(out, k, pa.data(), pb.data(), batch, m, tab.data(), nullptr, c, lazy != 0, bcast != 0);
  }
#ifndef EMU_SAN_BUILD
  if(arith == 6) { /* the wide integer policy, checked, in the class of q (3 below 2^58, 1 below 2^60, else 0) */
    if(q < (1ull << 40) || q >= (1ull << 61)) return -2;
    std::vector<TwU64> tab(wix.size());
    for(size_t i = 0; i < wix.size(); i++) tab[i] = h_tw_u64(wix[i], q);
    const auto c = h_consts_u64(q, N, wi);
    if(q < (1ull << 58)) return emu_dot_run<ArithU64XChk<3>, 3>(out, k, pa.data(), pb.data(), batch, m, tab.data(), nullptr, c, lazy != 0, bcast != 0);
    if(q < (1ull << 60)) return emu_dot_run<ArithU64XChk<1>, 1>(out, k, pa.data(), pb.data(), batch, m, tab.data(), nullptr, c, lazy != 0, bcast != 0);
    return emu_dot_run<ArithU64XChk<0>, 0>(out, k, pa.data(), pb.data(), batch, m, tab.data(), nullptr, c, lazy != 0, bcast != 0);
  }
#endif
  const bool wide = !h_f64_eligible(q);
  if(wide && !h_f64w_eligible(q)) return -2;
  std::vector<TwF64>  ti(wix.size());
  std::vector<double> ti8(wi.size());
  for(size_t i = 0; i < wix.size(); i++) ti[i] = h_tw_f64(wix[i], q);
  for(size_t i = 0; i < wi.size(); i++) ti8[i] = h_tw_f64(wi[i], q).w;
  const auto c = h_consts_f64(q, N, wi);
#ifndef EMU_SAN_BUILD
  if(wide) return emu_dot_run<WideChk, 0>(out, k, pa.data(), pb.data(), batch, m, ti.data(), ti8.data(), c, lazy != 0, bcast != 0);
  if(h_f64_ksh(q) >= 1) return emu_dot_run<ArithF64Chk, 1>(out, k, pa.data(), pb.data(), batch, m, ti.data(), ti8.data(), c, lazy != 0, bcast != 0);
#else
  if(wide) return -2;
#endif
  return emu_dot_run<ArithF64Chk, 0>(out, k, pa.data(), pb.data(), batch, m, ti.data(), ti8.data(), c, lazy != 0, bcast != 0);
}

/* out = fwd(a) (.) b^ (+ out), the result in the NTT domain, as fwd_mul_kernel (+ the forward column passes above 2^14) computes
 * it.  a: [batch][N] coefficients (overwritten above 2^14), b: [batch][N] or (bcast) [N]; arith as emu_inv_dot. */
int emu_fwd_mul(uint64_t *out, uint64_t *a, const uint64_t *b, uint64_t batch, int m, uint64_t q, uint64_t root, int arith, int lazy,
                int bcast, int acc)
{
  const uint64_t N = 1ull << m;
  const auto     w = h_power_table(root, N, q);
  std::vector<uint64_t> dummy(2, 1);
  if(arith == 0) {
    std::vector<TwU64> tab(w.size());
    for(size_t i = 0; i < w.size(); i++) tab[i] = h_tw_u64(w[i], q);
    const auto c = h_consts_u64(q, N, dummy);
    return emu_fwd_mul_run<ArithU64, 0>(out, a, b, batch, m, tab.data(), nullptr, c, lazy != 0, bcast != 0, acc != 0);
  }
#ifndef EMU_SAN_BUILD
  if(arith == 6) {
    if(q < (1ull << 40) || q >= (1ull << 61)) return -2;
    std::vector<TwU64> tab(w.size());
    for(size_t i = 0; i < w.size(); i++) tab[i] = h_tw_u64(w[i], q);
    const auto c = h_consts_u64(q, N, dummy);
    if(q < (1ull << 58)) return emu_fwd_mul_run<ArithU64XChk<3>, 3>(out, a, b, batch, m, tab.data(), nullptr, c, lazy != 0, bcast != 0, acc != 0);
    if(q < (1ull << 60)) return emu_fwd_mul_run<ArithU64XChk<1>, 1>(out, a, b, batch, m, tab.data(), nullptr, c, lazy != 0, bcast != 0, acc != 0);
    return emu_fwd_mul_run<ArithU64XChk<0>, 0>(out, a, b, batch, m, tab.data(), nullptr, c, lazy != 0, bcast != 0, acc != 0);
  }
#endif
  const bool wide = !h_f64_eligible(q);
  if(wide && !h_f64w_eligible(q)) return -2;
  std::vector<TwF64>  tf(w.size());
  std::vector<double> tf8(w.size());
  for(size_t i = 0; i < w.size(); i++) {
    tf[i]  = h_tw_f64(w[i], q);
    tf8[i] = tf[i].w;
  }
  const auto c = h_consts_f64(q, N, dummy);
#ifndef EMU_SAN_BUILD
  if(wide) return emu_fwd_mul_run<WideChk, 0>(out, a, b, batch, m, tf.data(), tf8.data(), c, lazy != 0, bcast != 0, acc != 0);
  if(h_f64_ksh(q) >= 1) return emu_fwd_mul_run<ArithF64Chk, 1>(out, a, b, batch, m, tf.data(), tf8.data(), c, lazy != 0, bcast != 0, acc != 0);
#else
  if(wide) return -2;
#endif
  return emu_fwd_mul_run<ArithF64Chk, 0>(out, a, b, batch, m, tf.data(), tf8.data(), c, lazy != 0, bcast != 0, acc != 0);
}

/* the compile-time fold schedule of the wide integer policy (ntt_arith.h u64x_schedule), for the bound simulation in tests/test_emu.py */
uint32_t emu_u64x_schedule(int inverse, int nstages, int k) { return u64x_schedule(inverse != 0, nstages, k); }

/* the queue-entry decode of the XCD-local kernels (ntt_core.h team_decode): out = {stop, valid, pass, item, v} */
void emu_team_decode(uint32_t k, uint32_t q, uint32_t total, uint32_t lag, uint32_t n0, uint32_t n1, uint32_t n2, uint32_t *out)
{
  const TeamItem it = team_decode(k, q, total, lag, n0, n1, n2);
  out[0] = it.stop;
  out[1] = it.valid;
  out[2] = it.pass;
  out[3] = it.item;
  out[4] = it.v;
}

/* fwd_ntt_radix4x4_lazy at log2 N = 4k+3 through the product's layer functions (ntt_core.h r4x4_layer_*), every butterfly
 * index of a layer in REVERSE order (the layers' butterflies are independent: any order gives the device's result) */
void emu_fwd_r4x4_layers(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *e, const uint64_t *econ)
{
  std::vector<TwU64> rec(2 * N);
  for(uint64_t k = 0; k < 2 * N; k++) rec[k] = TwU64{e[k], econ[k]};
  ArithU64::consts c{};
  c.q  = q;
  c.q2 = 2 * q;
  uint64_t blocks = 1, span = N / 4;
  for(; blocks < (N >> 3); blocks *= 4, span /= 4) {
    for(uint64_t id = N / 4; id-- > 0;) r4x4_layer_r4(a, rec.data(), blocks, span, id, c);
  }
  for(uint64_t id = N / 2; id-- > 0;) r4x4_layer_r2(a, rec.data(), N, id, c);
  for(uint64_t id = N / 4; id-- > 0;) r4x4_layer_r4(a, rec.data(), N / 4, 1, id, c);
}

/* the product's host-side builder of the 2N-entry radix-4 table (ntt_tables.h), for the table tests */
void emu_expand_radix4(uint64_t *e, const uint64_t *w, uint64_t N, uint64_t q)
{
  const auto v = h_expand_radix4(std::vector<uint64_t>(w, w + N), q);
  memcpy(e, v.data(), v.size() * sizeof(uint64_t));
}

/* counters of the checked policy: out = {violations, max |v|/q * 1e6, max |product|/q * 1e6}; reset */
void emu_chk_stats(uint64_t *out, int reset)
{
  out[0] = g_chk_fail;
  out[1] = (uint64_t)(g_chk_maxb * 1e6);
  out[2] = (uint64_t)(g_chk_maxr * 1e6);
  if(reset) {
    g_chk_fail = 0;
    g_chk_maxb = g_chk_maxr = 0;
  }
}

/* pointwise product through both arithmetic policies */
int emu_pointwise(uint64_t *c_out, const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t q, int arith)
{
  std::vector<uint64_t> dummy(2, 1);
  if(arith == 0) {
    const auto c = h_consts_u64(q, 2, dummy);
    for(uint64_t i = 0; i < n; i++) c_out[i] = ArithU64::mulmod_full(a[i], b[i], c);
    return 0;
  }
  if(!h_f64_eligible(q)) return -2;
  const auto c = h_consts_f64(q, 2, dummy);
  for(uint64_t i = 0; i < n; i++) c_out[i] = ArithF64::mulmod_full(a[i], b[i], c);
  return 0;
}

/* pointwise product of LAZY operands in [0,4q) */
int emu_pointwise_lazy(uint64_t *c_out, const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t q, int arith)
{
  std::vector<uint64_t> dummy(2, 1);
  if(arith == 0) {
    const auto c = h_consts_u64(q, 2, dummy);
    for(uint64_t i = 0; i < n; i++) c_out[i] = ArithU64::mulmod_full_lazy4(a[i], b[i], c);
    return 0;
  }
  if(!h_f64_eligible(q)) return -2;
  const auto c = h_consts_f64(q, 2, dummy);
  for(uint64_t i = 0; i < n; i++) c_out[i] = ArithF64::mulmod_full_lazy4(a[i], b[i], c);
  return 0;
}

int emu_plan_info(int logn, uint64_t *info)
{
  switch(logn) {
#define CASE(LN) \
  case LN: plan_info<LN>(info); return 0;
    CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
    default: return -1;
  }
}
}
#endif /* EMU_HAS(0) */
