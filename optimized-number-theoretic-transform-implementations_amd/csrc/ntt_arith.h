/*
 * ntt_arith.h -- modular arithmetic policies for the gfx950 NTT kernels.
 *
 * Interchangeable policies drive the same butterfly network (ntt_core.h):
 *
 *  ArithU64    exact restatement of the reference's Harvey/Shoup lazy arithmetic
 *              (reference include/internal/fast_mul_operators.h:15-106) on 64-bit
 *              integers; valid for every q with 4q < 2^64.  On gfx950 one butterfly is
 *              28 VALU instructions, 10 of them 32-bit multiplies (profiles/r02/ablations.txt).
 *  ArithU64R4  the reference's radix-4 butterflies with the shared-quotient double product
 *              (:62-70, :108-149) on the 2N-entry expanded table: bit-identical lazy values
 *              for the *_radix4 entry points.
 *  ArithF64    the MI355X fast path for q <= 2^51(1+2^-10): coefficients are kept
 *              as integer-valued doubles in balanced form, a product is
 *              h=t*w, l=fma(t,w,-h), k=rint(t*(w/q)), r=fma(-k,q,h)+l
 *              (6 FP64 ops, every step exact -- proof in DESIGN.md section 4), and the
 *              conditional subtracts of the Harvey butterfly become a
 *              compile-time schedule of rint-reductions.  Final outputs are
 *              reduced to [0,q), so results are bit-identical to the reference
 *              (SURVEY A.6).  No MFMA: this is element-wise 53-bit work.
 *  ArithF64W   (= WideF64<ArithF64>) the same for q up to 2^52: both operands of every
 *              butterfly are reduced first (14 instead of 8 FP64 instructions).
 *
 * Everything here is NTT_HD (host+device) so tests/emu can run the identical
 * code on the CPU against the oracle.
 */
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#  include <hip/hip_runtime.h>
#  define NTT_HD __host__ __device__ __forceinline__
#  define NTT_DEVICE_CODE 1
#else
#  define NTT_HD inline __attribute__((always_inline))
#endif

namespace ntt {

/* 16-byte twiddle record: one dwordx4 load fetches the multiplier and its
 * precomputed quotient helper. */
struct alignas(16) TwU64 {
  uint64_t w;   /* w in [0,q)                        */
  uint64_t con; /* floor(w * 2^64 / q)  (pre_compute.h:68-77 semantics) */
};
struct alignas(16) TwF64 {
  double w;  /* balanced representative of w, in (-q/2, q/2] */
  double wq; /* fl(w / q)                                     */
};

NTT_HD uint64_t mulhi64(uint64_t a, uint64_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}

NTT_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }
NTT_HD double rint_(double a) { return __builtin_rint(a); }

/* ------------------------------------------------------------------ */
/* ArithU64                                                            */
/* ------------------------------------------------------------------ */
struct ArithU64 {
  using val = uint64_t;
  using tw  = TwU64;
  struct consts {
    uint64_t q, q2;
    TwU64    ninv;  /* N^-1 and its precon                      */
    TwU64    wninv; /* N^-1 * winv[1] merged (ntt_reference.c:55-61) */
    TwU64    r64;   /* 2^64 mod q and its precon (pointwise product)  */
    TwU64    one;   /* {1, floor(2^64/q)}: Shoup form of "t mod q"    */
    uint64_t bmu;   /* floor(2^(64+bsh) / q): Barrett quotient of a 128-bit product (ArithU64X) */
    uint32_t bsh;   /* bit length of q, minus 1 */
    uint32_t pad_;
  };
  /* value-range bookkeeping is static for this policy: [0,4q) fwd, [0,2q) inv */
  static constexpr bool kTracksBounds = false;
  /* no 8-byte twiddle form: the Shoup quotient cannot be rebuilt without a division */
  static constexpr bool kCompact = false;
  static constexpr bool kRadix4  = false;
  static constexpr bool kWide52  = false;
  static constexpr bool kIntWide = false; /* ArithU64X: MASK = stages that fold the growing operand (u64x_schedule) */
  using ctw                      = uint64_t;
  static NTT_HD tw expand(ctw w, const consts &) { return tw{w, 0}; }

  static NTT_HD uint64_t csub(uint64_t v, uint64_t b) { return v < b ? v : v - b; }

  /* inputs may be anywhere in [0,8q) (the lazy radix-4 output range the
   * reference's bench feeds back in, tests/bench.c:123-137) */
  template <bool INV, bool WIDE> static NTT_HD val load(uint64_t raw, const consts &c)
  {
    if(!WIDE) return raw;                    /* strict API: [0,q)         */
    raw = csub(raw, 2 * c.q2);               /* -> [0,4q): forward range  */
    return INV ? csub(raw, c.q2) : raw;      /* -> [0,2q): inverse range  */
  }
  static NTT_HD uint64_t shoup(const tw &t, uint64_t y, const consts &c)
  {
    return t.w * y - mulhi64(t.con, y) * c.q; /* [0,2q) */
  }
  /* fast_mul_operators.h:72-81 */
  template <bool RED> static NTT_HD void fwd_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    const uint64_t x1 = csub(x, c.q2);
    const uint64_t m  = shoup(t, y, c);
    x                 = x1 + m;
    y                 = x1 - m + c.q2;
  }
  /* fast_mul_operators.h:83-92 */
  template <bool RED> static NTT_HD void inv_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    const uint64_t s = csub(x + y, c.q2);
    const uint64_t d = x - y + c.q2;
    x                = s;
    y                = shoup(t, d, c);
  }
  /* fast_mul_operators.h:94-106: last inverse stage, N^-1 folded in */
  static NTT_HD void inv_bfly_last(val &x, val &y, const consts &c)
  {
    const uint64_t s = x + y;
    const uint64_t d = x - y + c.q2;
    x                = shoup(c.ninv, s, c);
    y                = shoup(c.wninv, d, c);
  }
  static NTT_HD uint64_t store_fwd(val v, const consts &c) { return csub(csub(v, c.q2), c.q); }
  /* lazy outputs: the reference's documented ranges (include/ntt_reference.h:13-17): forward [0,4q),
   * inverse [0,2q) -- the conditional subtracts of the final reduction are left to the consumer */
  static NTT_HD uint64_t store_fwd_lazy(val v, const consts &) { return v; } /* < 4q */
  static NTT_HD uint64_t store_inv_lazy(val v, const consts &) { return v; } /* < 2q */
  /* both forms in one: keep = ~0 reduces, keep = 0 leaves the lazy value (a conditional subtract of 0 is the
   * identity) -- the launch-uniform choice costs a scalar AND instead of a branch or a second kernel */
  static NTT_HD uint64_t store_fwd_sel(val v, const consts &c, uint64_t keep) { return csub(csub(v, c.q2 & keep), c.q & keep); }
  static NTT_HD uint64_t store_inv_sel(val v, const consts &c, uint64_t keep) { return csub(v, c.q & keep); }
  static NTT_HD uint64_t store_inv(val v, const consts &c) { return csub(v, c.q); }
  static NTT_HD val      scale_ninv(val v, const consts &c) { return shoup(c.ninv, v, c); }
  /* pointwise product of two values in [0,q), result in [0,q).  The 128-bit
   * product hi:lo is folded with two Shoup products:
   * hi*2^64 + lo = hi*(2^64 mod q) + (lo mod q)  (mod q). */
  static NTT_HD uint64_t mulmod_full(uint64_t a, uint64_t b, const consts &c)
  {
    const uint64_t hi = mulhi64(a, b), lo = a * b;
    const uint64_t r  = shoup(c.r64, hi, c) + shoup(c.one, lo, c); /* < 4q */
    return csub(csub(r, c.q2), c.q);
  }
  /* lazy operands (any 64-bit values): the two Shoup folds accept them as they are */
  static NTT_HD uint64_t mulmod_full_lazy4(uint64_t a, uint64_t b, const consts &c) { return mulmod_full(a, b, c); }
  /* Inner product in the NTT domain, c = inv(sum_i a_i^ (.) b_i^), in front of the inverse transform's first stage
   * (dot_inv_kernel; generalises fast_mul_mod_q, include/internal/fast_mul_operators.h:56-60, to operand pairs).  One term
   * from two stored words -- canonical or lazy alike for this policy -- is fully reduced; the running sum is kept in
   * [0,q) by one conditional subtract per term: a valid input ([0,2q)) of the inverse butterflies. */
  static constexpr int kDotEvery = 1 << 30; /* terms between two folds of the running sum: never needed */
  static constexpr int kDotChunk = 2;       /* products the kernel lets the scheduler interleave (register budget) */
  template <bool LAZY> static NTT_HD val dot_term(uint64_t a, uint64_t b, const consts &c) { return mulmod_full(a, b, c); }
  static NTT_HD val dot_acc(val acc, val t, const consts &c) { return csub(acc + t, c.q); }
  static NTT_HD val dot_fold(val acc, const consts &) { return acc; }
  /* Product at the OUTPUT of a forward transform: c^ = fwd(a) (.) b^ (+ c^), the result staying in the NTT domain
   * (fwd_mul_kernel).  x: the forward transform's working value ([0,4q)), b: a stored word (canonical, or lazy); the
   * product is fully reduced (fast_mul_mod_q semantics, fast_mul_operators.h:56-60); acc: a canonical word. */
  template <bool LAZY> static NTT_HD val mul_out(val x, uint64_t b, const consts &c) { return mulmod_full(x, b, c); }
  static NTT_HD uint64_t mul_store(val r, const consts &) { return r; }
  static NTT_HD uint64_t mul_store_acc(val r, uint64_t acc, const consts &c) { return csub(r + acc, c.q); }
};

/* ------------------------------------------------------------------ */
/* ArithU64R4: the reference's radix-4 formulation on the device        */
/* ------------------------------------------------------------------ */
/*
 * Same integer arithmetic, but two stages at a time with the reference's radix-4 butterflies
 * (include/internal/fast_mul_operators.h:108-149) and their shared-quotient double product
 * (:62-70).  The plan's table is the reference's 2N-entry EXPANDED table (pre_compute.h:85-105)
 * as 16-byte records {e[k], e_con[k]}: for the radix-2 slot s of the upper stage of a pair,
 *   record 2s          = W1            (collect_roots, src/ntt_radix4.c:7-25: w[m1], m1 = 2(m+j))
 *   records 4s .. 4s+3 = W2, W1*W2, W3, -W1*W3        (w[2*m1 .. 2*m1+3])
 * Every butterfly is the reference's, applied to the same operands in the same order, so even the
 * LAZY outputs ([0,8q), or [0,4q) when log2 N is odd) are bit-identical to fwd_ntt_radix4_lazy.
 * Values travel between stage groups in [0,8q) (forward) / [0,2q) (inverse).
 */
struct ArithU64R4 : ArithU64 {
  static constexpr bool kRadix4 = true;
  struct pack {
    TwU64 w1, w2, w12, w3, nw13;
  };

  /* fast_mul_operators.h:62-70: Q = hi64(con1*t1 + con2*t2) over the 128-bit sum (mod 2^128, like
   * the reference's __uint128_t arithmetic); result t1*w1 + t2*w2 - Q*q in [0,2q) */
  static NTT_HD uint64_t dbl_shoup(const TwU64 &a, const TwU64 &b, uint64_t t1, uint64_t t2, const consts &c)
  {
    const uint64_t lo1 = a.con * t1, lo2 = b.con * t2;
    const uint64_t lo  = lo1 + lo2;
    const uint64_t Q   = mulhi64(a.con, t1) + mulhi64(b.con, t2) + (lo < lo1 ? 1u : 0u);
    return t1 * a.w + t2 * b.w - Q * c.q;
  }
  /* fast_mul_operators.h:56-60 */
  static NTT_HD uint64_t shoup_q(const TwU64 &t, uint64_t y, const consts &c) { return csub(shoup(t, y, c), c.q); }

  /* forward inputs are used as they come (the radix-4 butterfly takes [0,8q), :108-128); the inverse
   * starts from [0,2q) (src/ntt_radix4.c:78-81) */
  template <bool INV, bool WIDE> static NTT_HD val load(uint64_t raw, const consts &c)
  {
    if(!INV || !WIDE) return raw;
    return csub(csub(raw, 2 * c.q2), c.q2);
  }
  /* fast_mul_operators.h:108-128.  X = a[i], Y = a[i+t], Z = a[i+2t], T = a[i+3t] */
  static NTT_HD void r4_fwd(val &X, val &Y, val &Z, val &T, const pack &w, const consts &c)
  {
    const uint64_t q4 = 2 * c.q2;
    const uint64_t Y1 = dbl_shoup(w.w2, w.w12, Y, T, c);
    const uint64_t Y2 = dbl_shoup(w.w3, w.nw13, Y, T, c);
    const uint64_t T1 = csub(X, q4);
    const uint64_t T2 = shoup(w.w1, Z, c);
    X                 = T1 + T2 + Y1;
    Y                 = (T1 + T2 - Y1) + c.q2;
    Z                 = (T1 - T2 + Y2) + c.q2;
    T                 = (T1 - T2 - Y2) + q4;
  }
  /* fast_mul_operators.h:130-149 */
  static NTT_HD void r4_inv(val &X, val &Y, val &Z, val &T, const pack &w, const consts &c)
  {
    const uint64_t q4 = 2 * c.q2;
    const uint64_t T0 = Z + T;
    const uint64_t T1 = X + Y;
    const uint64_t T2 = q4 + X - Y;
    const uint64_t T3 = q4 + Z - T;
    X                 = csub(csub(T1 + T0, q4), c.q2);
    Z                 = shoup_q(w.w1, q4 + T1 - T0, c);
    Y                 = dbl_shoup(w.w2, w.w3, T2, T3, c);
    T                 = dbl_shoup(w.w12, w.nw13, T2, T3, c);
  }
  /* trailing radix-2 stage of the forward transform when log2 N is odd (src/ntt_radix4.c:56-61) */
  static NTT_HD void r2_fwd_tail(val &X, val &Y, const TwU64 &w, const consts &c)
  {
    X = csub(X, 2 * c.q2);
    ArithU64::fwd_bfly<false>(X, Y, w, c);
  }
  /* leading radix-2 stage of the inverse when log2 N is odd (src/ntt_radix4.c:85-93) */
  static NTT_HD void r2_inv_head(val &X, val &Y, const TwU64 &w, const consts &c)
  {
    X = csub(X, 2 * c.q2);
    ArithU64::inv_bfly<false>(X, Y, w, c);
  }
  static NTT_HD uint64_t store_fwd(val v, const consts &c) { return csub(csub(csub(v, 2 * c.q2), c.q2), c.q); }
  static NTT_HD uint64_t store_fwd_lazy(val v, const consts &) { return v; } /* < 8q */
  /* final normalisation pass of the inverse, fused into the store (src/ntt_radix4.c:111-113) */
  static NTT_HD uint64_t store_inv(val v, const consts &c) { return shoup_q(c.ninv, v, c); }
  static NTT_HD uint64_t store_inv_lazy(val v, const consts &c) { return shoup(c.ninv, v, c); } /* < 2q */
  static NTT_HD uint64_t store_fwd_sel(val v, const consts &c, uint64_t keep)
  {
    return csub(csub(csub(v, (2 * c.q2) & keep), c.q2 & keep), c.q & keep);
  }
  static NTT_HD uint64_t store_inv_sel(val v, const consts &c, uint64_t keep) { return csub(shoup(c.ninv, v, c), c.q & keep); }
};

/* ------------------------------------------------------------------ */
/* ArithU64X<K>: integer arithmetic that spends the headroom above q     */
/* ------------------------------------------------------------------ */
/*
 * The throughput form of the integer policy for the moduli the FP64 policies cannot serve (2^52 <= q < 2^61; round 4).
 * Same tables, same canonical results as ArithU64 -- other lazy words, so the reference-signature entry points and plans
 * created with NTT_ARITH_U64 keep the reference's butterflies (ArithU64) and only NTT_ARITH_AUTO plans run this one.
 * Two changes to fast_mul_operators.h:49-92, both paid for with the bits between q and 2^64 (B = 8 * 2^K values of q fit:
 * K = 0 for q < 2^61, 1 for q < 2^60, 3 for q < 2^58):
 *   1. ESTIMATED Shoup quotient: with con = c1 2^32 + c0, y = t1 2^32 + t0,
 *        Q' = c1 t1 + hi32(c1 t0) + hi32(c0 t1)          (3 multiplies instead of the 4 + carries of hi64(con * y))
 *      drops floor(c0 t0 / 2^32) and two carries: Q - 2 <= Q' <= Q, so w y - Q' q lies in [0,4q) instead of [0,2q).
 *   2. NO conditional subtraction per butterfly.  Forward: x' = x + m, y' = x + 4q - m with m in [0,4q): a value grows
 *      by 4q per stage and is folded (x >= (B/2) q ? x - (B/2) q : x) only at the stages u64x_schedule names -- never
 *      in a 14-stage block for K = 3 (4 + 4 * 14 = 60 < 64), every other stage for K = 1, every stage but the first
 *      for K = 0.  The Shoup product takes ANY 64-bit y.  Inverse: s = x + y doubles, d = x - y + (B/2) q stays below
 *      B q, the product comes back below 4q; the sum is brought below 4q by reduce_any (one 32-bit multiply estimates
 *      s / q) where the schedule says so: every fourth stage for K = 3, every second for K = 1, and the LAST stage of
 *      every pass, so that a pass hands on words below 4q whatever it computed.
 * Per butterfly: 19 VALU instructions (9 multiplies) against ArithU64's 28 (10); measured +16..17 % at 2^12 / 2^14 on the
 * forward transform (profiles/r04/ab_int_wide.txt), where the multiplies are 56 % of the issue time.
 * Outputs: forward values below (4 + 4 stages) q are reduced by reduce_any + two conditional subtractions (lazy: reduce_any
 * alone, < 4q: the lazy contract of include/ntt_reference.h:13-17); inverse values leave the last stage below 4q.
 * reduce_any needs floor(2^64 / q) < 2^32 and its error term 2^32 / q small: q >= 2^40 (ntt_host.hip: int_wide_class).
 */
constexpr uint32_t u64x_schedule(bool inverse, int nstages, int K)
{
  const int B    = 8 << K; /* values stay below B q < 2^64 */
  uint32_t  mask = 0;
  int       b    = 4;      /* every pass starts from words below 4q (canonical, lazy, or the previous pass's) */
  for(int s = 0; s < nstages; s++) {
    if(!inverse) {
      if(b + 4 > B) {
        mask |= 1u << s;
        b = b - B / 2 > B / 2 ? b - B / 2 : B / 2;
      }
      b += 4;
    } else {
      /* entry invariant 2b <= B: the sum fits, the difference's offset (B/2) q covers y */
      if(s == nstages - 1 || 4 * b > B) {
        mask |= 1u << s;
        b = 4;
      } else {
        b = 2 * b;
      }
    }
  }
  return mask;
}

template <int K> struct ArithU64X : ArithU64 {
  static_assert(K == 0 || K == 1 || K == 3, "headroom classes: B = 8, 16, 64 multiples of q below 2^64");
  static constexpr bool kIntWide = true;
  static constexpr int  kHead    = K;

  static NTT_HD uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
  static NTT_HD uint64_t half_range(const consts &c) { return c.q << (K + 2); } /* (B/2) q */

  /* w y - Q' q in [0,4q) for ANY 64-bit y (see above) */
  static NTT_HD uint64_t shoup_est(const tw &t, uint64_t y, const consts &c)
  {
    const uint32_t c1 = (uint32_t)(t.con >> 32), c0 = (uint32_t)t.con, t1 = (uint32_t)(y >> 32), t0 = (uint32_t)y;
    const uint64_t Q  = (uint64_t)c1 * t1 + mulhi32(c1, t0) + mulhi32(c0, t1);
    return t.w * y - Q * c.q;
  }
  /* any 64-bit v -> [0, 2.01 q), congruent: Q' = hi32(hi32(v) * floor(2^64/q)) satisfies v/q - 2 - 2^32/q < Q' <= v/q
   * (hi32(v) > v/2^32 - 1 and floor(2^64/q) > 2^64/q - 1 give hi32(v) floor(2^64/q) / 2^32 > v/q - v/2^64 - 2^32/q, v/2^64 < 1,
   * and the outer floor loses less than 1); the policy is used for q >= 2^40: 2^32/q <= 2^-8 */
  static NTT_HD uint64_t reduce_any(uint64_t v, const consts &c)
  {
    const uint32_t Q = mulhi32((uint32_t)(v >> 32), (uint32_t)c.one.con);
    return v - (uint64_t)Q * c.q;
  }
  /* the operand the schedule folds: forward x below B q -> below max(b, B) - B/2 ... (u64x_schedule); K = 0 is ArithU64's
   * conditional subtraction at doubled ranges */
  static NTT_HD uint64_t fold_fwd(uint64_t x, const consts &c) { return csub(x, half_range(c)); }
  static NTT_HD uint64_t fold_inv(uint64_t s, const consts &c) { return K == 0 ? csub(s, 2 * c.q2) : reduce_any(s, c); }

  template <bool INV, bool WIDE> static NTT_HD val load(uint64_t raw, const consts &c)
  {
    if(!WIDE) return raw;         /* [0,q), or a lazy word below 4q */
    return csub(raw, 2 * c.q2);   /* [0,8q) -> [0,4q) */
  }
  template <bool RED> static NTT_HD void fwd_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    const uint64_t x1 = RED ? fold_fwd(x, c) : x;
    const uint64_t m  = shoup_est(t, y, c);
    x                 = x1 + m;
    y                 = x1 + 2 * c.q2 - m;
  }
  template <bool RED> static NTT_HD void inv_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    const uint64_t s = x + y;
    const uint64_t d = x + half_range(c) - y;
    x                = RED ? fold_inv(s, c) : s;
    y                = shoup_est(t, d, c);
  }
  /* last inverse stage with N^-1 folded in (fast_mul_operators.h:94-106): both outputs are products, below 4q */
  static NTT_HD void inv_bfly_last(val &x, val &y, const consts &c)
  {
    const uint64_t s = x + y;
    const uint64_t d = x + half_range(c) - y;
    x                = shoup_est(c.ninv, s, c);
    y                = shoup_est(c.wninv, d, c);
  }
  static NTT_HD uint64_t store_fwd(val v, const consts &c) { return csub(csub(reduce_any(v, c), c.q2), c.q); }
  static NTT_HD uint64_t store_fwd_lazy(val v, const consts &c) { return reduce_any(v, c); } /* < 4q */
  static NTT_HD uint64_t store_inv(val v, const consts &c) { return csub(csub(v, c.q2), c.q); } /* the last stage left < 4q */
  static NTT_HD uint64_t store_inv_lazy(val v, const consts &c) { return csub(v, c.q2); } /* < 2q: the lazy inverse range */
  static NTT_HD uint64_t store_fwd_sel(val v, const consts &c, uint64_t keep)
  {
    return csub(csub(reduce_any(v, c), c.q2 & keep), c.q & keep);
  }
  static NTT_HD uint64_t store_inv_sel(val v, const consts &c, uint64_t keep) { return csub(csub(v, c.q2), c.q & keep); }
  static NTT_HD val      scale_ninv(val v, const consts &c) { return shoup_est(c.ninv, v, c); }

  /* ---- products of two stored words (dot_inv_kernel, fwd_mul_kernel): one Barrett reduction of the 128-bit product ----
   * ArithU64's fast_mul_mod_q form folds hi * 2^64 + lo with two Shoup products (27 32-bit multiplies per product); here
   *   x = a b = hi:lo < 2^(64+bsh),  q1 = floor(x / 2^bsh) < 2^64,  q3 = hi64(q1 * bmu),  r = lo - q3 q   (11 multiplies)
   * with bmu = floor(2^(64+bsh) / q): q1 bmu / 2^64 <= x/q, and > (x/2^bsh - 1)(2^(64+bsh)/q - 1)/2^64 > x/q - 2 (x < 2^(64+bsh),
   * 2^bsh < q), one more for the floor: x/q - 3 < q3 <= x/q, so r lies in [0,3q).  Operand ranges: canonical words (a b < q^2
   * < 2^(2 bsh + 2)) always fit; lazy words below 4q fit as they are for K = 3 (16 q^2 < 2^(2 bsh + 6), bsh <= 57), after
   * one / two conditional subtractions for K = 1 / K = 0.  The running sum of an inner product is NOT reduced per term:
   * it is folded by reduce_any every kDotEvery terms (2.01 + 3 * 20 < 64, 2.01 + 3 * 4 < 16, 2.01 + 3 < 8) and at the end,
   * which leaves a valid input of the inverse stages (< 4q; a single term, < 3q, is one as it is). */
  static NTT_HD void mul128(uint64_t a, uint64_t b, uint64_t &hi, uint64_t &lo)
  {
    const uint64_t a0 = (uint32_t)a, a1 = a >> 32, b0 = (uint32_t)b, b1 = b >> 32;
    const uint64_t p00 = a0 * b0, p01 = a0 * b1, p10 = a1 * b0, p11 = a1 * b1;
    const uint64_t mid = (p00 >> 32) + (uint32_t)p01 + (uint32_t)p10;
    lo                 = (uint32_t)p00 | (mid << 32);
    hi                 = p11 + (p01 >> 32) + (p10 >> 32) + (mid >> 32);
  }
  static NTT_HD uint64_t barrett(uint64_t hi, uint64_t lo, const consts &c)
  {
    const uint64_t q1 = (hi << (64u - c.bsh)) | (lo >> c.bsh); /* bsh in [39,60]: the policy serves 2^40 <= q < 2^61 */
    return lo - mulhi64(q1, c.bmu) * c.q;
  }
  static constexpr int kDotEvery = K == 3 ? 20 : (K == 1 ? 4 : 1);
  static constexpr int kDotChunk = 2;
  template <bool LAZY> static NTT_HD val dot_term(uint64_t a, uint64_t b, const consts &c)
  {
    if(LAZY && K <= 1) b = csub(b, c.q2);
    if(LAZY && K == 0) a = csub(a, c.q2);
    uint64_t hi, lo;
    mul128(a, b, hi, lo);
    return barrett(hi, lo, c);
  }
  static NTT_HD val dot_acc(val acc, val t, const consts &) { return acc + t; }
  static NTT_HD val dot_fold(val acc, const consts &c) { return reduce_any(acc, c); }
  /* product at the output of a forward transform: x (anything below B q) is brought below 2.01 q first */
  template <bool LAZY> static NTT_HD val mul_out(val x, uint64_t b, const consts &c)
  {
    if(LAZY && K <= 1) b = csub(b, c.q2);
    if(LAZY && K == 0) b = csub(b, c.q);
    uint64_t hi, lo;
    mul128(reduce_any(x, c), b, hi, lo);
    return barrett(hi, lo, c);
  }
  static NTT_HD uint64_t mul_store(val r, const consts &c) { return csub(csub(r, c.q2), c.q); }
  static NTT_HD uint64_t mul_store_acc(val r, uint64_t acc, const consts &c) { return csub(csub(r + acc, c.q2), c.q); }
};

/* ------------------------------------------------------------------ */
/* ArithF64                                                            */
/* ------------------------------------------------------------------ */
/*
 * Bound bookkeeping.  Values are integer-valued doubles v with |v| <= B*q.
 * theta = q / 2^52.  With balanced twiddles (|w| <= q/2) a product satisfies
 *   |r| <= (1/2 + B_y * theta / 2 + eps) * q                       
 * and stays exact while B*q < 2^53.  KSH = number of bits of headroom class:
 * the policy is instantiated for q <= 2^(51-KSH) * (1 + 2^-10).
 */
struct F64Consts {
  double q, qinv;  /* q and fl(1/q)                                      */
  double qinv_lo;  /* 1/q - fl(1/q): second word of the reciprocal       */
  double half_q;   /* q/2 rounded down, for balanced input conversion    */
  TwF64  ninv;     /* N^-1 (balanced) and ninv/q                         */
  TwF64  wninv;    /* N^-1 * winv[1] (balanced) and its /q               */
  uint64_t qi;     /* q as integer                                       */
  double   q2_sub; /* the double whose BIT PATTERN is the integer 2q (2q * 2^-1074): lazy outputs */
};

struct ArithF64 {
  using val    = double;
  using tw     = TwF64;
  using consts = F64Consts;
  static constexpr bool kTracksBounds = true;
  /* compact 8-byte twiddle: only the balanced multiplier is stored and w/q is
   * rebuilt with a two-word reciprocal, fma(w, qinv, w*qinv_lo): the result is
   * within 2^-53(1/2 + 2^-10) relative of w/q, i.e. as good as the stored
   * quotient of a full record, so the same bounds apply. */
  static constexpr bool kCompact = true;
  static constexpr bool kRadix4  = false;
  static constexpr bool kWide52  = false; /* WideF64: the kernels' MASK means "reduce the multiplied operand" */
  static constexpr bool kIntWide = false;
  using ctw                      = double;
  static NTT_HD tw expand(ctw w, const consts &c) { return tw{w, fma_(w, c.qinv, w * c.qinv_lo)}; }

  static NTT_HD double magic52() { return 4503599627370496.0; } /* 2^52 */

  /* u64 in [0,2^53) -> double, exact, in ONE instruction: the bit pattern of
   * u < 2^52 read as a double is the subnormal u * 2^-1074, and ldexp by 1074
   * (v_ldexp_f64, exact) scales it back to the integer u; for 2^52 <= u < 2^53 the pattern is a number of the first
   * normal binade, (1 + (u - 2^52) 2^-52) 2^-1022 = u * 2^-1074 just the same (lazy words up to 4q arrive here).
   * FP64 subnormals are never flushed on gfx9 (and not on the CPU emulation either).  (The name keeps its history:
   * canonical inputs are below 2^52.) */
  static NTT_HD double u64_to_f64_lt52(uint64_t u)
  {
    union {
      uint64_t u;
      double   d;
    } x;
    x.u = u;
    return __builtin_ldexp(x.d, 1074);
  }
  /* v - q*rint(v/q): |result| <= q/2 (+1 ulp of the quotient) */
  static NTT_HD double reduce(double v, const consts &c)
  {
    const double k = rint_(v * c.qinv);
    return fma_(-k, c.q, v);
  }
  static NTT_HD double mulmod(const tw &t, double y, const consts &c)
  {
    const double h = y * t.w;
    const double k = rint_(y * t.wq);
    const double l = fma_(y, t.w, -h);
    const double d = fma_(-k, c.q, h);
    return d + l;
  }
  /* product by a compact (8-byte) twiddle: the quotient is estimated from the
   * rounded product itself, k = rint(fl(y*w) * fl(1/q)), so no w/q has to be
   * stored or rebuilt.  Three roundings instead of two: the estimate is within
   * 1.5 * 2^-52 relative of y*w/q (2^-52 for a stored quotient), which the
   * reduction schedule accounts for (f64_schedule, cmask).  h - k*q is still an
   * integer below 2^53 in magnitude, hence exact. */
  static NTT_HD double mulmod_c(ctw w, double y, const consts &c)
  {
    const double h = y * w;
    const double k = rint_(h * c.qinv);
    const double l = fma_(y, w, -h);
    const double d = fma_(-k, c.q, h);
    return d + l;
  }
  /* the same with the quotient estimated through the two-word reciprocal: k = rint(fma(h, qinv, h * qinv_lo)).  One
   * instruction more than mulmod_c; the estimate then carries only the rounding of h and one final rounding -- within
   * 2^-52 relative of y*w/q, as good as a stored quotient (mulmod), so the bounds of a FULL record apply.  Used where the
   * 1.5x looser estimate of mulmod_c would force an extra reduction (WideF64's forward butterflies). */
  static NTT_HD double mulmod_c2(ctw w, double y, const consts &c)
  {
    const double h = y * w;
    const double k = rint_(fma_(h, c.qinv, h * c.qinv_lo));
    const double l = fma_(y, w, -h);
    const double d = fma_(-k, c.q, h);
    return d + l;
  }
  /* strict API: raw in [0,q).  WIDE (reference-signature shims): raw may be
   * anywhere in [0,8q) -- folded with integer conditional subtracts first. */
  template <bool INV, bool WIDE> static NTT_HD val load(uint64_t raw, const consts &c)
  {
    if(WIDE) {
      raw = raw < 4 * c.qi ? raw : raw - 4 * c.qi;
      raw = raw < 2 * c.qi ? raw : raw - 2 * c.qi;
      raw = raw < c.qi ? raw : raw - c.qi;
    }
    return u64_to_f64_lt52(raw);
  }
  template <bool RED> static NTT_HD void fwd_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    const double xr = RED ? reduce(x, c) : x;
    const double m  = mulmod(t, y, c);
    x               = xr + m;
    y               = xr - m;
  }
  template <bool RED> static NTT_HD void inv_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    const double s = x + y;
    const double d = x - y;
    x              = RED ? reduce(s, c) : s;
    y              = mulmod(t, d, c);
  }
  /* the same butterflies on a compact twiddle */
  template <bool RED> static NTT_HD void fwd_bfly(val &x, val &y, ctw w, const consts &c)
  {
    const double xr = RED ? reduce(x, c) : x;
    const double m  = mulmod_c(w, y, c);
    x               = xr + m;
    y               = xr - m;
  }
  template <bool RED> static NTT_HD void inv_bfly(val &x, val &y, ctw w, const consts &c)
  {
    const double s = x + y;
    const double d = x - y;
    x              = RED ? reduce(s, c) : s;
    y              = mulmod_c(w, d, c);
  }
  static NTT_HD void inv_bfly_last(val &x, val &y, const consts &c)
  {
    const double s = x + y;
    const double d = x - y;
    x              = mulmod(c.ninv, s, c);
    y              = mulmod(c.wninv, d, c);
  }
  /* Gentleman-Sande butterfly with the twiddle given as -w^-1 (the forward table read in mirrored order,
   * ntt_core.h load_stage_tw MIRROR): (x - y) * w^-1 = (y - x) * (-w^-1) */
  template <bool RED> static NTT_HD void inv_bfly_mirror(val &x, val &y, ctw wneg, const consts &c)
  {
    const double s = x + y;
    const double d = y - x;
    x              = RED ? reduce(s, c) : s;
    y              = mulmod_c(wneg, d, c);
  }
  /* Product in the NTT domain inside a kernel (fused product: forward transform of b -> times a^ -> inverse,
   * without leaving the registers).  x: a forward output in balanced form, |x| <= B*q with B*q < 2^53;
   * a: the other operand's transform as a stored word -- canonical [0,q) or, LAZY, a lazy output of this
   * library ([0,4q), below 2^53).  Both factors are reduced to |.| <= q/2 first (3 instructions each, exact), so
   * the product is at most q^2/4, the quotient estimate from the rounded product is within 0.2 of the truth,
   * h - k*q is an integer below q and the result satisfies |r| <= 0.7 q: a valid input bound for the inverse
   * transform's reduction plan (which assumes 1). */
  template <bool LAZY> static NTT_HD val product_in_domain(val x, uint64_t a, const consts &c)
  {
    const double xr = reduce(x, c);
    const double y0 = u64_to_f64_lt52(a);
    const double y  = reduce(LAZY ? y0 - (c.q + c.q) : y0, c);
    return mulmod_c(y, xr, c);
  }
  /* the same with BOTH factors still in registers (two forward transforms of one work item): each is reduced to |.| <= q/2
   * first, so the bound above holds unchanged */
  static NTT_HD val product_rr(val x, val y, const consts &c) { return mulmod_c(reduce(y, c), reduce(x, c), c); }
  /* balanced |v| < 2^53 -> canonical [0,q) as u64 */
  static NTT_HD uint64_t to_canonical(double v, const consts &c)
  {
    const double r = reduce(v, c); /* |r| <= q/2 (+tiny) */
    /* (r + (r < 0 ? q : 0)) * 2^-1074 in one exact fma: the result is a
     * subnormal whose bit pattern IS the canonical integer (the inverse of
     * u64_to_f64_lt52).  The addend's bit pattern is q or 0 for the same reason.
     * r >= 0: u = r <= q/2 + slack < q;  r < 0: u = r + q in [q/2 - slack, q):
     * always canonical, no further fold needed */
    union {
      double   d;
      uint64_t u;
    } k, x, s;
    /* q where r is negative, 0 elsewhere, from the sign bit by integer operations (an arithmetic shift and two ANDs
     * with the scalar q: full-rate instructions, where a compare and two selects cost an FP64-rate compare more).
     * r is never -0.0: an fma whose exact result is zero returns +0 in round-to-nearest. */
    s.d = r;
    k.u = c.qi & (uint64_t)((int64_t)s.u >> 63);
    x.d = fma_(r, 0x1p-1074, k.d);
    return x.u;
  }
  static NTT_HD uint64_t store_fwd(val v, const consts &c) { return to_canonical(v, c); }
  static NTT_HD uint64_t store_inv(val v, const consts &c) { return to_canonical(v, c); }
  /* lazy forward output in [0,4q) (the reference's radix-2 lazy range, include/ntt_reference.h:13-17):
   * v + 2q for |v| <= kLazyBound * q -- ONE exact fma instead of the seven instructions of to_canonical.  (v + 2q) is
   * an integer below (2 + kLazyBound) q < 2^53 for every modulus of the policy (q <= 2^51(1+2^-10): 3.99 q < 2^53; note
   * 4q itself may exceed 2^53 just above 2^51), and an integer u < 2^53 scaled by 2^-1074 is the double whose bit
   * pattern is u (subnormal below 2^52, first binade above).  The kernel's reduction schedule
   * guarantees the bound (fused_mask with LAZY). */
  static NTT_HD uint64_t store_fwd_lazy(val v, const consts &c)
  {
    union {
      double   d;
      uint64_t u;
    } x;
    x.d = fma_(v, 0x1p-1074, c.q2_sub);
    return x.u;
  }
  /* the inverse's last group leaves sums of scaled values (|v| up to ~3q): no cheap lazy form exists
   * below 2^53, so the "lazy" inverse output of this policy is the canonical one (a legal member of
   * [0,2q)) */
  static NTT_HD uint64_t store_inv_lazy(val v, const consts &c) { return to_canonical(v, c); }
  static NTT_HD val      scale_ninv(val v, const consts &c) { return mulmod(c.ninv, v, c); }
  /* pointwise product of two values in [0,q): b is re-centred so |a*b/q| <= q/2
   * and the on-the-fly quotient h*qinv is within 0.4 of the truth */
  static NTT_HD uint64_t mulmod_full(uint64_t a, uint64_t b, const consts &c)
  {
    const double x  = u64_to_f64_lt52(a);
    double       y  = u64_to_f64_lt52(b);
    y               = y > c.half_q ? y - c.q : y;
    const double h  = x * y;
    const double k  = rint_(h * c.qinv);
    const double l  = fma_(x, y, -h);
    const double d  = fma_(-k, c.q, h);
    return to_canonical(d + l, c);
  }
  /* the same for LAZY operands anywhere in [0,4q): one integer fold brings them below 2q (4q itself may exceed
   * 2^53 for q just above 2^51, so the fold comes before the conversion), both are re-centred by q, b is then
   * reduced to |y| <= q/2 so that the quotient estimate from the rounded product stays within 1 of the truth
   * and h - k*q below 2^53 */
  static NTT_HD uint64_t mulmod_full_lazy4(uint64_t a, uint64_t b, const consts &c)
  {
    a                  = a < 2 * c.qi ? a : a - 2 * c.qi;
    b                  = b < 2 * c.qi ? b : b - 2 * c.qi;
    const double x     = u64_to_f64_lt52(a) - c.q;
    const double y     = reduce(u64_to_f64_lt52(b) - c.q, c);
    const double h     = x * y;
    const double k     = rint_(h * c.qinv);
    const double l     = fma_(x, y, -h);
    const double d     = fma_(-k, c.q, h);
    return to_canonical(d + l, c);
  }
  /* Inner product in the NTT domain: c = inv(sum_i a_i^ (.) b_i^) with the products formed where the inverse transform
   * would convert its input words (dot_inv_kernel).  One term from two STORED words -- canonical [0,q) or, LAZY, anywhere
   * in [0,4q) (folded below 2q with integer operations first: 4q may exceed 2^53 just above 2^51) --: only b is reduced
   * to |y| <= q/2 (three exact instructions); with |x| < q the quotient estimate from the rounded product is within
   * 1/2 + 1.5 |x y / q| 2^-52 <= 1/2 + 1.5 theta2 of the truth (theta2 = q / 2^53 <= 0.2503), so |r| <= 0.8755 q, h - k q is
   * an integer below 1.01 q < 2^53 and the result is exact (the argument of DESIGN.md 4.1 / 4.8).  A single term is a valid
   * input of the inverse reduction plan (which assumes 1); a running sum is folded back to |.| <= q/2 every kDotEvery
   * terms, 1/2 + 3 * 0.8755 = 3.13 staying below every class's exactness limit (3.87 q for q ~ 2^51), and once at the end. */
  static constexpr int kDotEvery = 3;
  static constexpr int kDotChunk = 4; /* products the kernel lets the scheduler interleave (register budget) */
  template <bool LAZY> static NTT_HD val dot_term(uint64_t a, uint64_t b, const consts &c)
  {
    if(LAZY) {
      a = a < 2 * c.qi ? a : a - 2 * c.qi;
      b = b < 2 * c.qi ? b : b - 2 * c.qi;
    }
    const double x0 = u64_to_f64_lt52(a), y0 = u64_to_f64_lt52(b);
    const double x  = LAZY ? x0 - c.q : x0;
    const double y  = reduce(LAZY ? y0 - c.q : y0, c);
    return mulmod_c(y, x, c);
  }
  static NTT_HD val dot_acc(val acc, val t, const consts &) { return acc + t; }
  static NTT_HD val dot_fold(val acc, const consts &c) { return reduce(acc, c); }
  /* Product at the OUTPUT of a forward transform: c^ = fwd(a) (.) b^ (+ c^), the result staying in the NTT domain
   * (fwd_mul_kernel).  x: the last stage's value (|x| <= B q, B q < 2^53), b: a stored word -- canonical or, LAZY, anywhere in
   * [0,4q) (folded below 2q with integer operations, re-centred by q).  Both factors are reduced to |.| <= q/2, so the
   * bounds of product_in_domain hold (|r| <= 0.7 q; 0.875 q for moduli up to 2^52: theta2 < 1/2).  With an accumulator word
   * (canonical, < q) the sum is below 1.875 q < 2^53: exact, and to_canonical reduces it. */
  template <bool LAZY> static NTT_HD val mul_out(val x, uint64_t b, const consts &c)
  {
    if(LAZY) b = b < 2 * c.qi ? b : b - 2 * c.qi;
    const double y0 = u64_to_f64_lt52(b);
    const double y  = reduce(LAZY ? y0 - c.q : y0, c);
    return mulmod_c(y, reduce(x, c), c);
  }
  static NTT_HD uint64_t mul_store(val r, const consts &c) { return to_canonical(r, c); }
  static NTT_HD uint64_t mul_store_acc(val r, uint64_t acc, const consts &c) { return to_canonical(r + u64_to_f64_lt52(acc), c); }
};

/* ------------------------------------------------------------------ */
/* ArithF64W: FP64 arithmetic for moduli up to 2^52                      */
/* ------------------------------------------------------------------ */
/*
 * Above 2^51(1+2^-10) the balanced-double policy has less than two bits of headroom below 2^53: no value may
 * reach 2q.  The pass-through operand x of every butterfly is reduced to |.| <= q/2 (three exact instructions); the
 * MULTIPLIED operand y is reduced where a compile-time schedule says so (forward; round 4) or always (inverse):
 *   forward  x~ = red(x), m = y * w mod q with y reduced or not.  With |y| <= B q and theta2 = q / 2^53 < 1/2:
 *            full record (stored w/q), or a compact twiddle with the two-word reciprocal (mulmod_c2):
 *                                        |m| <= (1/2 +     B theta2) q,  |h - k q| <= (1/2 + 1.5 B theta2) q
 *            (a compact twiddle with mulmod_c's one-word estimate: 1/2 + 1.5 B theta2 and 1/2 + 2 B theta2 -- it would have to
 *            reduce y in EVERY stage, which is why this policy pays one instruction more per compact product instead)
 *            (|l| = |h - y w| <= ulp(h)/2 <= B theta2 q / 2), outputs |x~ +- m| <= (1/2 + eps) q + |m|.  Everything must stay
 *            below 2^53 = q / theta2, i.e. below 2 q in the worst case: f64w_fwd_schedule (below) follows B stage by stage
 *            and reduces y only where one of the three bounds would pass 2 (1 - 2^-6): one stage in five
 *            (B: 1 -> 1.5 -> 1.75 -> 1.875 -> 1.94 -> reduce).  At N = 2^14: stages 4 and 9 of 14 -- 11.9 instead of 14 instructions
 *            per butterfly on average (11 with a full record, 12 with a compact twiddle, +3 where y is reduced).
 *   inverse  s = x + y, d = x - y of values bounded by 1.38 q each (canonical inputs: < 2q): exact below 2^53 because
 *            the pairs a stage combines are either both reduced sums (<= q/2) or both products (<= 0.88 q) or both
 *            canonical inputs; s~ = red(s), y' = red(d) * w mod q.  (An unreduced d would leave products up to 1.25 q, and the
 *            next stage's difference of two of them is not below 2q: no schedule exists here.)
 * The emulator's checked policy runs the same butterflies (tests/test_emu.py: every value an integer below 2^53, every
 * product exact, for the largest 52-bit primes and adversarial inputs).  Lazy outputs do not exist below 2^53 (4q > 2^53)
 * and fall back to canonical ones.  Written as a mixin so that the checked policy goes through the very same code.
 */
template <class Base> struct WideF64 : Base {
  using val    = typename Base::val;
  using tw     = typename Base::tw;
  using ctw    = typename Base::ctw;
  using consts = typename Base::consts;
  static constexpr bool kWide52 = true;

  /* RED: reduce the multiplied operand too (f64w_fwd_schedule) */
  template <bool RED> static NTT_HD void fwd_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    const double xr = Base::reduce(x, c);
    const double m  = Base::mulmod(t, RED ? Base::reduce(y, c) : y, c);
    x               = xr + m;
    y               = xr - m;
  }
  /* compact twiddle: the quotient through the two-word reciprocal (mulmod_c2: +1 instruction), so that the stage obeys a full
   * record's bounds and the schedule need not reduce y in every compact stage (-3 instructions in four stages of five) */
  template <bool RED> static NTT_HD void fwd_bfly(val &x, val &y, ctw w, const consts &c)
  {
    const double xr = Base::reduce(x, c);
    const double m  = Base::mulmod_c2(w, RED ? Base::reduce(y, c) : y, c);
    x               = xr + m;
    y               = xr - m;
  }
  /* RED (round 5): reduce the difference before the product.  RED = false is chosen (ntt_core.h bfly_reduces) where both inputs
   * are REDUCED SUMS of the stage before, |x|, |y| <= q/2 + 2: then |d| <= q + 4, the true quotient Q = d w / q is at most
   * q/2 + 2 in magnitude, its estimate -- a stored w/q, or a compact twiddle through the two-word reciprocal (mulmod_c2) --
   * is within 2 |Q| 2^-53 < theta2 (1 + 2^-50) of it, so |y'| <= (1/2 + theta2) q + 2 <= q - N + 3 for an NTT-friendly
   * q <= 2^52 - 2N + 1 (theta2 = q / 2^53 <= 1/2 - N / 2^52); h - k q is an integer below 1.13 q: exact.  Two such products
   * add up to at most 2q - 2N + 6 < 2^53, the next stage's sum and difference are exact, and that stage reduces both.
   * 11 instead of 14 instructions (12 with a compact twiddle: mulmod_c's one-word estimate would be 1.5 times looser and
   * leave products up to 1.25 q).  The checked policy of tests/emu runs these very functions (test_wide_fp64_policy_52_bit_moduli). */
  template <bool RED> static NTT_HD void inv_bfly(val &x, val &y, const tw &t, const consts &c)
  {
    const double s = x + y;
    const double d = x - y;
    x              = Base::reduce(s, c);
    y              = Base::mulmod(t, RED ? Base::reduce(d, c) : d, c);
  }
  template <bool RED> static NTT_HD void inv_bfly(val &x, val &y, ctw w, const consts &c)
  {
    const double s = x + y;
    const double d = x - y;
    x              = Base::reduce(s, c);
    y              = RED ? Base::mulmod_c(w, Base::reduce(d, c), c) : Base::mulmod_c2(w, d, c);
  }
  template <bool RED> static NTT_HD void inv_bfly_mirror(val &x, val &y, ctw wneg, const consts &c)
  {
    const double s = x + y;
    const double d = y - x;
    x              = Base::reduce(s, c);
    y              = RED ? Base::mulmod_c(wneg, Base::reduce(d, c), c) : Base::mulmod_c2(wneg, d, c);
  }
  /* the block kernels' form (ntt_core.h w52_inv_plan): REDD as RED above; REDS = false leaves the sum of two REDUCED inputs
   * unreduced (|s| <= q + 4: the next stage adds or subtracts two such values, at most 2q + 8 < 2^53, and reduces both) */
  template <bool REDD, bool REDS> static NTT_HD void inv_bfly2(val &x, val &y, const tw &t, const consts &c)
  {
    const double s = x + y;
    const double d = x - y;
    x              = REDS ? Base::reduce(s, c) : s;
    y              = Base::mulmod(t, REDD ? Base::reduce(d, c) : d, c);
  }
  template <bool REDD, bool REDS> static NTT_HD void inv_bfly2(val &x, val &y, ctw w, const consts &c)
  {
    const double s = x + y;
    const double d = x - y;
    x              = REDS ? Base::reduce(s, c) : s;
    y              = REDD ? Base::mulmod_c(w, Base::reduce(d, c), c) : Base::mulmod_c2(w, d, c);
  }
  static NTT_HD void inv_bfly_last(val &x, val &y, const consts &c)
  {
    const double s = Base::reduce(x + y, c);
    const double d = Base::reduce(x - y, c);
    x              = Base::mulmod(c.ninv, s, c);
    y              = Base::mulmod(c.wninv, d, c);
  }
  /* 4q exceeds 2^53: no lazy form; reduced outputs satisfy the lazy contract */
  static NTT_HD uint64_t store_fwd_lazy(val v, const consts &c) { return Base::store_fwd(v, c); }
  /* inner-product terms (see ArithF64::dot_term): BOTH factors reduced to |.| <= q/2, so the quotient estimate is within
   * 1/2 + 1.5 (q/4) 2^-52 = 1/2 + 0.75 theta2 <= 0.875 of the truth for theta2 = q / 2^53 up to 1/2: |r| <= 0.875 q, a valid
   * operand of this policy's inverse butterflies (pairs of products: 1.75 q < 2^53); the running sum is folded after
   * every term: 1/2 + 0.875 < 2 */
  static constexpr int kDotEvery = 1;
  static constexpr int kDotChunk = 2;
  template <bool LAZY> static NTT_HD val dot_term(uint64_t a, uint64_t b, const consts &c)
  {
    if(LAZY) {
      a = a < 2 * c.qi ? a : a - 2 * c.qi;
      b = b < 2 * c.qi ? b : b - 2 * c.qi;
    }
    const double x0 = Base::u64_to_f64_lt52(a), y0 = Base::u64_to_f64_lt52(b);
    const double x  = Base::reduce(LAZY ? x0 - c.q : x0, c);
    const double y  = Base::reduce(LAZY ? y0 - c.q : y0, c);
    return Base::mulmod_c(y, x, c);
  }
};
using ArithF64W = WideF64<ArithF64>;

/* ------------------------------------------------------------------ */
/* compile-time reduction schedule for ArithF64                        */
/* ------------------------------------------------------------------ */
/*
 * Returns a bit mask over local stages 0..nstages-1: bit s set => reduce the
 * non-multiplied operand in that stage.  Model, in units of q,
 * theta2 = theta/2 = q/2^53 upper bound for the class:
 *   forward  no-reduce: B' = B + rho(B),      reduce: B' = 1/2 + e + rho(B)
 *            rho(B) = 1/2 + B*theta2*(1+e) + e
 *   inverse  s,d bounded by 2B;  no-reduce: B' = max(2B, rho(2B)),
 *            reduce: B' = max(1/2+e, rho(2B))
 * A stage must reduce when the no-reduce bound (or, inverse, 2B itself)
 * would exceed LIM = (2^53/q)*(1-2^-6).
 */
struct F64Sched {
  uint32_t mask;
  double   bout;
};

constexpr double f64_rho(double b, double theta2) { return 0.5 + b * theta2 * 1.001 + 0.001; }

/* cmask: bit s set => the stage processed at position s multiplies by a compact
 * (8-byte) twiddle, whose quotient estimate (mulmod_c) is 1.5x less accurate */
constexpr F64Sched f64_schedule_forced(bool inverse, int nstages, int ksh, double b_in, uint32_t cmask, uint32_t force);

/* bout_max: largest |value|/q the last stage may leave (forward only; 2 - slack for lazy outputs): stages are
 * forced to reduce from the last one backwards until the bound holds */
constexpr F64Sched f64_schedule(bool inverse, int nstages, int ksh, double b_in, uint32_t cmask = 0, double bout_max = 1e30)
{
  uint32_t force = 0;
  F64Sched sc    = f64_schedule_forced(inverse, nstages, ksh, b_in, cmask, force);
  for(int s = nstages - 1; s >= 0 && sc.bout > bout_max; s--) {
    force |= 1u << s;
    sc = f64_schedule_forced(inverse, nstages, ksh, b_in, cmask, force);
  }
  return sc;
}

constexpr F64Sched f64_schedule_forced(bool inverse, int nstages, int ksh, double b_in, uint32_t cmask, uint32_t force)
{
  /* class ksh: q <= 2^(51-ksh)*(1+2^-10) */
  double theta2 = 0.25 * 1.001; /* q/2^53 for ksh=0 */
  double lim    = 4.0 / 1.001;  /* 2^53/q           */
  for(int i = 0; i < ksh; i++) {
    theta2 *= 0.5;
    lim *= 2.0;
  }
  lim *= (1.0 - 1.0 / 64.0);
  double   b    = b_in;
  uint32_t mask = 0;
  const double theta2_full = theta2;
  for(int s = 0; s < nstages; s++) {
    theta2 = ((cmask >> s) & 1u) ? 1.5 * theta2_full : theta2_full;
    if(!inverse) {
      const double nr = b + f64_rho(b, theta2);
      /* also keep the *next* stage feasible: after a no-reduce stage the next
       * one can always reduce, so only the immediate bound matters */
      if(nr > lim || ((force >> s) & 1u)) {
        mask |= (1u << s);
        b = 0.501 + f64_rho(b, theta2);
      } else {
        b = nr;
      }
    } else {
      const double two_b = 2.0 * b;
      const double r     = f64_rho(two_b, theta2);
      /* s = x+y must itself stay exact: two_b <= lim is a precondition kept
       * by always reducing when the unreduced sum would break it next time */
      const double nr = two_b > r ? two_b : r;
      if(2.0 * nr > lim) {
        mask |= (1u << s);
        b = r > 0.501 ? r : 0.501;
      } else {
        b = nr;
      }
    }
  }
  return F64Sched{mask, b};
}

/* ------------------------------------------------------------------ */
/* compile-time schedule for WideF64's forward butterflies              */
/* ------------------------------------------------------------------ */
/*
 * Bit s set => the butterflies of the stage at processing position s reduce their MULTIPLIED operand (the pass-through
 * operand is always reduced).  Model in units of q at the class's worst case theta2 = q / 2^53 = 1/2 (q -> 2^52), with the
 * slack conventions of f64_schedule; c = 1 for a full twiddle record, 1.5 for a compact one (cmask bit):
 *   keep y (|y| <= B):  product  rho = 1/2 + c B theta2;  exactness  E = rho + B theta2 / 2;  outputs  B' = 1/2 + rho
 *   all of them must stay below LIM = 2 (1 - 2^-6); otherwise y is reduced first (B -> 1/2).
 */
constexpr F64Sched f64w_fwd_schedule(int nstages, double b_in, uint32_t cmask)
{
  const double theta2 = 0.5;
  const double lim    = 2.0 * (1.0 - 1.0 / 64.0);
  double       b      = b_in;
  uint32_t     mask   = 0;
  for(int s = 0; s < nstages; s++) {
    const double cc   = ((cmask >> s) & 1u) ? 1.5 : 1.0;
    const double rho  = 0.5 + cc * b * theta2 * 1.001 + 0.001;
    const double ex   = rho + 0.5 * b * theta2 * 1.001;
    const double outb = 0.501 + rho;
    if(ex > lim || outb > lim || b > lim) {
      mask |= 1u << s;
      b = 0.501 + (0.5 + cc * 0.501 * theta2 * 1.001 + 0.001);
    } else {
      b = outb;
    }
  }
  return F64Sched{mask, b};
}

} /* namespace ntt */
