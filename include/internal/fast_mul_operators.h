// SPDX-License-Identifier: Apache-2.0
// Interface-compatible with IBM/optimized-number-theoretic-transform-implementations (Copyright IBM Inc., Apache-2.0):
// function names, signatures and arithmetic follow include/internal/fast_mul_operators.h of that repository so that its unchanged test and benchmark sources compile against
// this directory.  Re-written for this library (bodies, ordering and comments are new); see NOTICE.
/*
 * internal/fast_mul_operators.h -- host-side scalar primitives of the boundary.
 *
 * Provides mul_op_t and the inline reducers / lazy products / butterflies under
 * the reference's names (reference include/internal/fast_mul_operators.h:10-149)
 * because the reference's own headers-inline wrappers and its tests use them
 * (tests/test_cases.h:62-67 stores n_inv as a mul_op_t and passes it BY VALUE to
 * inv_ntt_ref_harvey / inv_ntt_radix4).  mul_op_t must therefore keep the
 * reference ABI: two __uint128_t fields, sizeof 32, alignment 16.
 *
 * The GPU kernels do not use this file; their arithmetic lives in
 * optimized-number-theoretic-transform-implementations_amd/csrc/ntt_arith.h.
 */
#ifndef NTT_MI355X_FAST_MUL_OPERATORS_H
#define NTT_MI355X_FAST_MUL_OPERATORS_H

#include "defs.h"

EXTERNC_BEGIN

typedef struct mul_op_s {
  __uint128_t op;  /* multiplier w                 */
  __uint128_t con; /* floor(w * 2^word_size / q)   */
} mul_op_t;

/* one conditional subtract: values below `bound` pass through */
static inline uint64_t ntt_cond_sub(const uint64_t v, const uint64_t bound)
{
  return v - ((v >= bound) ? bound : 0);
}

/* [0,2q)->[0,q), [0,4q)->[0,2q), ... (reference :15-43) */
static inline uint64_t reduce_2q_to_q(const uint64_t val, const uint64_t q) { return ntt_cond_sub(val, q); }
static inline uint64_t reduce_4q_to_2q(const uint64_t val, const uint64_t q) { return ntt_cond_sub(val, q << 1); }
static inline uint64_t reduce_8q_to_4q(const uint64_t val, const uint64_t q) { return ntt_cond_sub(val, q << 2); }
static inline uint64_t reduce_4q_to_q(const uint64_t val, const uint64_t q)
{
  return ntt_cond_sub(ntt_cond_sub(val, q << 1), q);
}
static inline uint64_t reduce_8q_to_2q(const uint64_t val, const uint64_t q)
{
  return ntt_cond_sub(ntt_cond_sub(val, q << 2), q << 1);
}
static inline uint64_t reduce_8q_to_q(const uint64_t val, const uint64_t q)
{
  return ntt_cond_sub(reduce_8q_to_2q(val, q), q);
}

#ifndef L_HIGH_WORD
#  define L_HIGH_WORD HIGH_WORD
#endif

/* Shoup/Harvey lazy product, result in [0,2q) (reference :49-54) */
static inline uint64_t fast_mul_mod_q2(const mul_op_t w, const uint64_t t, const uint64_t q)
{
  const uint64_t quotient = (uint64_t)L_HIGH_WORD(w.con * t);
  return (uint64_t)w.op * t - quotient * q;
}

static inline uint64_t fast_mul_mod_q(const mul_op_t w, const uint64_t t, const uint64_t q)
{
  return reduce_2q_to_q(fast_mul_mod_q2(w, t, q), q);
}

/* w1*t1 + w2*t2 with a single quotient estimate (reference :62-70) */
static inline uint64_t fast_dbl_mul_mod_q2(const mul_op_t w1, const mul_op_t w2, const uint64_t t1,
                                           const uint64_t t2, const uint64_t q)
{
  const uint64_t quotient = (uint64_t)L_HIGH_WORD(w1.con * t1 + w2.con * t2);
  return (uint64_t)w1.op * t1 + (uint64_t)w2.op * t2 - quotient * q;
}

/* Cooley-Tukey butterfly on [0,4q) values (reference :72-81) */
static inline void harvey_fwd_butterfly(uint64_t *X, uint64_t *Y, const mul_op_t w, const uint64_t q)
{
  const uint64_t x = reduce_4q_to_2q(*X, q);
  const uint64_t m = fast_mul_mod_q2(w, *Y, q);
  *X               = x + m;
  *Y               = x + (q << 1) - m;
}

/* Gentleman-Sande butterfly on [0,2q) values (reference :83-92) */
static inline void harvey_bkw_butterfly(uint64_t *X, uint64_t *Y, const mul_op_t w, const uint64_t q)
{
  const uint64_t sum  = *X + *Y;
  const uint64_t diff = *X + (q << 1) - *Y;
  *X                  = reduce_4q_to_2q(sum, q);
  *Y                  = fast_mul_mod_q2(w, diff, q);
}

/* last inverse stage with N^-1 folded into both outputs (reference :94-106) */
static inline void harvey_bkw_butterfly_final(uint64_t *X, uint64_t *Y, const mul_op_t w,
                                              const mul_op_t n_inv, const uint64_t q)
{
  const uint64_t sum  = *X + *Y;
  const uint64_t diff = *X + (q << 1) - *Y;
  *X                  = fast_mul_mod_q(n_inv, sum, q);
  *Y                  = fast_mul_mod_q(w, diff, q);
}

/* two fused forward levels; w = {W1, W2, W1W2, W3, -W1W3} (reference :108-128) */
static inline void radix4_fwd_butterfly(uint64_t *X, uint64_t *Y, uint64_t *Z, uint64_t *T,
                                        const mul_op_t w[5], const uint64_t q)
{
  const uint64_t even = fast_dbl_mul_mod_q2(w[1], w[2], *Y, *T, q);
  const uint64_t odd  = fast_dbl_mul_mod_q2(w[3], w[4], *Y, *T, q);
  const uint64_t x    = reduce_8q_to_4q(*X, q);
  const uint64_t z    = fast_mul_mod_q2(w[0], *Z, q);
  const uint64_t up = x + z, down = x - z;
  *X = up + even;
  *Y = up - even + (q << 1);
  *Z = down + odd + (q << 1);
  *T = down - odd + (q << 2);
}

/* two fused inverse levels (reference :130-149) */
static inline void radix4_inv_butterfly(uint64_t *X, uint64_t *Y, uint64_t *Z, uint64_t *T,
                                        const mul_op_t w[5], const uint64_t q)
{
  const uint64_t q4  = q << 2;
  const uint64_t sxy = *X + *Y, szt = *Z + *T;
  const uint64_t dxy = q4 + *X - *Y, dzt = q4 + *Z - *T;
  *X = reduce_8q_to_2q(sxy + szt, q);
  *Z = fast_mul_mod_q(w[0], q4 + sxy - szt, q);
  *Y = fast_dbl_mul_mod_q2(w[1], w[3], dxy, dzt, q);
  *T = fast_dbl_mul_mod_q2(w[2], w[4], dxy, dzt, q);
}

EXTERNC_END
#endif /* NTT_MI355X_FAST_MUL_OPERATORS_H */
