// SPDX-License-Identifier: Apache-2.0
// Interface-compatible with IBM/optimized-number-theoretic-transform-implementations (Copyright IBM Inc., Apache-2.0):
// function names, signatures and table layouts follow include/internal/pre_compute.h of that repository so that its unchanged test and benchmark sources compile against
// this directory.  Re-written for this library (bodies, ordering and comments are new); see NOTICE.
/*
 * internal/pre_compute.h -- host-side table builders of the boundary.
 *
 * Same names, argument order and table layouts as reference
 * include/internal/pre_compute.h:16-105; reference tests/test_cases.h:212-251
 * calls calc_w, calc_w_inv, calc_w_con, calc_ninv_con and expand_w to build the
 * tables it then hands to the fwd_ntt_ and inv_ntt_ entry points.  (The AVX512/s390x-only
 * builders of reference :107-369 are outside this library's scope.)
 *
 * Layout contract:
 *   w_powers[k]      = w^{bitrev_m(k)} mod q                  k in [0,N)
 *   w_con[k]         = floor(w_powers[k] * 2^word_size / q)
 *   expanded e[2k]   = w[k];  e[4k+1] = w[k]*w[2k];  e[4k+3] = q - w[k]*w[2k+1]
 */
#ifndef NTT_MI355X_PRE_COMPUTE_H
#define NTT_MI355X_PRE_COMPUTE_H

#include <stdlib.h>
#include <string.h>

#include "defs.h"

EXTERNC_BEGIN

/* reverse the low `width` bits of idx */
static inline uint64_t bit_rev_idx(uint64_t idx, uint64_t width)
{
  uint64_t out = 0;
  for(uint64_t b = 0; b < width; b++, idx >>= 1) {
    out = (out << 1) | (idx & 1);
  }
  return out;
}

/* scatter w[] into bit-reversed order */
static inline void bit_rev(uint64_t w_powers[], const uint64_t w[], const uint64_t N, const uint64_t width)
{
  for(size_t src = 0; src < N; src++) {
    w_powers[bit_rev_idx(src, width)] = w[src];
  }
}

/* successive powers of `w`, written straight to their bit-reversed slot */
static inline void calc_w(uint64_t w_powers_rev[], const uint64_t w, const uint64_t N, const uint64_t q,
                          const uint64_t width)
{
  uint64_t power = 1;
  for(size_t e = 0; e < N; e++) {
    w_powers_rev[bit_rev_idx(e, width)] = power;
    power = (uint64_t)(((__uint128_t)power * w) % q);
  }
}

/* identical construction for the inverse root (kept as its own symbol for
 * source compatibility with the reference's callers) */
static inline void calc_w_inv(uint64_t w_inv_rev[], const uint64_t w_inv, const uint64_t N, const uint64_t q,
                              const uint64_t width)
{
  calc_w(w_inv_rev, w_inv, N, q, width);
}

/* Shoup precomputation for a whole table */
static inline void calc_w_con(uint64_t w_con[], const uint64_t w[], const uint64_t N, const uint64_t q,
                              const uint64_t word_size)
{
  for(size_t k = 0; k < N; k++) {
    w_con[k] = (uint64_t)(((__uint128_t)w[k] << word_size) / q);
  }
}

static UNUSED uint64_t calc_ninv_con(const uint64_t Ninv, const uint64_t q, const uint64_t word_size)
{
  return (uint64_t)(((__uint128_t)Ninv << word_size) / q);
}

/* 2N-entry radix-4 table: every radix-2 twiddle at the even slots, the two
 * merged products a radix-4 butterfly needs at the odd slots */
static inline void expand_w(uint64_t w_expanded[], const uint64_t w[], const uint64_t N, const uint64_t q)
{
  for(size_t k = 0; k < N; k++) {
    w_expanded[2 * k] = w[k];
  }
  w_expanded[1] = 0;
  w_expanded[3] = 0;
  for(size_t k = 1; 2 * k + 1 < N; k++) {
    const __uint128_t parent = w[k];
    w_expanded[4 * k + 1]    = (uint64_t)((parent * w[2 * k]) % q);
    w_expanded[4 * k + 3]    = q - (uint64_t)((parent * w[2 * k + 1]) % q);
  }
}

EXTERNC_END
#endif /* NTT_MI355X_PRE_COMPUTE_H */
