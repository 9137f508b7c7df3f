/*
 * ntt_reference.h -- radix-2 Harvey NTT entry points, MI355X implementation.
 *
 * Drop-in replacement for reference include/ntt_reference.h:13-65 /
 * src/ntt_reference.c:11-91: same symbol names, same argument lists (including
 * the by-value mul_op_t), same in-place contract on HOST pointers, same table
 * layouts (radix-2: N-entry bit-reversed power table and its 2^64 precon).
 * Behind the signature each call stages the polynomial to the GPU, runs the
 * gfx950 kernels and copies the result back (a correctness/compatibility path;
 * the throughput path is the batched API in ntt_mi355x.h).
 *
 * Differences a caller can observe: none.  By default the kernels run the
 * reference's own Harvey butterflies on the caller's w AND w_con, so the
 * *_lazy functions return the reference's lazy values in [0,4q) bit for bit and
 * the header-inline wrappers below do what they do in the reference.  (With
 * NTT_COMPAT_ARITH=f64 the FP64 kernels serve the call and the lazy functions
 * return values already reduced to [0,q) -- inside the documented lazy range.)
 * If no HIP device is usable the functions print the error to stderr and
 * abort(): the signatures return void and there is no CPU fallback.
 */
#ifndef NTT_MI355X_NTT_REFERENCE_H
#define NTT_MI355X_NTT_REFERENCE_H

#include "fast_mul_operators.h"

EXTERNC_BEGIN

/* forward, one polynomial; output in the lazy range [0,4q)
 * (replaces reference src/ntt_reference.c:11-31) */
NTT_EXPORT void fwd_ntt_ref_harvey_lazy(uint64_t a[], uint64_t N, uint64_t q, const uint64_t w[],
                                        const uint64_t w_con[]);

/* forward with final reduction to [0,q) (reference include/ntt_reference.h:19-31) */
static inline void fwd_ntt_ref_harvey(uint64_t a[], const uint64_t N, const uint64_t q, const uint64_t w[],
                                      const uint64_t w_con[])
{
  fwd_ntt_ref_harvey_lazy(a, N, q, w, w_con);
  for(size_t i = 0; i < N; i++) {
    a[i] = reduce_4q_to_q(a[i], q);
  }
}

/* inverse: bit-reversed in, natural out, scaled by n_inv, output in [0,q)
 * (replaces reference src/ntt_reference.c:33-66) */
NTT_EXPORT void inv_ntt_ref_harvey(uint64_t a[], uint64_t N, uint64_t q, mul_op_t n_inv, uint64_t word_size,
                                   const uint64_t w[], const uint64_t w_con[]);

/* forward on two polynomials sharing the tables
 * (replaces reference src/ntt_reference.c:71-91) */
NTT_EXPORT void fwd_ntt_ref_harvey_lazy_dbl(uint64_t a1[], uint64_t a2[], uint64_t N, uint64_t q,
                                            const uint64_t w[], const uint64_t w_con[]);

static inline void fwd_ntt_ref_harvey_dbl(uint64_t a1[], uint64_t a2[], const uint64_t N, const uint64_t q,
                                          const uint64_t w[], const uint64_t w_con[])
{
  fwd_ntt_ref_harvey_lazy_dbl(a1, a2, N, q, w, w_con);
  for(size_t i = 0; i < N; i++) {
    a1[i] = reduce_4q_to_q(a1[i], q);
    a2[i] = reduce_4q_to_q(a2[i], q);
  }
}

EXTERNC_END
#endif /* NTT_MI355X_NTT_REFERENCE_H */
