/*
 * ntt_seal.h -- link-compatible stand-ins for the reference's SEAL comparator.
 *
 * The reference vendors a C port of SEAL's radix-2 NTT as a comparator
 * (reference include/ntt_seal.h:10-36, third_party/seal/ntt_seal.c); it is not
 * on the accelerated path, but tests/test_correctness.c:61-79
 * and tests/bench.c call it, so the symbols must exist for those files to link
 * unchanged.  Both use the radix-2 tables and compute the same transform, so
 * they are served by the same GPU radix-2 engine as ntt_reference.h.
 */
#ifndef NTT_MI355X_NTT_SEAL_H
#define NTT_MI355X_NTT_SEAL_H

#include "fast_mul_operators.h"

EXTERNC_BEGIN

NTT_EXPORT void fwd_ntt_seal_lazy(uint64_t a[], uint64_t N, uint64_t q, const uint64_t w[],
                                  const uint64_t w_con[]);

static inline void fwd_ntt_seal(uint64_t a[], const uint64_t N, const uint64_t q, const uint64_t w[],
                                const uint64_t w_con[])
{
  fwd_ntt_seal_lazy(a, N, q, w, w_con);
  for(size_t i = 0; i < N; i++) {
    a[i] = reduce_4q_to_q(a[i], q);
  }
}

NTT_EXPORT void inv_ntt_seal(uint64_t a[], uint64_t N, uint64_t q, uint64_t n_inv, uint64_t n_inv_con,
                             const uint64_t w[], const uint64_t w_con[]);

EXTERNC_END
#endif /* NTT_MI355X_NTT_SEAL_H */
