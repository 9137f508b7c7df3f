/*
 * ntt_radix4.h -- radix-4 NTT entry points, MI355X implementation.
 *
 * Drop-in replacement for reference include/ntt_radix4.h:10-35 /
 * src/ntt_radix4.c:27-114.  The tables are the reference's 2N-entry EXPANDED
 * tables (include/internal/pre_compute.h:85-105) and their precomputation.  For
 * 2^6 <= N <= 2^18 the device runs the reference's radix-4 butterflies with the
 * shared-quotient double product on exactly these tables (collect_roots'
 * five-twiddle pack = records 2s and 4s..4s+3; csrc/ntt_arith.h ArithU64R4), so
 * the lazy forward output ([0,8q), [0,4q) for odd log2 N) equals the reference's bit
 * for bit -- for N = 2^15 .. 2^17 too (reference cases 14-18: one or two radix-4
 * levels as a strided column pass, then 2^13- or 2^14-point blocks that end with the
 * reference's radix-2 stage when log2 N is odd; csrc/ntt_passplan.h make_passes_r4,
 * csrc/ntt_core.h column_pass_thread_r4; round 3); inv_ntt_radix4 runs the same two
 * passes in the opposite order.  N < 2^6 is served by the radix-2 engine on the even
 * slots (slot 2k = w[k]): values in [0,4q), congruent to the reference's.
 */
#ifndef NTT_MI355X_NTT_RADIX4_H
#define NTT_MI355X_NTT_RADIX4_H

#include "fast_mul_operators.h"

EXTERNC_BEGIN

/* replaces reference src/ntt_radix4.c:27-62 */
NTT_EXPORT void fwd_ntt_radix4_lazy(uint64_t a[], uint64_t N, uint64_t q, const uint64_t w[],
                                    const uint64_t w_con[]);

/* reference include/ntt_radix4.h:16-28 */
static inline void fwd_ntt_radix4(uint64_t a[], const uint64_t N, const uint64_t q, const uint64_t w[],
                                  const uint64_t w_con[])
{
  fwd_ntt_radix4_lazy(a, N, q, w, w_con);
  for(size_t i = 0; i < N; i++) {
    a[i] = reduce_8q_to_q(a[i], q);
  }
}

/* replaces reference src/ntt_radix4.c:64-114 (inputs anywhere in [0,8q)) */
NTT_EXPORT void inv_ntt_radix4(uint64_t a[], uint64_t N, uint64_t q, mul_op_t n_inv, const uint64_t w[],
                               const uint64_t w_con[]);

EXTERNC_END
#endif /* NTT_MI355X_NTT_RADIX4_H */
