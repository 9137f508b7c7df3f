/*
 * ntt_radix4x4.h -- radix-16-blocked forward NTT entry point, MI355X implementation.
 *
 * Drop-in replacement for reference include/ntt_radix4x4.h:10-28 /
 * src/ntt_radix4x4.c:41-114 (same expanded tables as ntt_radix4.h).  On the GPU
 * the "two radix-4 levels per block" idea is the native shape of every kernel:
 * each thread keeps a 16-coefficient tile in registers for four stages
 * (csrc/ntt_core.h), so this symbol shares the radix-4 engine of ntt_radix4.h:
 * its lazy output is the one of fwd_ntt_radix4_lazy -- in [0,8q) and congruent to
 * the reference's radix-4x4 output (which pairs the stages differently when
 * log2 N is not a multiple of 4, src/ntt_radix4x4.c:81-113, and may then return
 * other representatives); identical after the header-inline reduction below.
 */
#ifndef NTT_MI355X_NTT_RADIX4X4_H
#define NTT_MI355X_NTT_RADIX4X4_H

#include "fast_mul_operators.h"

EXTERNC_BEGIN

NTT_EXPORT void fwd_ntt_radix4x4_lazy(uint64_t a[], uint64_t N, uint64_t q, const uint64_t w[],
                                      const uint64_t w_con[]);

static inline void fwd_ntt_radix4x4(uint64_t a[], const uint64_t N, const uint64_t q, const uint64_t w[],
                                    const uint64_t w_con[])
{
  fwd_ntt_radix4x4_lazy(a, N, q, w, w_con);
  for(size_t i = 0; i < N; i++) {
    a[i] = reduce_8q_to_q(a[i], q);
  }
}

EXTERNC_END
#endif /* NTT_MI355X_NTT_RADIX4X4_H */
