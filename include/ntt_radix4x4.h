/*
 * ntt_radix4x4.h -- radix-16-blocked forward NTT entry point, MI355X implementation.
 *
 * Drop-in replacement for reference include/ntt_radix4x4.h:10-28 /
 * src/ntt_radix4x4.c:41-114 (same expanded tables as ntt_radix4.h).  On the GPU
 * the "two radix-4 levels per block" idea is the native shape of every kernel:
 * each thread keeps a 16-coefficient tile in registers for four stages
 * (csrc/ntt_core.h), so this symbol shares the radix-4 engine of ntt_radix4.h
 * wherever the reference's radix-16 blocking is only another ORDER of the same
 * butterflies (log2 N = 4k, 4k+1, 4k+2: identical lazy words).  When log2 N =
 * 4k+3 the reference pairs the stages differently (src/ntt_radix4x4.c:91-111:
 * radix-2 stage BEFORE the last radix-4 layer, including its reduction of
 * a[group counter]); those sizes run layer by layer (csrc/ntt_core.h
 * r4x4_layer_*), so the lazy output equals the reference's bit for bit at every
 * size from 2^6 (tests/golden/lazy_words.json holds the reference's digests).
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
