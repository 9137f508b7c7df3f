/*
 * pointer_batch.c -- plain-C caller of libntt_mi355x.so: SEPARATELY ALLOCATED polynomials in one launch (round 6).
 *
 *   gcc -O2 -std=gnu11 -Iinclude examples/pointer_batch.c \
 *       -Loptimized-number-theoretic-transform-implementations_amd -lntt_mi355x -o build/pointer_batch
 *
 * The reference package's only batch form is one array per polynomial -- fwd_ntt_ref_harvey_lazy_dbl(a1[], a2[], ...)
 * (reference include/ntt_reference.h:44-49, src/ntt_reference.c:71-91).  An FHE caller holds its ciphertext polynomials the same
 * way: each one an allocation of its own.  Here COUNT polynomials of N = 2^14 coefficients are allocated one by one (ntt_dev_malloc
 * each: the allocator decides where they land), transformed
 *   (1) through ntt_transform_ptrs     -- the pointers in a host array: sorted, checked for overlap, uploaded, ONE launch;
 *   (2) through ntt_transform_dev_ptrs -- the pointers in a device array, as they are: no copy, usable inside a HIP graph;
 * checked against a contiguous copy of the same polynomials through ntt_fwd_batch (word for word), then multiplied pair by pair
 * through ntt_negacyclic_mul_dev_ptrs and checked against the schoolbook definition.  Exit code 0 = everything matched.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ntt_mi355x.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if(rc_ != NTT_OK) {                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ntt_last_error());    \
      return 1;                                                                  \
    }                                                                            \
  } while(0)

enum { COUNT = 96 };

int main(void)
{
  const uint64_t N = 1u << 14, q = 0x7fffffffe0001ULL, root = 83051296654ULL;
  ntt_plan *     plan = NULL;
  CHECK(ntt_plan_create(&plan, 0, N, q, root, NTT_ARITH_AUTO));

  /* COUNT polynomials, one allocation each (sizes padded unevenly so that the allocator does not hand out a progression) */
  uint64_t *poly[COUNT], *slab = NULL, **d_table = NULL;
  CHECK(ntt_dev_malloc(0, (void **)&slab, (size_t)COUNT * N * 8));
  for(int i = 0; i < COUNT; i++) {
    CHECK(ntt_dev_malloc(0, (void **)&poly[i], (size_t)(N + 64 * (i % 5)) * 8));
    CHECK(ntt_fill_uniform(0, poly[i], N, q, 7, (uint64_t)i * N, NULL));      /* the same words as polynomial i of the slab below */
  }
  CHECK(ntt_fill_uniform(0, slab, (uint64_t)COUNT * N, q, 7, 0, NULL));
  CHECK(ntt_fwd_batch(plan, slab, COUNT, NULL));                               /* the reference result: one contiguous batch */

  /* (1) host array of device pointers, shuffled */
  uint64_t *shuffled[COUNT];
  for(int i = 0; i < COUNT; i++) shuffled[i] = poly[(i * 37) % COUNT];
  CHECK(ntt_transform_ptrs(plan, shuffled, COUNT, 0, NULL));
  CHECK(ntt_stream_sync(0, NULL));
  uint64_t *h1 = malloc(N * 8), *h2 = malloc(N * 8);
  int       ok = 1;
  for(int i = 0; i < COUNT; i++) {
    CHECK(ntt_d2h(0, h1, poly[i], N * 8));
    CHECK(ntt_d2h(0, h2, slab + (uint64_t)i * N, N * 8));
    if(memcmp(h1, h2, N * 8) != 0) ok = 0;
  }
  printf("%s: %d separately allocated polynomials through ntt_transform_ptrs: %s\n", ntt_version(), COUNT, ok ? "equal to the slab" : "DIFFER");

  /* (2) the same pointers in a DEVICE array: the inverse brings the inputs back */
  CHECK(ntt_dev_malloc(0, (void **)&d_table, COUNT * sizeof(uint64_t *)));
  CHECK(ntt_h2d(0, d_table, shuffled, COUNT * sizeof(uint64_t *)));
  CHECK(ntt_transform_dev_ptrs(plan, (const uint64_t *const *)d_table, COUNT, NTT_FLAG_INVERSE, NULL));
  CHECK(ntt_inv_batch(plan, slab, COUNT, NULL));
  CHECK(ntt_stream_sync(0, NULL));
  for(int i = 0; i < COUNT; i += 17) {
    CHECK(ntt_d2h(0, h1, poly[i], N * 8));
    CHECK(ntt_d2h(0, h2, slab + (uint64_t)i * N, N * 8));
    if(memcmp(h1, h2, N * 8) != 0) ok = 0;
  }
  printf("ntt_transform_dev_ptrs (inverse): %s\n", ok ? "equal to the slab" : "DIFFER");

  /* products over tables: c_p = a_p * b_p with a = the first half of the polynomials, b = the second half, c on a's table */
  uint64_t *a0 = malloc(N * 8), *b0 = malloc(N * 8);
  CHECK(ntt_d2h(0, a0, shuffled[0], N * 8));
  CHECK(ntt_d2h(0, b0, shuffled[COUNT / 2], N * 8));
  CHECK(ntt_negacyclic_mul_dev_ptrs(plan, (const uint64_t *const *)d_table, (const uint64_t *const *)d_table,
                                    (const uint64_t *const *)(d_table + COUNT / 2), COUNT / 2, NULL));
  CHECK(ntt_stream_sync(0, NULL));
  CHECK(ntt_d2h(0, h1, shuffled[0], N * 8));
  const uint64_t    k   = 4321;
  unsigned __int128 pos = 0, neg = 0;
  for(uint64_t i = 0; i < N; i++) {
    const uint64_t          j = (k + N - i) % N;
    const unsigned __int128 t = (unsigned __int128)a0[i] * b0[j] % q;
    if(i <= k) pos += t;
    else neg += t;
  }
  const uint64_t expect = (uint64_t)((pos % q + q - neg % q) % q);
  printf("ntt_negacyclic_mul_dev_ptrs: c[0][%llu] = %llu, schoolbook %llu\n", (unsigned long long)k, (unsigned long long)h1[k], (unsigned long long)expect);
  ok = ok && h1[k] == expect;

  for(int i = 0; i < COUNT; i++) ntt_dev_free(0, poly[i]);
  ntt_dev_free(0, slab);
  ntt_dev_free(0, d_table);
  ntt_plan_destroy(plan);
  free(h1), free(h2), free(a0), free(b0);
  return ok ? 0 : 1;
}
