/*
 * rns_chain_product.c -- plain-C caller of the RNS entry points over a modulus chain as FHE libraries build them: a 60-bit
 * first prime (served by the integer arithmetic in its throughput form), three 50-bit primes (FP64 arithmetic) and two 57-bit
 * primes; the primes and their roots come from the library's own parameter generation (ntt_find_prime / ntt_min_root, the
 * reference's "minimum root" rule, tests/test_cases.h:113-142).  The limb list is served as runs of compatible limbs, one
 * launch per pass and run.  Layout [limb][batch][N] first; then the same product with the operands as an FHE library holds them --
 * [batch][limb][N]: a polynomial's limbs side by side -- through the *_strided entry points, no transpose; then two
 * RNS polynomials handed over as a pointer batch (the reference's one-array-per-polynomial form, include/ntt_reference.h:44-49).
 *
 *   gcc -O2 -std=gnu11 -Iinclude examples/rns_chain_product.c \
 *       -Loptimized-number-theoretic-transform-implementations_amd -lntt_mi355x -o build/rns_chain_product
 *
 * Computes c = a * b limb by limb in Z_q[X]/(X^N+1), N = 2^12, two polynomials per limb, and checks one coefficient of every
 * limb against the schoolbook definition; then c^ += fwd(a) (.) key^ (ntt_rns_fwd_mul_batch, the multiply-accumulate of a key
 * switch) followed by ntt_rns_inv_batch against the same value.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ntt_mi355x.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if(rc_ != NTT_OK) {                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ntt_last_error());    \
      return 1;                                                                  \
    }                                                                            \
  } while(0)

#define LIMBS 6

static uint64_t schoolbook_coeff(const uint64_t *a, const uint64_t *b, uint64_t N, uint64_t q, uint64_t k)
{
  unsigned __int128 pos = 0, neg = 0;
  for(uint64_t i = 0; i < N; i++) {
    const uint64_t          j = (k + N - i) % N;
    const unsigned __int128 t = (unsigned __int128)a[i] * b[j] % q;
    if(i <= k) pos += t;
    else neg += t;
  }
  return (uint64_t)((pos % q + q - neg % q) % q);
}

int main(void)
{
  const uint64_t N = 1u << 12, batch = 2, slab = batch * N, k = 1234;
  const unsigned bits[LIMBS] = {60, 50, 50, 50, 57, 57};
  unsigned       seen[64]    = {0};
  uint64_t       q[LIMBS];
  ntt_plan *     plans[LIMBS];
  for(int l = 0; l < LIMBS; l++) {
    q[l] = ntt_find_prime(bits[l], N, seen[bits[l]]++); /* the skip-th largest prime below 2^bits with 2N | q - 1 */
    const uint64_t root = ntt_min_root(q[l], N);
    if(!q[l] || !root) return 3;
    CHECK(ntt_plan_create(&plans[l], 0, N, q[l], root, NTT_ARITH_AUTO));
  }
  uint64_t *d_a = NULL, *d_b = NULL, *d_c = NULL;
  CHECK(ntt_dev_malloc(0, (void **)&d_a, LIMBS * slab * 8));
  CHECK(ntt_dev_malloc(0, (void **)&d_b, LIMBS * slab * 8));
  CHECK(ntt_dev_malloc(0, (void **)&d_c, LIMBS * slab * 8));
  uint64_t *a = malloc(LIMBS * slab * 8), *b = malloc(LIMBS * slab * 8), *c = malloc(LIMBS * slab * 8);
  for(int l = 0; l < LIMBS; l++) {
    CHECK(ntt_fill_uniform(0, d_a + l * slab, slab, q[l], 11, l * slab, NULL));
    CHECK(ntt_fill_uniform(0, d_b + l * slab, slab, q[l], 12, l * slab, NULL));
  }
  CHECK(ntt_d2h(0, a, d_a, LIMBS * slab * 8));
  CHECK(ntt_d2h(0, b, d_b, LIMBS * slab * 8));

  CHECK(ntt_rns_negacyclic_mul_batch(LIMBS, plans, d_c, d_a, d_b, batch, NULL)); /* (a and b are scratch) */
  CHECK(ntt_stream_sync(0, NULL));
  CHECK(ntt_d2h(0, c, d_c, LIMBS * slab * 8));
  int      ok = 1;
  uint64_t expect[LIMBS];
  for(int l = 0; l < LIMBS; l++) {
    const uint64_t off = l * slab + N; /* polynomial 1 of the limb */
    expect[l]          = schoolbook_coeff(a + off, b + off, N, q[l], k);
    printf("limb %d q = %#llx (%u bits): c[1][%llu] = %llu, schoolbook %llu\n", l, (unsigned long long)q[l], bits[l],
           (unsigned long long)k, (unsigned long long)c[off + k], (unsigned long long)expect[l]);
    ok = ok && c[off + k] == expect[l];
  }

  /* the same product as c^ = 0 + fwd(a) (.) key^ with key^ = fwd(b), then back: the multiply-accumulate form of a key switch */
  CHECK(ntt_h2d(0, d_a, a, LIMBS * slab * 8));
  CHECK(ntt_h2d(0, d_b, b, LIMBS * slab * 8));
  CHECK(ntt_rns_fwd_batch(LIMBS, plans, d_b, batch, NULL));
  for(uint64_t i = 0; i < LIMBS * slab; i++) c[i] = 0;
  CHECK(ntt_h2d(0, d_c, c, LIMBS * slab * 8));
  CHECK(ntt_rns_fwd_mul_batch(LIMBS, plans, d_c, d_a, d_b, batch, NTT_MUL_ACCUMULATE, NULL));
  CHECK(ntt_rns_inv_batch(LIMBS, plans, d_c, batch, NULL));
  CHECK(ntt_stream_sync(0, NULL));
  CHECK(ntt_d2h(0, c, d_c, LIMBS * slab * 8));
  for(int l = 0; l < LIMBS; l++) ok = ok && c[l * slab + N + k] == expect[l];
  printf("multiply-accumulate in the NTT domain + inverse: %s\n", ok ? "same coefficients" : "MISMATCH");

  /* the same product with the operands laid out [batch][limb][N]: limb l of polynomial p at (p * LIMBS + l) * N.  The two
   * strides (words) are all the library needs: limb_stride = N, poly_stride = LIMBS * N. */
  uint64_t *abm = malloc(LIMBS * slab * 8), *bbm = malloc(LIMBS * slab * 8);
  for(int l = 0; l < LIMBS; l++) {
    for(uint64_t p = 0; p < batch; p++) {
      for(uint64_t i = 0; i < N; i++) {
        abm[(p * LIMBS + l) * N + i] = a[l * slab + p * N + i];
        bbm[(p * LIMBS + l) * N + i] = b[l * slab + p * N + i];
      }
    }
  }
  CHECK(ntt_h2d(0, d_a, abm, LIMBS * slab * 8));
  CHECK(ntt_h2d(0, d_b, bbm, LIMBS * slab * 8));
  CHECK(ntt_rns_negacyclic_mul_batch_strided(LIMBS, plans, d_c, d_a, d_b, N, LIMBS * N, batch, NULL));
  CHECK(ntt_stream_sync(0, NULL));
  CHECK(ntt_d2h(0, c, d_c, LIMBS * slab * 8));
  int ok_bm = 1;
  for(int l = 0; l < LIMBS; l++) ok_bm = ok_bm && c[(1 * LIMBS + l) * N + k] == expect[l];
  printf("the product on [batch][limb][N] operands (ntt_rns_negacyclic_mul_batch_strided): %s\n", ok_bm ? "same coefficients" : "MISMATCH");

  /* a pointer batch: the two RNS polynomials of d_a as two pointers (each [limb][N]); forward and back */
  CHECK(ntt_h2d(0, d_a, abm, LIMBS * slab * 8));
  uint64_t *polys[2] = {d_a + LIMBS * N, d_a}; /* (any order) */
  CHECK(ntt_rns_transform_ptrs(LIMBS, plans, polys, 2, N, 0, NULL));
  CHECK(ntt_rns_transform_ptrs(LIMBS, plans, polys, 2, N, NTT_FLAG_INVERSE, NULL));
  CHECK(ntt_stream_sync(0, NULL));
  CHECK(ntt_d2h(0, c, d_a, LIMBS * slab * 8));
  int ok_ptr = 1;
  for(uint64_t i = 0; i < LIMBS * slab; i++) ok_ptr = ok_ptr && c[i] == abm[i];
  printf("pointer batch of two RNS polynomials, forward and inverse (ntt_rns_transform_ptrs): %s\n", ok_ptr ? "round trip exact" : "MISMATCH");
  ok = ok && ok_bm && ok_ptr;
  free(abm);
  free(bbm);

  ntt_dev_free(0, d_a);
  ntt_dev_free(0, d_b);
  ntt_dev_free(0, d_c);
  for(int l = 0; l < LIMBS; l++) ntt_plan_destroy(plans[l]);
  free(a);
  free(b);
  free(c);
  return ok ? 0 : 2;
}
