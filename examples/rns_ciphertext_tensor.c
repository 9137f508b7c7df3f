/*
 * rns_ciphertext_tensor.c -- plain-C caller of the RNS entry points over DEVICE POINTER TABLES on the batch an FHE library really
 * has: the tensor step of a ciphertext multiplication.  Two ciphertexts (c0, c1) and (d0, d1); every polynomial is an allocation of
 * its own holding its LIMBS residues side by side ([limb][N], N = 2^14, eight 50-bit primes) -- four separately held RNS polynomials,
 * i.e. four workgroups' worth of work per prime.  The step computes
 *     e0 = c0 * d0,   e1 = c0 * d1 + c1 * d0,   e2 = c1 * d1        in Z_q[X]/(X^N+1), prime by prime
 * as   ntt_rns_transform_dev_ptrs            all four polynomials into the NTT domain            (one launch over all limbs)
 *      ntt_rns_inv_dot_dev_ptrs, k = 1, 2    e0, e2 (count 2: the pairs (c0,d0), (c1,d1)) and e1  (one launch over all limbs each)
 * -- three launches for the whole step; a loop over the primes would make 8 x 3 launches of one or two workgroups each
 * (profiles/r06/rns_pointer_small_batch.txt: 15-19 x slower at 16 primes).  Then e0 again as a coefficient-domain product of the
 * untouched copies (ntt_rns_negacyclic_mul_dev_ptrs).  One coefficient of every limb of e0, e1, e2 is checked against the schoolbook
 * definition.  The reference's batching precedent: one array per polynomial, fwd_ntt_ref_harvey_lazy_dbl(a1[], a2[], ...)
 * (reference include/ntt_reference.h:44-49, src/ntt_reference.c:71-91).
 *
 *   gcc -O2 -std=gnu11 -Iinclude examples/rns_ciphertext_tensor.c \
 *       -Loptimized-number-theoretic-transform-implementations_amd -lntt_mi355x -o build/rns_ciphertext_tensor
 * Exit code 0 = everything matched.
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

enum { LIMBS = 8 };

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
  const uint64_t N = 1u << 14, k = 4321, words = (uint64_t)LIMBS * N;
  uint64_t       q[LIMBS];
  ntt_plan *     plans[LIMBS];
  for(int l = 0; l < LIMBS; l++) {
    q[l] = ntt_find_prime(50, N, (unsigned)l);
    const uint64_t root = ntt_min_root(q[l], N);
    if(!q[l] || !root) return 3;
    CHECK(ntt_plan_create(&plans[l], 0, N, q[l], root, NTT_ARITH_AUTO));
  }
  /* c0, c1, d0, d1 (inputs), their untouched copies, e0, e1, e2: one allocation per RNS polynomial, unevenly padded */
  enum { C0, C1, D0, D1, KC0, KD0, E0, E1, E2, POLYS };
  uint64_t *poly[POLYS], *host[4];
  for(int i = 0; i < POLYS; i++) CHECK(ntt_dev_malloc(0, (void **)&poly[i], (size_t)(words + 64 * (i % 3)) * 8));
  for(int i = 0; i < 4; i++) {
    host[i] = malloc(words * 8);
    for(int l = 0; l < LIMBS; l++) CHECK(ntt_fill_uniform(0, poly[i] + (uint64_t)l * N, N, q[l], 11 + i, (uint64_t)l * N, NULL));
    CHECK(ntt_stream_sync(0, NULL));
    CHECK(ntt_d2h(0, host[i], poly[i], words * 8));
  }
  CHECK(ntt_h2d(0, poly[KC0], host[C0], words * 8));
  CHECK(ntt_h2d(0, poly[KD0], host[D0], words * 8));

  /* the device tables (entry = limb 0 of an RNS polynomial, its limbs N words apart) */
  uint64_t *h_tab[] = {/* all inputs */ poly[C0], poly[C1], poly[D0], poly[D1],
                       /* a of e0|e2 */ poly[C0], poly[C1], /* b of e0|e2 */ poly[D0], poly[D1], /* c of e0|e2 */ poly[E0], poly[E2],
                       /* e1: a_0, a_1 */ poly[C0], poly[C1], /* b_0, b_1 */ poly[D1], poly[D0], /* c */ poly[E1],
                       /* coefficient-domain e0 */ poly[KC0], poly[KD0]};
  uint64_t **d_tab = NULL;
  CHECK(ntt_dev_malloc(0, (void **)&d_tab, sizeof h_tab));
  CHECK(ntt_h2d(0, d_tab, h_tab, sizeof h_tab));
  typedef const uint64_t *const *table;
  const table all = (table)d_tab, a02 = (table)(d_tab + 4), b02 = (table)(d_tab + 6), c02 = (table)(d_tab + 8);
  const table a10 = (table)(d_tab + 10), a11 = (table)(d_tab + 11), b10 = (table)(d_tab + 12), b11 = (table)(d_tab + 13), c1 = (table)(d_tab + 14);
  const table kc0 = (table)(d_tab + 15), kd0 = (table)(d_tab + 16);

  CHECK(ntt_rns_transform_dev_ptrs(LIMBS, plans, all, 4, N, 0, NULL));
  const table a_e02[1] = {a02}, b_e02[1] = {b02};
  CHECK(ntt_rns_inv_dot_dev_ptrs(LIMBS, plans, c02, 1, a_e02, b_e02, 2, N, 0, NULL));          /* e0 = c0 d0, e2 = c1 d1 */
  const table a_e1[2] = {a10, a11}, b_e1[2] = {b10, b11};
  CHECK(ntt_rns_inv_dot_dev_ptrs(LIMBS, plans, c1, 2, a_e1, b_e1, 1, N, 0, NULL));             /* e1 = c0 d1 + c1 d0 */
  CHECK(ntt_stream_sync(0, NULL));

  uint64_t *e = malloc(words * 8);
  int       ok = 1;
  const int    which[3]  = {E0, E1, E2};
  const char * name[3]   = {"e0 = c0 d0", "e1 = c0 d1 + c1 d0", "e2 = c1 d1"};
  for(int r = 0; r < 3; r++) {
    CHECK(ntt_d2h(0, e, poly[which[r]], words * 8));
    int good = 1;
    for(int l = 0; l < LIMBS; l++) {
      const uint64_t *c0 = host[C0] + (uint64_t)l * N, *c1h = host[C1] + (uint64_t)l * N, *d0 = host[D0] + (uint64_t)l * N, *d1 = host[D1] + (uint64_t)l * N;
      uint64_t want = r == 0 ? schoolbook_coeff(c0, d0, N, q[l], k)
                      : r == 2 ? schoolbook_coeff(c1h, d1, N, q[l], k)
                               : (schoolbook_coeff(c0, d1, N, q[l], k) + schoolbook_coeff(c1h, d0, N, q[l], k)) % q[l];
      if(e[(uint64_t)l * N + k] != want) good = 0;
    }
    printf("%-20s %d limbs, coefficient %llu of every limb: %s\n", name[r], LIMBS, (unsigned long long)k, good ? "equal to the schoolbook value" : "DIFFERS");
    ok = ok && good;
  }
  /* e0 once more from the coefficient-domain copies: forward transforms, products and inverse in one launch over all limbs */
  uint64_t *first = malloc(words * 8);
  CHECK(ntt_d2h(0, first, poly[E0], words * 8));
  CHECK(ntt_rns_negacyclic_mul_dev_ptrs(LIMBS, plans, c02, kc0, kd0, 1, N, NULL));
  CHECK(ntt_stream_sync(0, NULL));
  CHECK(ntt_d2h(0, e, poly[E0], words * 8));
  int same = 1;
  for(uint64_t i = 0; i < words; i++) same = same && e[i] == first[i];
  printf("ntt_rns_negacyclic_mul_dev_ptrs on the untouched copies: %s\n", same ? "equal to e0, word for word" : "DIFFERS");
  ok = ok && same;

  for(int i = 0; i < POLYS; i++) ntt_dev_free(0, poly[i]);
  ntt_dev_free(0, d_tab);
  for(int l = 0; l < LIMBS; l++) ntt_plan_destroy(plans[l]);
  for(int i = 0; i < 4; i++) free(host[i]);
  free(e), free(first);
  printf("%s: %s\n", ntt_version(), ok ? "ok" : "FAILED");
  return ok ? 0 : 1;
}
