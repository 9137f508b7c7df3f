/*
 * ntt_oracle.h -- CPU oracle for the negacyclic NTT hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (IBM/optimized-number-theoretic-transform-implementations).  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product library (libntt_mi355x.so) never links, loads or calls it.
 *
 * Parity status: PINNED.  oracle/Makefile builds the real reference from
 * /root/reference (outputs in oracle/_ref/) and oracle/gen_golden.py proves the
 * restatement bit-identical on all 19 reference parameter sets (reference
 * rand()%q stream and full-range inputs) and freezes the digests in
 * tests/golden/.
 *
 * Each function cites the reference file:line it follows.
 */
#ifndef NTT_ORACLE_H
#define NTT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef unsigned __int128 orc_u128;

/* ---- scalar arithmetic (include/internal/fast_mul_operators.h:15-70) ---- */
uint64_t orc_csub(uint64_t v, uint64_t bound);           /* v<bound ? v : v-bound   */
uint64_t orc_reduce_to_q(uint64_t v, uint64_t q, int k); /* [0,kq) -> [0,q), k=2,4,8 */
uint64_t orc_shoup_lazy(uint64_t w, uint64_t wcon, uint64_t t, uint64_t q); /* [0,2q) */
uint64_t orc_shoup_dbl_lazy(uint64_t w1, uint64_t c1, uint64_t w2, uint64_t c2,
                            uint64_t t1, uint64_t t2, uint64_t q);
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q); /* exact (u128 %) */
uint64_t orc_powmod(uint64_t b, uint64_t e, uint64_t q);
uint64_t orc_invmod(uint64_t a, uint64_t q);             /* q prime */

/* ---- table builders (include/internal/pre_compute.h:16-105) ---- */
uint64_t orc_bitrev(uint64_t idx, unsigned width);
/* w_powers[k] = root^{bitrev_m(k)} mod q, k in [0,N)            (:38-66) */
void orc_build_powers(uint64_t *out, uint64_t root, uint64_t N, uint64_t q);
/* con[k] = floor(w[k] * 2^word / q)                              (:68-83) */
void orc_build_precon(uint64_t *con, const uint64_t *w, uint64_t n, uint64_t q,
                      unsigned word);
uint64_t orc_precon1(uint64_t w, uint64_t q, unsigned word);
/* 2N-entry radix-4 table                                          (:85-105) */
void orc_expand_radix4(uint64_t *e, const uint64_t *w, uint64_t N, uint64_t q);

/* ---- radix-2 Harvey path (src/ntt_reference.c:11-91, include/ntt_reference.h) ---- */
void orc_fwd_r2_lazy(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *w,
                     const uint64_t *wcon);                       /* out < 4q */
void orc_fwd_r2(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *w,
                const uint64_t *wcon);                            /* out < q  */
void orc_inv_r2(uint64_t *a, uint64_t N, uint64_t q, uint64_t ninv,
                uint64_t ninv_con, unsigned word, const uint64_t *winv,
                const uint64_t *winv_con);                        /* out < q  */

/* ---- radix-4 path (src/ntt_radix4.c:7-114, include/ntt_radix4.h) ---- */
void orc_fwd_r4_lazy(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *e,
                     const uint64_t *econ);                       /* out < 8q */
/* fwd_ntt_radix4x4_lazy (src/ntt_radix4x4.c:41-114): same residues, its own lazy words when log2 N = 4k+3 */
void orc_fwd_r4x4_lazy(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *e, const uint64_t *econ);
void orc_fwd_r4(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *e,
                const uint64_t *econ);
void orc_inv_r4(uint64_t *a, uint64_t N, uint64_t q, uint64_t ninv,
                uint64_t ninv_con, const uint64_t *einv, const uint64_t *einv_con);

/* ---- independent definition check: O(N^2) evaluation ---- */
void orc_fwd_naive(uint64_t *out, const uint64_t *a, uint64_t N, uint64_t q,
                   uint64_t root);

/* ---- callers either side of the path (f1, f2) ---- */
void orc_pointwise(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n,
                   uint64_t q);
/* schoolbook a*b mod (X^N+1, q) */
void orc_negacyclic_schoolbook(uint64_t *c, const uint64_t *a, const uint64_t *b,
                               uint64_t N, uint64_t q);
int      orc_is_prime(uint64_t n);
/* smallest primitive 2N-th root of unity ("minimum root" rule of
 * tests/test_cases.h:113-142); 0 if 2N does not divide q-1 */
uint64_t orc_min_root(uint64_t q, uint64_t N);
/* largest prime p < 2^bits with p = 1 mod 2N, skipping `skip` hits */
uint64_t orc_find_prime(unsigned bits, uint64_t N, unsigned skip);

/* ---- deterministic inputs and digests ---- */
uint64_t orc_splitmix64(uint64_t x);
void orc_fill_uniform(uint64_t *a, uint64_t n, uint64_t q, uint64_t seed,
                      uint64_t offset); /* a[i]=splitmix64(seed^(offset+i)) % q */
uint64_t orc_fnv1a64(const uint64_t *a, uint64_t n);

/* ---- convenience context: every table one (q,N,root) needs ---- */
typedef struct orc_ctx {
  uint64_t N, q, root, root_inv, ninv, ninv_con;
  unsigned m;
  uint64_t *w, *wcon, *winv, *winv_con;     /* N entries each   */
  uint64_t *e, *econ, *einv, *einv_con;     /* 2N entries each  */
} orc_ctx;
orc_ctx *orc_ctx_new(uint64_t N, uint64_t q, uint64_t root);
void     orc_ctx_free(orc_ctx *c);

/* CPU baseline helpers for bench.py: run `reps` forward radix-4 transforms
 * (reduced output) over a batch laid out [batch][N], single thread. */
void orc_fwd_r4_batch(uint64_t *a, uint64_t batch, const orc_ctx *c);
void orc_fwd_r2_batch(uint64_t *a, uint64_t batch, const orc_ctx *c);
void orc_inv_r2_batch(uint64_t *a, uint64_t batch, const orc_ctx *c);
void orc_inv_r4_batch(uint64_t *a, uint64_t batch, const orc_ctx *c);

/* CPU timing harness for bench.py's cpu_baseline leg (oracle/cpu_bench.inc; the same code is compiled around the
 * real reference in oracle/ref_shim.c as ref_bench_*).  op: 0 forward, 1 forward + inverse, 2 limb-product. */
double orc_bench_single(int op, uint64_t N, uint64_t q, uint64_t ninv, const uint64_t *e, const uint64_t *econ,
                        const uint64_t *einv, const uint64_t *einvcon, int warmup, int outer, int inner);
int    orc_bench_threads(int op, uint64_t N, uint64_t q, uint64_t ninv, const uint64_t *e, const uint64_t *econ,
                         const uint64_t *einv, const uint64_t *einvcon, int nthreads, double seconds, uint64_t slab_ops,
                         double out[5]);

#ifdef __cplusplus
}
#endif
#endif
