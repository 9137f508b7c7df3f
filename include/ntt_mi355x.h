/*
 * ntt_mi355x.h -- batched, device-resident negacyclic NTT engine for MI355X (gfx950).
 *
 * This is the throughput API of libntt_mi355x.so.  The reference package
 * (IBM/optimized-number-theoretic-transform-implementations) has no batch API:
 * its only precedent is the two-polynomial fwd_ntt_ref_harvey_lazy_dbl
 * (reference include/ntt_reference.h:44-49, src/ntt_reference.c:71-91).  The
 * functions below generalise that to `batch` independent polynomials that stay
 * in HBM, with exactly the reference's transform semantics, so results compare
 * element-for-element with
 *
 *   fwd_ntt_ref_harvey / fwd_ntt_radix4      (include/ntt_reference.h:19-31,
 *                                             include/ntt_radix4.h:16-28)
 *   inv_ntt_ref_harvey / inv_ntt_radix4      (src/ntt_reference.c:33-66,
 *                                             src/ntt_radix4.c:64-114)
 *
 * i.e. forward: natural order in -> bit-reversed order out, values in [0,q);
 * inverse: bit-reversed in -> natural out, scaled by N^-1, values in [0,q).
 * The single-polynomial reference signatures themselves are exported by the same
 * library (include/ntt_reference.h, ntt_radix4.h, ntt_radix4x4.h, ntt_seal.h).
 *
 * Plain C ABI: pointers and sizes only.  `stream` arguments are hipStream_t
 * passed as void* (NULL = the device's default stream).  All functions return
 * NTT_OK (0) or a negative ntt_status; ntt_last_error() describes the failure.
 * There is no CPU fallback: without a HIP device every compute entry point
 * fails with NTT_ERR_NO_DEVICE.
 */
#ifndef NTT_MI355X_H
#define NTT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#  define NTT_API __attribute__((visibility("default")))
#else
#  define NTT_API
#endif

typedef enum ntt_status {
  NTT_OK              = 0,
  NTT_ERR_ARG         = -1, /* bad argument (N not a power of two, q not NTT-friendly ...) */
  NTT_ERR_NO_DEVICE   = -2, /* no HIP device / HIP runtime unusable                        */
  NTT_ERR_HIP         = -3, /* a HIP call failed; see ntt_last_error()                     */
  NTT_ERR_UNSUPPORTED = -4, /* e.g. FP64 arithmetic requested for q > 2^51                 */
  NTT_ERR_NOMEM       = -5
} ntt_status;

typedef enum ntt_arith {
  NTT_ARITH_AUTO = 0, /* FP64 path when q allows it (q < 2^52), else 64-bit integer Shoup in its throughput form
                       * (NTT_OPT_INT_WIDE) */
  NTT_ARITH_U64  = 1, /* reference-identical Harvey/Shoup lazy arithmetic, any q<2^61 */
  NTT_ARITH_F64  = 2, /* balanced FP64 arithmetic: q <= 2^51(1+2^-10) with a compile-time reduction schedule,
                       * up to q < 2^52 with both operands of every butterfly reduced (info[4] == 52)  */
  NTT_ARITH_U64_R4 = 3 /* the reference's radix-4 butterflies with the shared-quotient double
                        * product (include/internal/fast_mul_operators.h:62-70,108-149) on the 2N-entry
                        * expanded table (src/ntt_radix4.c:7-114); q < 2^60; 2^6..2^18 (two passes above 2^14).
                        * Never chosen by AUTO. */
} ntt_arith;

typedef enum ntt_option {
  NTT_OPT_MAX_GRID  = 1, /* cap on workgroups per launch (0 = the kernels' own choice)             */
  NTT_OPT_CHUNK_MIB = 2, /* bytes of one chunk of a multi-pass transform, MiB (default 256)        */
  NTT_OPT_F64_CLASS = 3, /* force a coarser FP64 headroom class (0, 1 or 18) than q permits: tests  */
  NTT_OPT_TWO_PHASE = 4, /* N = 2^16, 2^17 (FP64): 1 = both passes of a polynomial in one workgroup (one launch),
                          * 0 = one launch per pass over the whole batch, -1 (default) = the faster of the two as
                          * measured: one launch for the forward transform at 2^16, per pass elsewhere */
  NTT_OPT_BLOCK_LOG = 6, /* N = 2^15, 2^16: log2 of the blocks the fused pass works on below the column pass: 12
                          * (3 or 4 column stages), 14 (1 or 2), 0 (default) = the faster one as measured.  Results
                          * are identical. */
  NTT_OPT_XCD_LOCAL = 7,  /* N = 2^15..2^17 (FP64 policies; wide integer policy): 1 = both passes of a transform as items of ONE launch, every polynomial
                           * handled by the workgroups of one XCD, the intermediate handed over inside that XCD (MEASURED
                           * fabric traffic per transform, FETCH x2 + WRITE counters: 24N bytes at 2^15, where the L2 retains
                           * the intermediate at the shipped lag; 32N at 2^16 and 2^17, where it does not and the second pass is
                           * served by the Infinity Cache -- the gain there is one launch instead of two per chunk);
                           * 0 = one launch per pass; -1 (default) = where it measured faster: forward transforms of 512
                           * polynomials or more (wide integer policy: +18..23 %, where the memory-bound column items overlap the
                           * multiplier-bound row items; also its inverse at 2^17).  The NTT-domain products (ntt_inv_product_batch,
                           * ntt_inv_dot_batch, their RNS forms) follow the same switch: 1 = the k products with the inverse's block
                           * stages and the inverse's column stages as the items of one launch, 0 = two launches per 128 / 256 MiB
                           * chunk, -1 = one launch from 2^25 coefficients per operand on (measured +11..26 %,
                           * profiles/r05/domain_bench_xcd_local.txt); so does ntt_fwd_mul_batch (the forward column stages of a and
                           * the blocks with the product as one launch: +8..28 %, automatic from 2^26 coefficients on).  Results are
                           * identical. */
  NTT_OPT_XCD_LOCAL_LAG = 8,        /* tuning: polynomials between the two passes of a queue (0 = default: transforms 10, 8, 10 at 2^15,
                                     * 2^16, 2^17; NTT-domain products 20-24, 10-14, 6-8 -- about 5-6 MiB of c per queue) */
  NTT_OPT_XCD_LOCAL_WGS_PER_CU = 9, /* tuning: resident workgroups per CU, 1..4 (0 = default 4) */
  NTT_OPT_INT_WIDE = 10, /* integer policy, 2^40 <= q < 2^61: 1 = transforms through the throughput form of the integer
                          * arithmetic (estimated Shoup quotient, no conditional subtraction per butterfly: the bits between q
                          * and 2^64 absorb the growth; 19 instead of 28 instructions per butterfly, +16 % measured) -- same
                          * tables, same canonical results, lazy outputs inside the same ranges but NOT the reference's lazy
                          * words; 0 = the reference's Harvey butterflies.  Default: 1 for NTT_ARITH_AUTO plans (q >= 2^52),
                          * 0 for plans created with NTT_ARITH_U64.  10 + K forces headroom class K in {0, 1, 3} (tests) */
  NTT_OPT_RNS_LAUNCH = 12, /* ntt_rns_*: how a run of compatible limbs is launched -- 0 = ONE launch (per pass) over the run wherever
                          * the kernels have the variant, 1 = one launch chain per limb, -1 (default) = one launch where a limb's share
                          * alone cannot fill the chip and for the XCD-local launches.  Read from the run's first plan; results are
                          * identical (tests, measurements) */
  NTT_OPT_DOT_FUSED = 13, /* NTT-domain products (ntt_inv_dot_batch, ntt_fwd_mul_batch, ...): 1 (default) = the products inside the
                          * transform's first / last pass; 0 = pointwise(-accumulate) launches around a plain transform (measurements) */
  NTT_OPT_MAX_BATCH_HINT = 14, /* polynomials x limbs of the largest batched call the plan will serve: the control blocks of the
                          * XCD-local launches (N >= 2^15) are sized for it -- for the null stream at once, for any other stream at its
                          * first call or by ntt_plan_reserve -- so that no later call allocates (an allocation synchronises the device).
                          * 0 (default): sized by the first call, doubled when outgrown */
  NTT_OPT_CTL_ALLOCATIONS = 15, /* READ-ONLY (ntt_plan_get_option): device allocations the plan has made for its control blocks so far --
                          * unchanged across a call = that call did not allocate (what ntt_plan_reserve promises) */
  NTT_OPT_BLOCK_OVERSUB = 11, /* persistent block kernels: workgroups launched per resident slot (0 = default: 8 for the 2^12-point
                          * block kernels, whose four workgroups per CU otherwise run in phase -- measured +5 % forward, +4 % inverse,
                          * profiles/r05/grid_sweep.txt --, 1 elsewhere: 2^13 and 2^14 measured no gain) */
  NTT_OPT_ONE_PASS = 16, /* N = 2^15, FP64 policies (q < 2^52): 1 = the transform in ONE pass over the data -- a 1024-thread workgroup holds
                          * the whole polynomial (32 words per thread) in its registers, the stage on pairs 2^14 apart runs thread-locally
                          * and the two halves go through the 2^14-point block stages one after the other: 16N bytes cross HBM, where the
                          * two-pass forms move 24N..32N across the fabric (measured forward 0.43 -> see profiles/r06/onepass_2p15.txt);
                          * 0 = the two-pass forms (XCD-local launch / per-pass launches); -1 (default) = one pass when the batch gives
                          * every second CU a polynomial (measured crossover: 64..96 polynomials).  Calls that ask for lazy outputs get canonical words from it
                          * (inside the lazy ranges).  Results are identical. */
  NTT_OPT_FUSED_PRODUCT = 5 /* N = 2^8..2^17, FP64: 1 (default) = ntt_negacyclic_mul_batch as ONE launch that takes both
                          * operands through the forward stages, multiplies in registers and runs the inverse: 24N bytes up to
                          * 2^14; from 2^23 coefficients per operand of N >= 2^15 on likewise one launch (all limbs of an RNS set
                          * included), whose MEASURED fabric traffic is about 85N bytes per product at 2^17 (a, b, two of the
                          * three intermediates and the twiddles: the L2 retains nothing at that size) against 56N algorithmic;
                          * smaller batches of N >= 2^15: block by block between the column passes of both operands, 72N
                          * bytes); 2 = a's forward transform always as a launch of its own in front of the fused
                          * fwd(b)*a^ -> inverse kernel (40N bytes up to 2^14); 0 = fwd, fwd, then the products inside the inverse's
                          * first pass (three launches, 56N bytes up to 2^14: what plans of the integer policies and squarings
                          * always take).  Results are identical. */
} ntt_option;

typedef struct ntt_plan ntt_plan; /* opaque: tables for one (device, N, q, root) */

/* ---- library / device ---- */
NTT_API const char *ntt_last_error(void);
NTT_API int         ntt_device_count(void);             /* <0: ntt_status */
NTT_API const char *ntt_version(void);

/* ---- plans ----
 * root must be a primitive 2N-th root of unity mod q (the `w` column of
 * reference tests/test_cases.h:145-208).  The plan derives every table itself
 * with the reference's layouts (include/internal/pre_compute.h:38-105) -- ON THE DEVICE: the host squares the
 * root log2 N times, one GPU thread produces each table entry. */
NTT_API int  ntt_plan_create(ntt_plan **out, int device, uint64_t N, uint64_t q,
                             uint64_t root, int arith);
/* build from caller tables in the reference's radix-2 layout: w_powers[k] =
 * root^bitrev(k), N entries each; w_inv_powers may be NULL (forward-only plan) */
NTT_API int  ntt_plan_create_from_tables(ntt_plan **out, int device, uint64_t N, uint64_t q,
                                         const uint64_t *w_powers,
                                         const uint64_t *w_inv_powers, int arith);
NTT_API void ntt_plan_destroy(ntt_plan *p);
/* info[0..7] = {N, q, log2N, arith actually used, FP64 headroom class,
 *               number of HBM passes, device, root (0 if built from tables)} */
NTT_API int  ntt_plan_info(const ntt_plan *p, uint64_t info[8]);
/* copy a device table back (tests / debugging): which = 0 forward records, 1 inverse records (+16 folded N^-1 records),
 * 2 / 3 the FP64 policy's compact forward / inverse tables.  Records are 16 bytes: {w, floor(w 2^64/q)} as two
 * uint64_t (integer policies; 2N records of the expanded table for NTT_ARITH_U64_R4) or {balanced w, w/q} as two
 * doubles (FP64). */
NTT_API int  ntt_plan_export_table(const ntt_plan *p, int which, void *h_dst, size_t bytes);
/* force the strided multi-pass path (self-check of the fused kernels) */
NTT_API int  ntt_plan_set_generic(ntt_plan *p, int on);
/* tuning / test knobs of one plan (ntt_option).  The batched API reads NO environment variable; the reference-signature entry points
 * (which have no argument to carry a choice) read NTT_DEVICE, NTT_COMPAT_ARITH and NTT_COMPAT_ZERO_COPY (0 = stage single-pass
 * transforms through device memory instead of running them on a pinned, device-mapped host buffer) once, at their first call. */
NTT_API int  ntt_plan_set_option(ntt_plan *p, int option, int64_t value);
NTT_API int  ntt_plan_get_option(const ntt_plan *p, int option, int64_t *value); /* the value in force (0 / -1 = the default, as set) */
/* Allocates the control blocks (queue heads + one counter per polynomial, 4 bytes each; the direct one and the one captured
 * launches use) the XCD-local launches of this plan need on `stream` for batches of up to `polys` polynomials x limbs.  The
 * batched entry points take a const plan; these blocks are the one thing they may create or grow (under the plan's mutex), and
 * after this call they do not.  Call it outside stream capture. */
NTT_API int  ntt_plan_reserve(const ntt_plan *p, void *stream, uint64_t polys);

/* ---- batched transforms: d_a is device memory laid out [batch][N], in place ----
 * Input contract: ntt_fwd_batch / ntt_inv_batch (and every entry point that does not say otherwise) take CANONICAL words, 0 <= a < q.
 * The library does not check it, and plans for 2^51 < q < 2^52 (info[4] == 52) rely on it: the first inverse stage multiplies the
 * unreduced difference of two inputs, exact only while that difference is below q in magnitude -- words in [q, 2q) handed to
 * ntt_inv_batch give wrong results there (rounds 3-4 happened to tolerate them).  Lazy words of any range this header names go
 * through the *_wide entry points (NTT_FLAG_WIDE_IN), which fold them first. */
NTT_API int ntt_fwd_batch(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream);
NTT_API int ntt_inv_batch(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream);
/* lazy outputs: the reference's *_lazy contract (include/ntt_reference.h:13-17, tests/bench.c:123-137) --
 * the final reduction is left to the consumer.  Forward: values in [0,4q) (radix-4 policy: [0,8q), bit-identical
 * to fwd_ntt_radix4_lazy); inverse: [0,2q).  Feed them to the *_wide entry points, to
 * ntt_pointwise_mul_batch... after reduce_*_to_q, or to the fused product below.  A policy that has no
 * cheaper lazy form for a direction (FP64 inverse) returns reduced values, which satisfy the contract. */
NTT_API int ntt_fwd_batch_lazy(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream);
NTT_API int ntt_inv_batch_lazy(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream);
/* every combination in one call: flags = NTT_FLAG_* or'ed together */
enum { NTT_FLAG_INVERSE = 1, NTT_FLAG_WIDE_IN = 2, NTT_FLAG_LAZY_OUT = 4 };
NTT_API int ntt_transform_batch(const ntt_plan *p, uint64_t *d_a, uint64_t batch, unsigned flags, void *stream);
/* same as ntt_fwd_batch / ntt_inv_batch, but inputs may be lazy values in [0,8q) (what the reference's *_lazy
 * entry points accept and emit); outputs are still in [0,q) */
NTT_API int ntt_fwd_batch_wide(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream);
NTT_API int ntt_inv_batch_wide(const ntt_plan *p, uint64_t *d_a, uint64_t batch, void *stream);

/* ---- callers either side of the path ---- */
/* d_c[i] = d_a[i]*d_b[i] mod q over n = batch*N values in [0,q); c may alias a or b */
NTT_API int ntt_pointwise_mul_batch(const ntt_plan *p, uint64_t *d_c, const uint64_t *d_a,
                                    const uint64_t *d_b, uint64_t batch, void *stream);
/* the same with LAZY operands in [0,4q) (ntt_fwd_batch_lazy outputs); the product is fully reduced */
NTT_API int ntt_pointwise_mul_batch_lazy(const ntt_plan *p, uint64_t *d_c, const uint64_t *d_a,
                                         const uint64_t *d_b, uint64_t batch, void *stream);
/* c = a*b in Z_q[X]/(X^N+1) for every polynomial of the batch:
 * ONE launch (FP64 policies, NTT_OPT_FUSED_PRODUCT), else fwd(a), fwd(b) and the products inside the inverse transform's first
 * pass -- the chain stays in the lazy domain until the inverse's output.
 * d_a is overwritten (left in the NTT domain as LAZY values in [0,4q), congruent to the reference's
 * transform -- or, large batches of N >= 2^15, holding only the column stages of it -- or, the one-launch form up to
 * 2^14, not written at all); d_b is overwritten likewise
 * (three-launch chain), overwritten by the column passes of its forward transform
 * (fused product, N = 2^15 .. 2^17) or left as it was (fused product, N = 2^8 .. 2^14); callers must not rely on any of these.  Aliasing rules: d_c may alias d_a or d_b; d_a == d_b computes the square
 * a*a (the shared operand is transformed once); any other overlap is undefined.
 * NTT_ARITH_U64_R4 plans run the reference's radix-4 formulation end to end (fwd_ntt_radix4 on both operands, the
 * pointwise product, inv_ntt_radix4): d_a and d_b are left canonical there. */
NTT_API int ntt_negacyclic_mul_batch(const ntt_plan *p, uint64_t *d_c, uint64_t *d_a,
                                     uint64_t *d_b, uint64_t batch, void *stream);

/* ---- operands that already ARE in the NTT domain ("fusing the multiply into the inverse's first load
 * saves 16N bytes").  Keys, plaintexts and ciphertexts of an FHE caller live in the NTT domain (bit-reversed order, as
 * ntt_fwd_batch leaves them); what such a caller issues is the element-wise product of two transformed operands, or the
 * inner product of a digit-decomposed ciphertext with a key, followed by ONE inverse transform.  The reference's
 * primitive for it is fast_mul_mod_q (include/internal/fast_mul_operators.h:56-60) in a loop of its own in front of
 * inv_ntt_*; here the products are formed inside the inverse transform's first pass, so neither the product nor the sum
 * ever exists in memory.
 *   flags: NTT_MUL_LAZY_IN      transformed operand words may be LAZY -- anywhere in [0,4q) (ntt_fwd_batch_lazy outputs, the
 *                               reference's *_lazy range, include/ntt_reference.h:13-17) -- instead of canonical
 *                               ([0,q)); ntt_mul_transformed_batch additionally needs them below 2^53
 *          NTT_MUL_B_BROADCAST  every d_bhat[i] is ONE polynomial (N words; RNS: [limb][N]) shared by all polynomials of
 *                               the batch: a key.  Traffic 8kN + 8N instead of 16kN + 8N bytes per output polynomial.
 * Outputs are canonical coefficients in natural order, exactly inv_ntt_ref_harvey(pointwise products) of the reference. */
enum { NTT_MUL_LAZY_IN = 1, NTT_MUL_B_BROADCAST = 2,
       NTT_MUL_ACCUMULATE = 4 /* ntt_fwd_mul_batch: c^ += ... instead of c^ = ... (c^ canonical on entry) */ };
/* c = inv( a^ (.) b^ ): ONE launch up to N = 2^14 (24N bytes instead of 40N for pointwise + inverse); above, the product
 * rides in the first pass of the inverse (40N instead of 56N).  d_c may alias d_ahat or d_bhat (not a broadcast b^). */
NTT_API int ntt_inv_product_batch(const ntt_plan *p, uint64_t *d_c, const uint64_t *d_ahat, const uint64_t *d_bhat,
                                  uint64_t batch, unsigned flags, void *stream);
/* c = inv( sum_{i<k} a_i^ (.) b_i^ ), 1 <= k <= 32 (key switching: digits x key): d_ahat / d_bhat are HOST arrays of k
 * device pointers, each operand laid out [batch][N].  Reads 16kN bytes (8kN with a broadcast key), writes 8N, one
 * inverse transform.  For k > 1 d_c must not overlap any operand. */
NTT_API int ntt_inv_dot_batch(const ntt_plan *p, uint64_t *d_c, int k, const uint64_t *const *d_ahat,
                              const uint64_t *const *d_bhat, uint64_t batch, unsigned flags, void *stream);
/* c = inv( fwd(a) (.) b^ ): a in coefficients, b^ transformed beforehand (a plaintext or key kept in the NTT domain).  One
 * launch that takes a through the forward stages, multiplies by b^ in registers and runs the inverse (24N bytes up to
 * 2^14).  d_a is left as it was up to N = 2^14 and OVERWRITTEN (scratch) above -- and at every size by plans the fused kernels
 * are not built for (integer and radix-4 policies, column-only plans: forward transform in place, pointwise, inverse): treat
 * it as scratch unless you know the plan.  d_c may alias d_a or d_bhat. */
NTT_API int ntt_mul_transformed_batch(const ntt_plan *p, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat, uint64_t batch,
                                      unsigned flags, void *stream);
/* c^ = fwd(a) (.) b^, or with NTT_MUL_ACCUMULATE c^ += fwd(a) (.) b^: a in coefficients, the result STAYS in the NTT domain
 * (canonical words, bit-reversed order as ntt_fwd_batch leaves them): a plaintext or key-switching key kept transformed is
 * multiplied in where the forward transform would reduce and store its outputs -- the multiply-accumulate of a key-switching
 * inner product, digit by digit.  ONE launch up to N = 2^14: 24N bytes (16N with NTT_MUL_B_BROADCAST) instead of 40N for
 * ntt_fwd_batch + ntt_pointwise_mul_batch; accumulating 32N (24N) instead of 48N.  N = 2^15 (FP64 policies, NTT_OPT_ONE_PASS): the
 * one-pass transform with the product at its output, 24N bytes, d_a left as it was (round 6).  Else above 2^14 the product rides in
 * the block pass of the forward transform: one launch over both passes from 2^26 coefficients per operand on (NTT_OPT_XCD_LOCAL), else a column and a
 * block launch per 256 MiB chunk.  d_a is left as it was up to 2^14 and
 * OVERWRITTEN (scratch) above -- and at every size by plans without the fused kernel (radix-4 policy, column-only plans, N <
 * 2^6: forward transform in place, then a pointwise launch); d_c may alias d_a (not when accumulating) or d_bhat. */
NTT_API int ntt_fwd_mul_batch(const ntt_plan *p, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat, uint64_t batch, unsigned flags,
                              void *stream);

/* ---- RNS limbs (BASELINE config 5: 4-prime RNS pipeline).  plans[l] is the plan of
 * prime q_l (same N, same device); data layout is [limb][batch][N], i.e. limb l of
 * every polynomial is the contiguous [batch][N] slab at d_x + l*batch*N.  Each limb is
 * an independent transform -- the reference has no counterpart; its closest
 * primitive is fast_mul_mod_q (include/internal/fast_mul_operators.h:56-60).
 * When one limb's share alone cannot fill the GPU (a ciphertext: a few polynomials x tens of primes) and the limbs'
 * plans agree in policy and options (the headroom class may differ: the launch takes the coarsest), ONE launch per pass serves up to 16 limbs: a
 * workgroup picks its limb's tables and constants from an array in the kernel arguments -- for the FP64 policies (q < 2^52)
 * and for the wide integer policy (NTT_ARITH_AUTO plans of 2^52 <= q < 2^61; a product of such a set is three launches:
 * both forward transforms and the products inside the inverse's first pass).  At N = 2^15..2^17 (FP64 policies) large per-limb
 * batches are ONE launch over the limbs too (the XCD-local kernels take the limb as part of their queue entries); other large
 * batches are served limb by limb.  A modulus chain with primes of several sizes (a 60-bit first prime in front of 50-bit
 * ones) is served as maximal RUNS of consecutive compatible limbs: one launch per pass and run, single limbs by themselves.
 * Results are identical either way (NTT_OPT_RNS_LAUNCH on the limbs' plans forces either form). ---- */
NTT_API int ntt_rns_fwd_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_a, uint64_t batch, void *stream);
NTT_API int ntt_rns_inv_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_a, uint64_t batch, void *stream);
NTT_API int ntt_rns_negacyclic_mul_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a,
                                         uint64_t *d_b, uint64_t batch, void *stream);
/* the NTT-domain products above over RNS limbs: every operand laid out [limb][batch][N] (a broadcast b^: [limb][N]) */
NTT_API int ntt_rns_inv_dot_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, int k, const uint64_t *const *d_ahat,
                                  const uint64_t *const *d_bhat, uint64_t batch, unsigned flags, void *stream);
NTT_API int ntt_rns_mul_transformed_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat,
                                          uint64_t batch, unsigned flags, void *stream);
NTT_API int ntt_rns_fwd_mul_batch(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat,
                                  uint64_t batch, unsigned flags, void *stream);

/* ---- caller-native layouts (round 5).  The entry points above take RNS operands as [limb][batch][N].  SURVEY 8(d) config 5
 * -- and every FHE library -- keeps a polynomial's limbs side by side: [batch][prime][N].  The *_strided forms take the two
 * distances in WORDS instead of assuming either:
 *     coefficient i of limb l of polynomial p  =  d_x[l * limb_stride + p * poly_stride + i]
 *   [limb][batch][N]:  limb_stride = batch * N, poly_stride = N           (what the plain entry points pass)
 *   [batch][limb][N]:  limb_stride = N,         poly_stride = nlimbs * N  (no transpose on either side of the call)
 * Padded variants of either are accepted; strides under which two (limb, polynomial) ranges would overlap are refused
 * (NTT_ERR_ARG).  All operands of one call (a, b, c, every a_i^ / b_i^) share the layout; a broadcast b^ (NTT_MUL_B_BROADCAST) stays
 * [limb][N].  Same kernels, same launch choices (one launch over the limbs where it pays, the XCD-local launches for large
 * batches of N >= 2^15), same results: the layout is one address computation per block (csrc/ntt_core.h block_offset).
 * The reference's own batching precedent is two caller arrays side by side, fwd_ntt_ref_harvey_lazy_dbl(a1[], a2[], ...)
 * (include/ntt_reference.h:44-49, src/ntt_reference.c:71-91); these generalise it to any regular placement. ---- */
NTT_API int ntt_rns_fwd_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_a, uint64_t limb_stride, uint64_t poly_stride,
                                      uint64_t batch, void *stream);
NTT_API int ntt_rns_inv_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_a, uint64_t limb_stride, uint64_t poly_stride,
                                      uint64_t batch, void *stream);
NTT_API int ntt_rns_negacyclic_mul_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, uint64_t *d_b,
                                                 uint64_t limb_stride, uint64_t poly_stride, uint64_t batch, void *stream);
NTT_API int ntt_rns_inv_dot_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, int k, const uint64_t *const *d_ahat,
                                          const uint64_t *const *d_bhat, uint64_t limb_stride, uint64_t poly_stride, uint64_t batch,
                                          unsigned flags, void *stream);
NTT_API int ntt_rns_mul_transformed_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat,
                                                  uint64_t limb_stride, uint64_t poly_stride, uint64_t batch, unsigned flags, void *stream);
NTT_API int ntt_rns_fwd_mul_batch_strided(int nlimbs, ntt_plan *const *plans, uint64_t *d_c, uint64_t *d_a, const uint64_t *d_bhat,
                                          uint64_t limb_stride, uint64_t poly_stride, uint64_t batch, unsigned flags, void *stream);
/* one plan, `batch` polynomials poly_stride words apart (one limb of a [batch][limb][N] operand; a column of a caller's
 * matrix of polynomials): ntt_transform_batch with a stride */
NTT_API int ntt_transform_batch_strided(const ntt_plan *p, uint64_t *d_a, uint64_t poly_stride, uint64_t batch, unsigned flags,
                                        void *stream);

/* ---- pointer batches: one DEVICE pointer per polynomial.  The reference's own batch form is one array per polynomial --
 * fwd_ntt_ref_harvey_lazy_dbl(a1[], a2[], ...) (include/ntt_reference.h:44-49, src/ntt_reference.c:71-91) --; this is that form for
 * `count` polynomials resident on the device and placed ANYWHERE: ONE launch chain serves the whole batch, the kernels read each
 * polynomial's address from a device table (separately allocated ciphertexts, a shuffled pool, rows of several matrices: 4096
 * separately held 2^14-point polynomials run at the rate of one contiguous slab; rounds 1-5 launched every arithmetic progression
 * of the sorted pointers by itself).  Pointers must be 8-byte aligned.  flags = NTT_FLAG_*.
 *   ntt_transform_ptrs      h_polys is a HOST array.  The polynomials are independent and transformed in place, so the call is free
 *                           to reorder them: the pointers are sorted (address order), checked -- overlapping polynomials or a pointer
 *                           listed twice are refused, NTT_ERR_ARG -- and uploaded through a pinned staging buffer the plan keeps per
 *                           stream (asynchronous; its first use on a stream allocates).  A batch that is one arithmetic progression
 *                           takes the strided launch, no table.  While `stream` is being captured into a HIP graph nothing can be
 *                           uploaded: the call then launches progression by progression (no regularity: one launch chain per
 *                           polynomial) -- in graphs use
 *   ntt_transform_dev_ptrs  d_polys is a DEVICE array of `count` device pointers, used as it is: no copy, no allocation, no check
 *                           (overlapping polynomials are the caller's responsibility), capturable; it must stay valid and unchanged
 *                           until the call's kernels have run.
 * ntt_rns_transform_ptrs / _dev_ptrs: entry i points at limb 0 of RNS polynomial i, whose limbs are limb_stride words apart ([limb][N]
 * per polynomial: limb_stride = N; pointers INTO a [limb][batch][N] slab: limb_stride = batch * N -- the polynomials then interleave
 * without overlapping, which the host form's check, made limb image by limb image, accepts); flags: NTT_FLAG_INVERSE only. ---- */
NTT_API int ntt_transform_ptrs(const ntt_plan *p, uint64_t *const *h_polys, uint64_t count, unsigned flags, void *stream);
NTT_API int ntt_transform_dev_ptrs(const ntt_plan *p, const uint64_t *const *d_polys, uint64_t count, unsigned flags, void *stream);
NTT_API int ntt_rns_transform_ptrs(int nlimbs, ntt_plan *const *plans, uint64_t *const *h_polys, uint64_t count, uint64_t limb_stride,
                                   unsigned flags, void *stream);
NTT_API int ntt_rns_transform_dev_ptrs(int nlimbs, ntt_plan *const *plans, const uint64_t *const *d_polys, uint64_t count,
                                       uint64_t limb_stride, unsigned flags, void *stream);

/* ---- products over pointer batches (round 6).  Every operand is a DEVICE array of `count` device pointers (entry p = polynomial p of
 * that operand; RNS: its limb 0, the limbs limb_stride words apart), taken as it is: no copy, no overlap check, capturable.
 * Up to N = 2^14 the fused product kernels read every operand through its own table: ONE launch per call and limb at the slab forms'
 * bytes (24N for c = a * b, (2k + 1) 8N for an inner product of k pairs, 24N / 32N for fwd(a) (.) b^ (+ c^)) -- measured 0.84-1.14 x the
 * rate of the same product over contiguous slabs (profiles/r06/pointer_products.txt).  fwd(a) (.) b^ at 2^15 takes the one-pass kernel
 * the same way, and at 2^15..2^17 the XCD-local one-launch kernels read the tables from 64 polynomials on (they need the plan's control
 * block: ntt_plan_reserve before capturing such a call into a graph).  Smaller batches of those sizes, and plans the fused kernels are
 * not built for, run the element-wise products as a table-reading kernel of their own and the transforms over the tables: k + 1 launch
 * chains per call whatever the batch.  The ntt_rns_* twins: a run of compatible limbs (as for the slab forms: same policy and size
 * class; up to 16) whose per-limb share cannot fill the chip -- a few ciphertext polynomials x many primes -- goes out as ONE launch of
 * the fused kernels over all of its limbs up to N = 2^14 (the limb an index of the grid; NTT_OPT_RNS_LAUNCH 0 / 1 forces the
 * one-launch / the per-limb form; large batches up to 2^14 take one launch per limb); above 2^14 a run is ONE XCD-local launch from 64
 * polynomials x limbs on (the limb part of the queue entry) and one chain over all of its limbs below that.  Semantics, flags and operand ranges
 * are those of the slab forms:
 *   ntt_inv_dot_dev_ptrs         c_p = inv( sum_{i<k} a_{i,p}^ (.) b_{i,p}^ ); h_ahat / h_bhat: HOST arrays of k device tables;
 *                                NTT_MUL_B_BROADCAST: h_bhat[i] is a device pointer to ONE polynomial (RNS: [limb][N]); k = 1: c's table
 *                                may be an operand's
 *   ntt_fwd_mul_dev_ptrs         c_p^ = fwd(a_p) (.) b_p^ (NTT_MUL_ACCUMULATE: += ...); a is SCRATCH: left as it was where a fused kernel
 *                                serves (up to 2^14; 2^15 in one pass), overwritten otherwise
 *   ntt_negacyclic_mul_dev_ptrs  c_p = a_p * b_p; a and b are SCRATCH: left as they were by the one-launch form up to 2^14 (FP64 plans),
 *                                overwritten otherwise; c's table may be a's or b's (the table itself, entry for entry -- not a
 *                                permutation of it); d_a == d_b squares ---- */
NTT_API int ntt_inv_dot_dev_ptrs(const ntt_plan *p, const uint64_t *const *d_c, int k, const uint64_t *const *const *h_ahat,
                                 const uint64_t *const *const *h_bhat, uint64_t count, unsigned flags, void *stream);
NTT_API int ntt_fwd_mul_dev_ptrs(const ntt_plan *p, const uint64_t *const *d_c, const uint64_t *const *d_a, const uint64_t *const *d_bhat,
                                 uint64_t count, unsigned flags, void *stream);
NTT_API int ntt_negacyclic_mul_dev_ptrs(const ntt_plan *p, const uint64_t *const *d_c, const uint64_t *const *d_a, const uint64_t *const *d_b,
                                        uint64_t count, void *stream);
NTT_API int ntt_rns_inv_dot_dev_ptrs(int nlimbs, ntt_plan *const *plans, const uint64_t *const *d_c, int k, const uint64_t *const *const *h_ahat,
                                     const uint64_t *const *const *h_bhat, uint64_t count, uint64_t limb_stride, unsigned flags, void *stream);
NTT_API int ntt_rns_fwd_mul_dev_ptrs(int nlimbs, ntt_plan *const *plans, const uint64_t *const *d_c, const uint64_t *const *d_a,
                                     const uint64_t *const *d_bhat, uint64_t count, uint64_t limb_stride, unsigned flags, void *stream);
NTT_API int ntt_rns_negacyclic_mul_dev_ptrs(int nlimbs, ntt_plan *const *plans, const uint64_t *const *d_c, const uint64_t *const *d_a,
                                            const uint64_t *const *d_b, uint64_t count, uint64_t limb_stride, void *stream);

/* ---- device memory / streams / timing (thin HIP wrappers for C callers) ---- */
NTT_API int ntt_dev_malloc(int device, void **d_ptr, size_t bytes);
NTT_API int ntt_dev_free(int device, void *d_ptr);
NTT_API int ntt_dev_mem_info(int device, size_t *free_bytes, size_t *total_bytes); /* hipMemGetInfo */
NTT_API int ntt_h2d(int device, void *d_dst, const void *h_src, size_t bytes);
NTT_API int ntt_d2h(int device, void *h_dst, const void *d_src, size_t bytes);
NTT_API int ntt_stream_create(int device, void **stream); /* hipStreamNonBlocking: does NOT synchronise with the null stream (ntt_h2d / ntt_d2h are blocking copies on the null stream: ntt_stream_sync first) */
NTT_API int ntt_stream_destroy(int device, void *stream);
NTT_API int ntt_stream_sync(int device, void *stream);
NTT_API int ntt_event_create(int device, void **event);
NTT_API int ntt_event_destroy(int device, void *event);
NTT_API int ntt_event_record(int device, void *event, void *stream);
NTT_API int ntt_event_elapsed_ms(int device, void *start, void *stop, float *ms); /* syncs stop */

/* ---- synthetic data and digests, device side ----
 * d_a[i] = splitmix64(seed ^ (offset + i)) mod q : the same function as the
 * oracle's orc_fill_uniform, so a host can regenerate any sampled polynomial */
NTT_API int ntt_fill_uniform(int device, uint64_t *d_a, uint64_t n, uint64_t q, uint64_t seed,
                             uint64_t offset, void *stream);
/* d_out[p] = sum_i splitmix64(i) * d_a[p*N+i]  (mod 2^64): position-sensitive
 * per-polynomial checksum for full-size parity checks */
NTT_API int ntt_poly_checksum(int device, uint64_t *d_out, const uint64_t *d_a, uint64_t N,
                              uint64_t batch, void *stream);
/* measured ceiling of the transform's memory shape (SURVEY 8d "fraction of a measured copy-kernel ceiling"):
 * every 16 bytes of d_a[0..n) are read, XORed with mask and written back in place by a plain grid-stride kernel
 * -- no arithmetic, no LDS, the same 8 B in + 8 B out per coefficient as an in-place NTT.  mask = 0 leaves the
 * data unchanged.  n must be even. */
NTT_API int ntt_rmw_probe(int device, uint64_t *d_a, uint64_t n, uint64_t mask, void *stream);
/* out-of-place copy of n words (16 bytes per lane, grid-stride): the copy shape the microarchitecture guide quotes the achievable
 * HBM rate for (about 6.3 TB/s of read + written bytes); 2 x 8 x n bytes move.  n must be even. */
NTT_API int ntt_copy_probe(int device, uint64_t *d_dst, const uint64_t *d_src, uint64_t n, void *stream);
/* the same measurement in the memory shape of the 2^14 block kernels themselves: one persistent 1024-thread workgroup per CU,
 * 2^14-word blocks as 16-byte loads with the next block prefetched in registers, XOR, 16-byte stores (whole KiB per wave and
 * instruction) -- the best memory-only skeleton of the transform kernels (profiles/r02/skeleton.txt), measured in the run
 * that quotes it.  n must be a multiple of 2^14. */
NTT_API int ntt_shape_probe(int device, uint64_t *d_a, uint64_t n, uint64_t mask, void *stream);

/* ---- multi-GPU: one call drives every listed device (per-device streams, no
 * collective -- polynomials are independent).  plans[g], d_a[g] and
 * batch[g] describe the shard resident on plans[g]'s device; the call returns
 * when all shards are done.  inverse != 0 selects the inverse transform. */
NTT_API int ntt_batch_multi(int ndev, ntt_plan *const *plans, uint64_t *const *d_a,
                            const uint64_t *batch, int inverse);
/* the same for RNS products c = a * b (BASELINE config 5 on several GPUs): plans[g * nlimbs + l] is limb l's plan on
 * shard g's device; d_c[g], d_a[g], d_b[g] are that shard's [limb][batch[g]][N] slabs (aliasing rules of
 * ntt_rns_negacyclic_mul_batch) */
NTT_API int ntt_rns_mul_multi(int ndev, int nlimbs, ntt_plan *const *plans, uint64_t *const *d_c, uint64_t *const *d_a,
                              uint64_t *const *d_b, const uint64_t *batch);

/* ---- reference-signature entry points: housekeeping ----
 * The single-polynomial functions of ntt_reference.h / ntt_radix4.h / ntt_radix4x4.h / ntt_seal.h keep
 * device tables for the caller tables they have seen (keyed on a hash of every entry, at most 32 plans,
 * least recently used evicted) and one staging buffer.  ntt_compat_release() frees all of it. */
NTT_API void ntt_compat_release(void);
NTT_API int  ntt_compat_cached_plans(void);

/* ---- parameter helpers (reference: SageMath script, tests/test_cases.h:113-142) ---- */
/* smallest primitive 2N-th root of unity mod q ("minimum root" rule); 0 if none */
NTT_API uint64_t ntt_min_root(uint64_t q, uint64_t N);
/* skip-th largest prime p < 2^bits with p = 1 (mod 2N); 0 if none */
NTT_API uint64_t ntt_find_prime(unsigned bits, uint64_t N, unsigned skip);

#ifdef __cplusplus
}
#endif
#endif /* NTT_MI355X_H */
