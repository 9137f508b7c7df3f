/*
 * ref_shim.c -- exports the REAL reference implementation for oracle pinning.
 *
 * TEST INFRASTRUCTURE ONLY.  This translation unit is compiled by
 * oracle/Makefile together with the reference's own sources, which are read in
 * place from $(REF) (= /root/reference); nothing of the reference is copied into
 * this repository.  The reference keeps several entry points as header-inline
 * functions and owns its parameter table as a static array inside
 * tests/test_cases.h, so this shim gives them linkable names for ctypes.
 *
 * Output: oracle/_ref/libntt_ref.so (git-ignored, travels to the GPU box).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ntt_radix4.h"
#include "ntt_radix4x4.h"
#include "ntt_reference.h"
#include "ntt_seal.h"
#include "pre_compute.h"
#include "test_cases.h" /* reference tests/: tests[], _init_test */
#include "utils.h"      /* reference tests/: random_buf = rand()%q */

#define EXPORT __attribute__((visibility("default")))

EXPORT int ref_num_cases(void) { return (int)NUM_OF_TEST_CASES; }

/* out = {m, q, w, w_inv, n_inv} */
EXPORT void ref_case_params(int i, uint64_t out[5])
{
  out[0] = tests[i].m;
  out[1] = tests[i].q;
  out[2] = tests[i].w;
  out[3] = tests[i].w_inv;
  out[4] = (uint64_t)tests[i].n_inv.op;
}

static int g_inited = 0;
EXPORT void ref_init(void)
{
  if(!g_inited) {
    init_test_cases();
    g_inited = 1;
  }
}

/* which: 0 w_powers 1 w_powers_con 2 w_inv_powers 3 w_inv_powers_con (N each)
 *        4 r4 5 r4_con 6 inv_r4 7 inv_r4_con (2N each) */
EXPORT const uint64_t *ref_case_table(int i, int which)
{
  ref_init();
  const test_case_t *t = &tests[i];
  switch(which) {
    case 0: return t->w_powers.ptr;
    case 1: return t->w_powers_con.ptr;
    case 2: return t->w_inv_powers.ptr;
    case 3: return t->w_inv_powers_con.ptr;
    case 4: return t->w_powers_r4.ptr;
    case 5: return t->w_powers_con_r4.ptr;
    case 6: return t->w_inv_powers_r4.ptr;
    case 7: return t->w_inv_powers_con_r4.ptr;
    default: return NULL;
  }
}

EXPORT uint64_t ref_case_ninv_con(int i)
{
  ref_init();
  return (uint64_t)tests[i].n_inv.con;
}

/* the reference's input stream: unseeded glibc rand() % q (tests/utils.h:12-17) */
EXPORT void ref_random_buf(uint64_t *a, uint64_t n, uint64_t q) { random_buf(a, n, q); }

/* variant: 0 radix-2 ref, 1 radix-4, 2 radix-4x4, 3 seal   (fully reduced) */
EXPORT void ref_fwd(int i, int variant, uint64_t *a)
{
  ref_init();
  const test_case_t *t = &tests[i];
  switch(variant) {
    case 0: fwd_ntt_ref_harvey(a, t->n, t->q, t->w_powers.ptr, t->w_powers_con.ptr); break;
    case 1: fwd_ntt_radix4(a, t->n, t->q, t->w_powers_r4.ptr, t->w_powers_con_r4.ptr); break;
    case 2: fwd_ntt_radix4x4(a, t->n, t->q, t->w_powers_r4.ptr, t->w_powers_con_r4.ptr); break;
    default: fwd_ntt_seal(a, t->n, t->q, t->w_powers.ptr, t->w_powers_con.ptr); break;
  }
}

/* variant: 0 radix-2 ref lazy (<4q), 1 radix-4 lazy (<8q), 2 radix-4x4 lazy (<8q) */
EXPORT void ref_fwd_lazy(int i, int variant, uint64_t *a)
{
  ref_init();
  const test_case_t *t = &tests[i];
  if(variant == 0) {
    fwd_ntt_ref_harvey_lazy(a, t->n, t->q, t->w_powers.ptr, t->w_powers_con.ptr);
  } else if(variant == 1) {
    fwd_ntt_radix4_lazy(a, t->n, t->q, t->w_powers_r4.ptr, t->w_powers_con_r4.ptr);
  } else {
    fwd_ntt_radix4x4_lazy(a, t->n, t->q, t->w_powers_r4.ptr, t->w_powers_con_r4.ptr);
  }
}

EXPORT void ref_fwd_dbl(int i, uint64_t *a, uint64_t *b)
{
  ref_init();
  const test_case_t *t = &tests[i];
  fwd_ntt_ref_harvey_dbl(a, b, t->n, t->q, t->w_powers.ptr, t->w_powers_con.ptr);
}

/* variant: 0 radix-2 ref, 1 radix-4, 3 seal */
EXPORT void ref_inv(int i, int variant, uint64_t *a)
{
  ref_init();
  const test_case_t *t = &tests[i];
  switch(variant) {
    case 0:
      inv_ntt_ref_harvey(a, t->n, t->q, t->n_inv, WORD_SIZE, t->w_inv_powers.ptr,
                         t->w_inv_powers_con.ptr);
      break;
    case 1:
      inv_ntt_radix4(a, t->n, t->q, t->n_inv, t->w_inv_powers_r4.ptr,
                     t->w_inv_powers_con_r4.ptr);
      break;
    default:
      inv_ntt_seal(a, t->n, t->q, (uint64_t)t->n_inv.op, (uint64_t)t->n_inv.con,
                   t->w_inv_powers.ptr, t->w_inv_powers_con.ptr);
      break;
  }
}

/* generic-parameter entry points (tables supplied by the caller), used to pin
 * the oracle on (q,N) pairs that tests[] does not hold and as the "reference"
 * CPU baseline of bench.py. */
EXPORT void ref_build_tables(uint64_t N, uint64_t m, uint64_t q, uint64_t root,
                             uint64_t *w, uint64_t *wcon, uint64_t *e, uint64_t *econ)
{
  calc_w(w, root, N, q, m);
  calc_w_con(wcon, w, N, q, WORD_SIZE);
  expand_w(e, w, N, q);
  calc_w_con(econ, e, 2 * N, q, WORD_SIZE);
}

EXPORT void ref_fwd_r2_generic(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *w,
                               const uint64_t *wcon)
{
  fwd_ntt_ref_harvey(a, N, q, w, wcon);
}

EXPORT void ref_fwd_r4_generic(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *e,
                               const uint64_t *econ)
{
  fwd_ntt_radix4(a, N, q, e, econ);
}

/* the unreduced words of fwd_ntt_radix4_lazy (variant 1) / fwd_ntt_radix4x4_lazy (variant 2) at any size */
EXPORT void ref_fwd_r4_lazy_generic(int variant, uint64_t *a, uint64_t N, uint64_t q, const uint64_t *e, const uint64_t *econ)
{
  if(variant == 2) fwd_ntt_radix4x4_lazy(a, N, q, e, econ);
  else fwd_ntt_radix4_lazy(a, N, q, e, econ);
}

EXPORT void ref_fwd_r4_batch(uint64_t *a, uint64_t batch, uint64_t N, uint64_t q,
                             const uint64_t *e, const uint64_t *econ)
{
  for(uint64_t p = 0; p < batch; p++) fwd_ntt_radix4(a + p * N, N, q, e, econ);
}

EXPORT void ref_inv_r2_generic(uint64_t *a, uint64_t N, uint64_t q, uint64_t ninv,
                               const uint64_t *winv, const uint64_t *winv_con)
{
  mul_op_t n;
  n.op  = ninv;
  n.con = calc_ninv_con(ninv, q, WORD_SIZE);
  inv_ntt_ref_harvey(a, N, q, n, WORD_SIZE, winv, winv_con);
}

EXPORT void ref_inv_r4_generic(uint64_t *a, uint64_t N, uint64_t q, uint64_t ninv,
                               const uint64_t *einv, const uint64_t *einv_con)
{
  mul_op_t n;
  n.op  = ninv;
  n.con = calc_ninv_con(ninv, q, WORD_SIZE);
  inv_ntt_radix4(a, N, q, n, einv, einv_con);
}

/* ---- CPU timing harness on the reference's own functions (bench.py cpu_baseline, kind "reference") ---- */
static inline void shim_inv_r4(uint64_t *a, uint64_t N, uint64_t q, uint64_t ninv, const uint64_t *einv,
                               const uint64_t *einv_con)
{
  mul_op_t n;
  n.op  = ninv;
  n.con = calc_ninv_con(ninv, q, WORD_SIZE);
  inv_ntt_radix4(a, N, q, n, einv, einv_con);
}
#define CB_PREFIX(name)                      ref_##name
#define CB_FWD(a, N, q, e, econ)             fwd_ntt_radix4(a, N, q, e, econ)
#define CB_INV(a, N, q, ninv, einv, einvcon) shim_inv_r4(a, N, q, ninv, einv, einvcon)
#include "cpu_bench.inc"
