/*
 * ntt_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY, see ntt_oracle.h).
 *
 * A from-scratch restatement of the reference algorithm; every routine names the
 * reference lines whose behaviour it reproduces (paths relative to
 * /root/reference).  Parity pinned by oracle/gen_golden.py against the compiled
 * reference (oracle/_ref/libntt_ref.so) and frozen in tests/golden/.
 */
#include "ntt_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* scalar arithmetic                                                   */
/* ------------------------------------------------------------------ */

/* include/internal/fast_mul_operators.h:15-33 -- one conditional subtract */
uint64_t orc_csub(uint64_t v, uint64_t bound) { return v < bound ? v : v - bound; }

/* include/internal/fast_mul_operators.h:25-43 -- the reduce_{2,4,8}q_to_q chains */
uint64_t orc_reduce_to_q(uint64_t v, uint64_t q, int k)
{
  if(k >= 8) v = orc_csub(v, 4 * q);
  if(k >= 4) v = orc_csub(v, 2 * q);
  return orc_csub(v, q);
}

/* include/internal/fast_mul_operators.h:49-54 -- Shoup/Harvey lazy product:
 * Q = floor(wcon*t / 2^64); result = w*t - Q*q (mod 2^64), lands in [0,2q). */
uint64_t orc_shoup_lazy(uint64_t w, uint64_t wcon, uint64_t t, uint64_t q)
{
  const uint64_t quot = (uint64_t)(((orc_u128)wcon * t) >> 64);
  return w * t - quot * q;
}

/* include/internal/fast_mul_operators.h:62-70 -- two products, one quotient */
uint64_t orc_shoup_dbl_lazy(uint64_t w1, uint64_t c1, uint64_t w2, uint64_t c2,
                            uint64_t t1, uint64_t t2, uint64_t q)
{
  const orc_u128 s    = (orc_u128)c1 * t1 + (orc_u128)c2 * t2;
  const uint64_t quot = (uint64_t)(s >> 64);
  return t1 * w1 + t2 * w2 - quot * q;
}

uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q)
{
  return (uint64_t)(((orc_u128)a * b) % q);
}

uint64_t orc_powmod(uint64_t b, uint64_t e, uint64_t q)
{
  uint64_t r = 1 % q;
  b %= q;
  while(e) {
    if(e & 1) r = orc_mulmod(r, b, q);
    b = orc_mulmod(b, b, q);
    e >>= 1;
  }
  return r;
}

uint64_t orc_invmod(uint64_t a, uint64_t q) { return orc_powmod(a, q - 2, q); }

/* ------------------------------------------------------------------ */
/* table builders                                                      */
/* ------------------------------------------------------------------ */

/* include/internal/pre_compute.h:16-26 */
uint64_t orc_bitrev(uint64_t idx, unsigned width)
{
  uint64_t r = 0;
  for(unsigned b = 0; b < width; b++) {
    r = (r << 1) | ((idx >> b) & 1);
  }
  return r;
}

static unsigned log2_exact(uint64_t n)
{
  unsigned m = 0;
  while((1ULL << m) < n) m++;
  return m;
}

/* include/internal/pre_compute.h:38-66 -- successive powers scattered to the
 * bit-reversed slot: out[bitrev_m(i)] = root^i. */
void orc_build_powers(uint64_t *out, uint64_t root, uint64_t N, uint64_t q)
{
  const unsigned m = log2_exact(N);
  uint64_t       p = 1;
  for(uint64_t i = 0; i < N; i++) {
    out[orc_bitrev(i, m)] = p;
    p                     = orc_mulmod(p, root, q);
  }
}

/* include/internal/pre_compute.h:68-83 */
uint64_t orc_precon1(uint64_t w, uint64_t q, unsigned word)
{
  return (uint64_t)(((orc_u128)w << word) / q);
}

void orc_build_precon(uint64_t *con, const uint64_t *w, uint64_t n, uint64_t q,
                      unsigned word)
{
  for(uint64_t i = 0; i < n; i++) con[i] = orc_precon1(w[i], q, word);
}

/* include/internal/pre_compute.h:85-105 -- for k>=1: e[2k]=w[k];
 * e[4k+1]=w[k]*w[2k]; e[4k+3]=q-w[k]*w[2k+1]; slots 1 and 3 are zero. */
void orc_expand_radix4(uint64_t *e, const uint64_t *w, uint64_t N, uint64_t q)
{
  for(uint64_t k = 0; k < N; k++) e[2 * k] = w[k];
  e[1] = 0;
  e[3] = 0;
  for(uint64_t k = 1; 4 * k + 3 < 2 * N; k++) {
    e[4 * k + 1] = orc_mulmod(w[k], w[2 * k], q);
    e[4 * k + 3] = q - orc_mulmod(w[k], w[2 * k + 1], q);
  }
}

/* ------------------------------------------------------------------ */
/* radix-2 Harvey path                                                 */
/* ------------------------------------------------------------------ */

/* src/ntt_reference.c:11-31 with the butterfly of
 * include/internal/fast_mul_operators.h:72-81.  Stage s has 2^s blocks of
 * 2*half elements; block b uses twiddle slot 2^s + b. */
void orc_fwd_r2_lazy(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *w,
                     const uint64_t *wcon)
{
  const uint64_t q2 = 2 * q;
  for(uint64_t blocks = 1, half = N / 2; blocks < N; blocks *= 2, half /= 2) {
    for(uint64_t b = 0; b < blocks; b++) {
      const uint64_t tw = w[blocks + b], tc = wcon[blocks + b];
      uint64_t *     lo = a + 2 * half * b;
      uint64_t *     hi = lo + half;
      for(uint64_t j = 0; j < half; j++) {
        const uint64_t x = orc_csub(lo[j], q2);
        const uint64_t t = orc_shoup_lazy(tw, tc, hi[j], q);
        lo[j]            = x + t;
        hi[j]            = x - t + q2;
      }
    }
  }
}

/* include/ntt_reference.h:19-31 */
void orc_fwd_r2(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *w,
                const uint64_t *wcon)
{
  orc_fwd_r2_lazy(a, N, q, w, wcon);
  for(uint64_t i = 0; i < N; i++) a[i] = orc_reduce_to_q(a[i], q, 4);
}

/* src/ntt_reference.c:33-66; butterflies fast_mul_operators.h:83-106.
 * Gentleman-Sande stages with growing span; the last stage multiplies both
 * outputs by N^-1 (the odd output through the merged twiddle ninv*w[1]). */
void orc_inv_r2(uint64_t *a, uint64_t N, uint64_t q, uint64_t ninv,
                uint64_t ninv_con, unsigned word, const uint64_t *winv,
                const uint64_t *winv_con)
{
  const uint64_t q2   = 2 * q;
  uint64_t       half = 1;
  for(uint64_t blocks = N / 2; blocks > 1; blocks /= 2, half *= 2) {
    for(uint64_t b = 0; b < blocks; b++) {
      const uint64_t tw = winv[blocks + b], tc = winv_con[blocks + b];
      uint64_t *     lo = a + 2 * half * b;
      uint64_t *     hi = lo + half;
      for(uint64_t j = 0; j < half; j++) {
        const uint64_t s = orc_csub(lo[j] + hi[j], q2);
        const uint64_t d = lo[j] - hi[j] + q2;
        lo[j]            = s;
        hi[j]            = orc_shoup_lazy(tw, tc, d, q);
      }
    }
  }
  /* final stage: blocks == 1, half == N/2 (:55-65) */
  const uint64_t mw  = orc_shoup_lazy(ninv, ninv_con, winv[1], q);
  const uint64_t mwc = orc_precon1(mw, q, word);
  for(uint64_t j = 0; j < half; j++) {
    const uint64_t s = a[j] + a[j + half];
    const uint64_t d = a[j] - a[j + half] + q2;
    a[j]             = orc_csub(orc_shoup_lazy(ninv, ninv_con, s, q), q);
    a[j + half]      = orc_csub(orc_shoup_lazy(mw, mwc, d, q), q);
  }
}

/* ------------------------------------------------------------------ */
/* radix-4 path                                                        */
/* ------------------------------------------------------------------ */

typedef struct {
  uint64_t w[5], c[5];
} r4pack;

/* src/ntt_radix4.c:7-25 -- pack for block index k=blocks+j:
 * {e[2k], e[4k], e[4k+1], e[4k+2], e[4k+3]} = {W1, W2, W1W2, W3, -W1W3}. */
static r4pack r4_pack(const uint64_t *e, const uint64_t *econ, uint64_t k)
{
  r4pack p;
  p.w[0] = e[2 * k];
  p.c[0] = econ[2 * k];
  for(int i = 0; i < 4; i++) {
    p.w[1 + i] = e[4 * k + i];
    p.c[1 + i] = econ[4 * k + i];
  }
  return p;
}

/* include/internal/fast_mul_operators.h:108-128 */
static void r4_fwd_bfly(uint64_t *x, uint64_t *y, uint64_t *z, uint64_t *t,
                        const r4pack *p, uint64_t q)
{
  const uint64_t y1 = orc_shoup_dbl_lazy(p->w[1], p->c[1], p->w[2], p->c[2], *y, *t, q);
  const uint64_t y2 = orc_shoup_dbl_lazy(p->w[3], p->c[3], p->w[4], p->c[4], *y, *t, q);
  const uint64_t t1 = orc_csub(*x, 4 * q);
  const uint64_t t2 = orc_shoup_lazy(p->w[0], p->c[0], *z, q);
  *x                = t1 + t2 + y1;
  *y                = t1 + t2 - y1 + 2 * q;
  *z                = t1 - t2 + y2 + 2 * q;
  *t                = t1 - t2 - y2 + 4 * q;
}

/* include/internal/fast_mul_operators.h:130-149 */
static void r4_inv_bfly(uint64_t *x, uint64_t *y, uint64_t *z, uint64_t *t,
                        const r4pack *p, uint64_t q)
{
  const uint64_t q4 = 4 * q;
  const uint64_t s0 = *z + *t, s1 = *x + *y;
  const uint64_t d2 = q4 + *x - *y, d3 = q4 + *z - *t;
  *x = orc_csub(orc_csub(s1 + s0, q4), 2 * q);
  *z = orc_csub(orc_shoup_lazy(p->w[0], p->c[0], q4 + s1 - s0, q), q);
  *y = orc_shoup_dbl_lazy(p->w[1], p->c[1], p->w[3], p->c[3], d2, d3, q);
  *t = orc_shoup_dbl_lazy(p->w[2], p->c[2], p->w[4], p->c[4], d2, d3, q);
}

static int log2_is_even(uint64_t N) { return (log2_exact(N) & 1) == 0; }

/* one layer of radix-4 butterflies: `blocks` groups of 4*span coefficients, group b with pack blocks+b
 * (src/ntt_radix4.c:36-47; the butterflies of a layer are independent of each other) */
static void r4_fwd_layer(uint64_t *a, uint64_t q, const uint64_t *e, const uint64_t *econ, uint64_t blocks, uint64_t span)
{
  for(uint64_t b = 0; b < blocks; b++) {
    const r4pack p    = r4_pack(e, econ, blocks + b);
    uint64_t *   base = a + 4 * span * b;
    for(uint64_t i = 0; i < span; i++) {
      r4_fwd_bfly(base + i, base + i + span, base + i + 2 * span, base + i + 3 * span, &p, q);
    }
  }
}

/* trailing radix-2 stage on neighbours, twiddle e[N+i] (src/ntt_radix4.c:56-61, src/ntt_radix4x4.c:83-90) */
static void r2_fwd_tail_layer(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *e, const uint64_t *econ)
{
  for(uint64_t i = 0; i < N; i += 2) {
    const uint64_t x = orc_csub(orc_csub(a[i], 4 * q), 2 * q);
    const uint64_t t = orc_shoup_lazy(e[N + i], econ[N + i], a[i + 1], q);
    a[i]             = x + t;
    a[i + 1]         = x - t + 2 * q;
  }
}

/* src/ntt_radix4.c:27-62 */
void orc_fwd_r4_lazy(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *e,
                     const uint64_t *econ)
{
  const int      even  = log2_is_even(N);
  const uint64_t bound = even ? N : N / 2;
  uint64_t       span  = N / 4;
  for(uint64_t blocks = 1; blocks < bound; blocks *= 4, span /= 4) r4_fwd_layer(a, q, e, econ, blocks, span);
  if(!even) r2_fwd_tail_layer(a, N, q, e, econ);
}

/* src/ntt_radix4x4.c:41-114.  The radix-16 steps (:53-78) are pairs of radix-4 layers whose butterflies run in a
 * cache-friendlier ORDER; values only depend on the layers, so they are restated layer by layer here.  What differs
 * from fwd_ntt_radix4_lazy is the remainder when log2 N = 4k+3 (:91-111): 4k stages as radix-16 steps, THEN one
 * radix-2 stage on distance-4 pairs, THEN the last radix-4 layer (ntt_radix4.c ends on the radix-2 stage), so the
 * lazy words differ although the residues agree.  That radix-2 loop (:95-104) reduces a[i] for its GROUP counter i
 * (not for the coefficients of group i): by the time group i runs, coefficient i >= 1 has already been through its
 * butterfly (it belongs to group i/8 < i), coefficient 0 has not.  Restated as: coefficient 0 is brought below 4q
 * before its butterfly, coefficients 1 .. N/8-1 after theirs -- the same words the sequential loop leaves. */
void orc_fwd_r4x4_lazy(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *e, const uint64_t *econ)
{
  const int      rem    = log2_exact(N) & 3; /* :27-39 */
  uint64_t       blocks = 1, span = N / 4;
  for(; blocks < (N >> rem); blocks *= 16, span /= 16) { /* :53 */
    r4_fwd_layer(a, q, e, econ, blocks, span);           /* :66-69, roots m + j      */
    r4_fwd_layer(a, q, e, econ, 4 * blocks, span / 4);   /* :71-74, roots 4m + 4j + x */
  }
  if(rem == 1) r2_fwd_tail_layer(a, N, q, e, econ); /* :83-90 */
  if(rem == 3) {                                    /* :91-104 */
    const uint64_t groups = N / 8;
    a[0]                  = orc_csub(a[0], 4 * q);
    for(uint64_t g = 0; g < groups; g++) {
      const uint64_t w = e[2 * (groups + g)], wc = econ[2 * (groups + g)];
      for(uint64_t j = 8 * g; j < 8 * g + 4; j++) {
        const uint64_t x = orc_csub(a[j], 2 * q); /* harvey_fwd_butterfly, fast_mul_operators.h:72-81 */
        const uint64_t t = orc_shoup_lazy(w, wc, a[j + 4], q);
        a[j]             = x + t;
        a[j + 4]         = x - t + 2 * q;
      }
    }
    for(uint64_t i = 1; i < groups; i++) a[i] = orc_csub(a[i], 4 * q);
  }
  if(rem >= 2) r4_fwd_layer(a, q, e, econ, N / 4, 1); /* :105-111 */
}

/* include/ntt_radix4.h:16-28 */
void orc_fwd_r4(uint64_t *a, uint64_t N, uint64_t q, const uint64_t *e,
                const uint64_t *econ)
{
  orc_fwd_r4_lazy(a, N, q, e, econ);
  for(uint64_t i = 0; i < N; i++) a[i] = orc_reduce_to_q(a[i], q, 8);
}

/* src/ntt_radix4.c:64-114 */
void orc_inv_r4(uint64_t *a, uint64_t N, uint64_t q, uint64_t ninv,
                uint64_t ninv_con, const uint64_t *einv, const uint64_t *einv_con)
{
  uint64_t span = 1, blocks = N;
  if(log2_is_even(N)) {
    for(uint64_t i = 0; i < N; i++) a[i] = orc_csub(orc_csub(a[i], 4 * q), 2 * q);
  } else {
    /* leading radix-2 GS stage (:85-93) */
    for(uint64_t i = 0; i < N; i += 2) {
      const uint64_t x = orc_csub(a[i], 4 * q);
      const uint64_t y = a[i + 1];
      a[i]             = orc_csub(x + y, 2 * q);
      a[i + 1] = orc_shoup_lazy(einv[N + i], einv_con[N + i], x - y + 2 * q, q);
    }
    blocks /= 2;
    span *= 2;
  }
  for(blocks /= 4; blocks > 0; blocks /= 4, span *= 4) {
    for(uint64_t b = 0; b < blocks; b++) {
      const r4pack p    = r4_pack(einv, einv_con, blocks + b);
      uint64_t *   base = a + 4 * span * b;
      for(uint64_t i = 0; i < span; i++) {
        r4_inv_bfly(base + i, base + i + span, base + i + 2 * span,
                    base + i + 3 * span, &p, q);
      }
    }
  }
  for(uint64_t i = 0; i < N; i++) {
    a[i] = orc_csub(orc_shoup_lazy(ninv, ninv_con, a[i], q), q);
  }
}

/* ------------------------------------------------------------------ */
/* definition check, pointwise product, schoolbook                     */
/* ------------------------------------------------------------------ */

/* out[bitrev_m(i)] = sum_j a_j * root^{(2i+1) j}. */
void orc_fwd_naive(uint64_t *out, const uint64_t *a, uint64_t N, uint64_t q,
                   uint64_t root)
{
  const unsigned m = log2_exact(N);
  for(uint64_t i = 0; i < N; i++) {
    const uint64_t step = orc_powmod(root, 2 * i + 1, q);
    uint64_t       acc = 0, p = 1;
    for(uint64_t j = 0; j < N; j++) {
      acc = (acc + orc_mulmod(a[j] % q, p, q)) % q;
      p   = orc_mulmod(p, step, q);
    }
    out[orc_bitrev(i, m)] = acc;
  }
}

void orc_pointwise(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n,
                   uint64_t q)
{
  for(uint64_t i = 0; i < n; i++) c[i] = orc_mulmod(a[i], b[i], q);
}

void orc_negacyclic_schoolbook(uint64_t *c, const uint64_t *a, const uint64_t *b,
                               uint64_t N, uint64_t q)
{
  for(uint64_t k = 0; k < N; k++) c[k] = 0;
  for(uint64_t i = 0; i < N; i++) {
    for(uint64_t j = 0; j < N; j++) {
      const uint64_t p = orc_mulmod(a[i], b[j], q);
      const uint64_t k = i + j;
      if(k < N) {
        c[k] = (c[k] + p) % q;
      } else {
        c[k - N] = (c[k - N] + q - p) % q;
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* parameter generation                                                */
/* ------------------------------------------------------------------ */

int orc_is_prime(uint64_t n)
{
  static const uint64_t bases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  if(n < 2) return 0;
  for(size_t i = 0; i < sizeof(bases) / sizeof(bases[0]); i++) {
    if(n % bases[i] == 0) return n == bases[i];
  }
  uint64_t d = n - 1;
  unsigned r = 0;
  while((d & 1) == 0) {
    d >>= 1;
    r++;
  }
  for(size_t i = 0; i < sizeof(bases) / sizeof(bases[0]); i++) {
    uint64_t x = orc_powmod(bases[i], d, n);
    if(x == 1 || x == n - 1) continue;
    unsigned j = 1;
    for(; j < r; j++) {
      x = orc_mulmod(x, x, n);
      if(x == n - 1) break;
    }
    if(j == r) return 0;
  }
  return 1;
}

/* tests/test_cases.h:113-142: take any primitive 2N-th root g, walk its odd
 * powers g^(2i+1) (all primitive 2N-th roots) and keep the numerically
 * smallest. */
uint64_t orc_min_root(uint64_t q, uint64_t N)
{
  if((q - 1) % (2 * N) != 0 || !orc_is_prime(q)) return 0;
  const uint64_t cof = (q - 1) / (2 * N);
  uint64_t       g   = 0;
  for(uint64_t x = 2; x < q && x < 100000; x++) {
    const uint64_t c = orc_powmod(x, cof, q);
    if(orc_powmod(c, N, q) == q - 1) { /* order exactly 2N */
      g = c;
      break;
    }
  }
  if(!g) return 0;
  const uint64_t g2 = orc_mulmod(g, g, q);
  uint64_t       best = g, cur = g;
  for(uint64_t i = 0; i < N; i++) {
    if(cur < best) best = cur;
    cur = orc_mulmod(cur, g2, q);
  }
  return best;
}

uint64_t orc_find_prime(unsigned bits, uint64_t N, unsigned skip)
{
  const uint64_t step = 2 * N;
  uint64_t       p    = ((((uint64_t)1 << bits) - 1) / step) * step + 1;
  for(; p > step; p -= step) {
    if(orc_is_prime(p)) {
      if(skip == 0) return p;
      skip--;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------ */
/* inputs and digests                                                  */
/* ------------------------------------------------------------------ */

uint64_t orc_splitmix64(uint64_t x)
{
  x += 0x9e3779b97f4a7c15ULL;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
  return x ^ (x >> 31);
}

void orc_fill_uniform(uint64_t *a, uint64_t n, uint64_t q, uint64_t seed,
                      uint64_t offset)
{
  for(uint64_t i = 0; i < n; i++) a[i] = orc_splitmix64(seed ^ (offset + i)) % q;
}

uint64_t orc_fnv1a64(const uint64_t *a, uint64_t n)
{
  uint64_t h = 0xcbf29ce484222325ULL;
  for(uint64_t i = 0; i < n; i++) {
    for(int b = 0; b < 8; b++) {
      h ^= (a[i] >> (8 * b)) & 0xff;
      h *= 0x100000001b3ULL;
    }
  }
  return h;
}

/* ------------------------------------------------------------------ */
/* context                                                             */
/* ------------------------------------------------------------------ */

/* Mirrors the table set of tests/test_cases.h:212-251 (_init_test). */
orc_ctx *orc_ctx_new(uint64_t N, uint64_t q, uint64_t root)
{
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(*c));
  if(!c) return NULL;
  c->N        = N;
  c->q        = q;
  c->m        = log2_exact(N);
  c->root     = root;
  c->root_inv = orc_invmod(root, q);
  c->ninv     = orc_invmod(N % q, q);
  c->ninv_con = orc_precon1(c->ninv, q, 64);
  uint64_t *buf = (uint64_t *)malloc(sizeof(uint64_t) * 12 * N);
  if(!buf) {
    free(c);
    return NULL;
  }
  c->w        = buf;
  c->wcon     = buf + N;
  c->winv     = buf + 2 * N;
  c->winv_con = buf + 3 * N;
  c->e        = buf + 4 * N;
  c->econ     = buf + 6 * N;
  c->einv     = buf + 8 * N;
  c->einv_con = buf + 10 * N;
  orc_build_powers(c->w, root, N, q);
  orc_build_precon(c->wcon, c->w, N, q, 64);
  orc_build_powers(c->winv, c->root_inv, N, q);
  orc_build_precon(c->winv_con, c->winv, N, q, 64);
  orc_expand_radix4(c->e, c->w, N, q);
  orc_build_precon(c->econ, c->e, 2 * N, q, 64);
  orc_expand_radix4(c->einv, c->winv, N, q);
  orc_build_precon(c->einv_con, c->einv, 2 * N, q, 64);
  return c;
}

void orc_ctx_free(orc_ctx *c)
{
  if(!c) return;
  free(c->w);
  free(c);
}

void orc_fwd_r4_batch(uint64_t *a, uint64_t batch, const orc_ctx *c)
{
  for(uint64_t p = 0; p < batch; p++) orc_fwd_r4(a + p * c->N, c->N, c->q, c->e, c->econ);
}

void orc_fwd_r2_batch(uint64_t *a, uint64_t batch, const orc_ctx *c)
{
  for(uint64_t p = 0; p < batch; p++) orc_fwd_r2(a + p * c->N, c->N, c->q, c->w, c->wcon);
}

void orc_inv_r2_batch(uint64_t *a, uint64_t batch, const orc_ctx *c)
{
  for(uint64_t p = 0; p < batch; p++) {
    orc_inv_r2(a + p * c->N, c->N, c->q, c->ninv, c->ninv_con, 64, c->winv,
               c->winv_con);
  }
}

void orc_inv_r4_batch(uint64_t *a, uint64_t batch, const orc_ctx *c)
{
  for(uint64_t p = 0; p < batch; p++) {
    orc_inv_r4(a + p * c->N, c->N, c->q, c->ninv, c->ninv_con, c->einv, c->einv_con);
  }
}

/* ---- CPU timing harness on the restatement (bench.py cpu_baseline, kind "port": used only when oracle/_ref did
 * not travel) ---- */
static inline void orc_cb_inv_r4(uint64_t *a, uint64_t N, uint64_t q, uint64_t ninv, const uint64_t *einv,
                                 const uint64_t *einv_con)
{
  orc_inv_r4(a, N, q, ninv, orc_precon1(ninv, q, 64), einv, einv_con);
}
#define CB_PREFIX(name)                      orc_##name
#define CB_FWD(a, N, q, e, econ)             orc_fwd_r4(a, N, q, e, econ)
#define CB_INV(a, N, q, ninv, einv, einvcon) orc_cb_inv_r4(a, N, q, ninv, einv, einvcon)
#include "cpu_bench.inc"
