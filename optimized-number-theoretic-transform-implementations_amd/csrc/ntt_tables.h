/*
 * ntt_tables.h -- host-side construction of the device twiddle tables.
 *
 * Product code (no oracle involved).  Table semantics are the reference's:
 * slot k holds root^{bitrev_m(k)} mod q (include/internal/pre_compute.h:38-66)
 * and, for the integer policy, floor(w*2^64/q) (:68-77).  The FP64 policy
 * stores the balanced representative of the same residue and fl(w/q).
 */
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "ntt_arith.h"

namespace ntt {

using u128 = unsigned __int128;

inline uint64_t h_mulmod(uint64_t a, uint64_t b, uint64_t q) { return (uint64_t)(((u128)a * b) % q); }

inline uint64_t h_powmod(uint64_t b, uint64_t e, uint64_t q)
{
  uint64_t r = 1 % q;
  b %= q;
  for(; e; e >>= 1) {
    if(e & 1) r = h_mulmod(r, b, q);
    b = h_mulmod(b, b, q);
  }
  return r;
}

inline uint64_t h_bitrev(uint64_t v, unsigned bits)
{
  uint64_t r = 0;
  for(unsigned i = 0; i < bits; i++) r |= ((v >> i) & 1ULL) << (bits - 1 - i);
  return r;
}

inline unsigned h_log2(uint64_t n)
{
  unsigned m = 0;
  while((1ULL << m) < n) m++;
  return m;
}

inline uint64_t h_precon64(uint64_t w, uint64_t q) { return (uint64_t)(((u128)w << 64) / q); }

/* powers of `root` in bit-reversed slot order */
inline std::vector<uint64_t> h_power_table(uint64_t root, uint64_t N, uint64_t q)
{
  const unsigned        m = h_log2(N);
  std::vector<uint64_t> t(N);
  uint64_t              p = 1 % q;
  for(uint64_t i = 0; i < N; i++) {
    t[h_bitrev(i, m)] = p;
    p                 = h_mulmod(p, root, q);
  }
  return t;
}

/* inverse power table followed by 16 records N^-1 * winv[k], k < 16: the last inverse group's
 * twiddles with the scaling folded in (run_group0_folded reads them at slots N + k) */
inline std::vector<uint64_t> h_with_folded_ninv(const std::vector<uint64_t> &winv, uint64_t ninv, uint64_t q)
{
  std::vector<uint64_t> v = winv;
  for(size_t k = 0; k < 16; k++) v.push_back(k < winv.size() ? h_mulmod(ninv % q, winv[k], q) : ninv % q);
  return v;
}

inline TwU64 h_tw_u64(uint64_t w, uint64_t q) { return TwU64{w, h_precon64(w, q)}; }

/* the reference's 2N-entry radix-4 table from a radix-2 power table (pre_compute.h:85-105):
 * e[2k] = w[k];  e[4k+1] = w[k]*w[2k];  e[4k+3] = q - w[k]*w[2k+1]  (k < N/2) */
inline std::vector<uint64_t> h_expand_radix4(const std::vector<uint64_t> &w, uint64_t q)
{
  const size_t          n = w.size();
  std::vector<uint64_t> e(2 * n, 0);
  for(size_t k = 0; k < n; k++) e[2 * k] = w[k];
  for(size_t k = 1; k < n / 2; k++) { /* slots 1 and 3 stay 0 as in the reference (:90-93) */
    e[4 * k + 1] = h_mulmod(w[k], w[2 * k], q);
    e[4 * k + 3] = q - h_mulmod(w[k], w[2 * k + 1], q);
  }
  return e;
}

/* balanced representative and its quotient by q, correctly rounded to within
 * one long-double rounding (relative error < 2^-53 * (1 + 2^-10)) */
inline TwF64 h_tw_f64(uint64_t w, uint64_t q)
{
  const long double wb = (w > q / 2) ? -(long double)(q - w) : (long double)w;
  TwF64             t;
  t.w  = (double)wb;
  t.wq = (double)(wb / (long double)q);
  return t;
}

/* largest modulus the FP64 policy accepts, and its headroom class */
inline bool h_f64_eligible(uint64_t q) { return q <= ((1ULL << 51) + (1ULL << 41)); }

/* moduli the wide FP64 policy (ArithF64W) takes over from there: everything below 2^52 */
inline bool h_f64w_eligible(uint64_t q) { return q < (1ULL << 52); }

/* class ksh <=> q <= 2^(51-ksh) * (1 + 2^-10) */
inline int h_f64_ksh(uint64_t q)
{
  int k = 0;
  while(k < 40) {
    const long double lim = std::ldexp((long double)1.0, 51 - (k + 1)) * (1.0L + 1.0L / 1024.0L);
    if((long double)q <= lim) {
      k++;
    } else {
      break;
    }
  }
  return k;
}

inline ArithU64::consts h_consts_u64(uint64_t q, uint64_t N, const std::vector<uint64_t> &winv)
{
  ArithU64::consts c{};
  c.q  = q;
  c.q2 = 2 * q;
  const uint64_t ninv = h_powmod(N % q, q - 2, q);
  c.ninv              = h_tw_u64(ninv, q);
  /* merged last-stage twiddle N^-1 * winv[1]  (reference src/ntt_reference.c:55-61) */
  const uint64_t mw = winv.size() > 1 ? h_mulmod(ninv, winv[1], q) : ninv;
  c.wninv           = h_tw_u64(mw, q);
  c.r64             = h_tw_u64((uint64_t)((((u128)1) << 64) % q), q);
  c.one             = h_tw_u64(1 % q, q);
  /* Barrett constants of ArithU64X's products: bsh = bit length of q - 1, bmu = floor(2^(64+bsh) / q) (< 2^64: q > 2^bsh) */
  unsigned n = 0;
  while((q >> n) != 0) n++;
  c.bsh = n - 1;
  c.bmu = (uint64_t)((((u128)1) << (64 + c.bsh)) / q);
  return c;
}

inline F64Consts h_consts_f64(uint64_t q, uint64_t N, const std::vector<uint64_t> &winv)
{
  F64Consts c{};
  c.q      = (double)q;
  c.qinv   = (double)(1.0L / (long double)q);
  c.qinv_lo = (double)(1.0L / (long double)q - (long double)c.qinv);
  c.half_q = (double)(q / 2);
  c.qi     = q;
  {
    union {
      uint64_t u;
      double   d;
    } b;
    b.u      = 2 * q;
    c.q2_sub = b.d;
  }
  const uint64_t ninv = h_powmod(N % q, q - 2, q);
  c.ninv              = h_tw_f64(ninv, q);
  const uint64_t mw   = winv.size() > 1 ? h_mulmod(ninv, winv[1], q) : ninv;
  c.wninv             = h_tw_f64(mw, q);
  return c;
}

} /* namespace ntt */
