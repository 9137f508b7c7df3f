/*
 * ntt_passplan.h -- how a 2^m-point transform is cut into HBM passes.
 *
 *   6 <= m <= 14 : one fused pass (block = whole polynomial, one HBM round trip)
 *   m  > 14      : strided column passes of <= 4 stages over the leading
 *                  m-14 stages, then one fused pass over 2^14-point blocks
 *                  (measured 2-6 % faster than 2^13-point blocks)
 *                  (two or more HBM round trips; reference sizes m = 15,16,17,
 *                  tests/test_cases.h:184-203)
 *   m  < 6       : column passes only
 * The inverse runs the same passes in the opposite order.
 */
#pragma once
#include <cstdint>

namespace ntt {

constexpr int kFusedMin   = 6;
constexpr int kFusedMax   = 14;
constexpr int kFusedLarge = 14; /* block size used below column passes */

struct Pass {
  int fused; /* 1: fused block pass, 0: column pass            */
  int r;     /* fused: log2 block size; column: stages (1..4)  */
  int s;     /* first global stage covered                     */
};

struct PassList {
  int  n;
  Pass p[16];
};

/* forward order; generic=true forces column passes only (cross-check path) */
inline PassList make_passes(int m, bool generic)
{
  PassList L{};
  int      lead  = 0;
  int      block = 0;
  if(!generic && m >= kFusedMin && m <= kFusedMax) {
    block = m;
  } else if(!generic && m > kFusedMax) {
    block = kFusedLarge;
    lead  = m - block;
  } else {
    lead = m;
  }
  int s = 0;
  while(s < lead) {
    /* keep passes balanced: 5 = 3+2 rather than 4+1 */
    const int left = lead - s;
    int       r    = left > 4 ? (left >= 8 ? 4 : (left + 1) / 2) : left;
    L.p[L.n++]     = Pass{0, r, s};
    s += r;
  }
  if(block) L.p[L.n++] = Pass{1, block, s};
  return L;
}

} /* namespace ntt */
