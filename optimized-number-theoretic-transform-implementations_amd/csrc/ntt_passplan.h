/*
 * ntt_passplan.h -- how a 2^m-point transform is cut into HBM passes.
 *
 *   6 <= m <= 14 : one fused pass (block = whole polynomial, one HBM round trip)
 *   m  > 14      : strided column passes of <= 4 stages over the leading
 *                  stages, then one fused pass over 2^14- or 2^12-point blocks
 *                  (two or more HBM round trips; reference sizes m = 15,16,17,
 *                  tests/test_cases.h:184-203).  The column pass is memory-bound
 *                  whatever its stage count, the block pass is bound by its FP64 work
 *                  (board power): at m = 15, 16 a 3- or 4-stage column pass over
 *                  2^12-point blocks measured 2-5 % faster than 1 or 2 stages over
 *                  2^14-point blocks (profiles/r02/ablations.txt); m = 17 would need
 *                  two column passes and keeps 2^14.
 *   m  < 6       : column passes only
 * The inverse runs the same passes in the opposite order.
 */
#pragma once
#include <cstdint>

namespace ntt {

constexpr int kFusedMin   = 6;
constexpr int kFusedMax   = 14;
constexpr int kFusedLarge = 14; /* block size used below column passes */
constexpr int kFusedSmallBlock = 12; /* ... and the alternative: more stages in the memory-bound column pass */

struct Pass {
  int fused; /* 1: fused block pass, 0: column pass            */
  int r;     /* fused: log2 block size; column: stages (1..4)  */
  int s;     /* first global stage covered                     */
};

struct PassList {
  int  n;
  Pass p[16];
};

/* block size below the column passes for a transform of 2^m points (m > kFusedMax) */
inline int multi_pass_block(int m, bool inverse, bool fp64)
{
  /* (integer policy: 2^12 blocks measured +5..7 % for the inverse, -3..-17 % for the forward transform) */
  if((m == 15 || m == 16) && (fp64 || inverse)) return kFusedSmallBlock;
  return kFusedLarge;
}

/* forward order; generic=true forces column passes only (cross-check path) */
inline PassList make_passes(int m, bool generic, int block_log = kFusedLarge)
{
  PassList L{};
  int      lead  = 0;
  int      block = 0;
  if(!generic && m >= kFusedMin && m <= kFusedMax) {
    block = m;
  } else if(!generic && m > kFusedMax) {
    block = block_log;
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

/* Radix-4 formulation (ArithU64R4) beyond one block: the reference pairs global stages (0,1), (2,3), ... and ends an odd
 * size with one radix-2 stage (src/ntt_radix4.c:33-61), so the column pass takes an EVEN number of leading stages (one or
 * two radix-4 levels) and the block pass -- which already ends odd-sized blocks with that radix-2 stage -- the rest:
 * m = 15: 2 + 13, m = 16: 2 + 14, m = 17: 4 + 13, m = 18: 4 + 14.  The inverse runs the same two passes in the opposite
 * order (:64-114: the blocks with the leading radix-2 stage of an odd size first, the column levels and the N^-1 pass last). */
constexpr int kRadix4Max = 18;
inline PassList make_passes_r4(int m)
{
  if(m <= kFusedMax) return make_passes(m, false);
  PassList  L{};
  const int lead = (m - kFusedMax + 1) & ~1;
  L.p[L.n++]     = Pass{0, lead, 0};
  L.p[L.n++]     = Pass{1, m - lead, lead};
  return L;
}

} /* namespace ntt */
