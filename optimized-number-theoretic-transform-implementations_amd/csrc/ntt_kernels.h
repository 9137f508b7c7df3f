/*
 * ntt_kernels.h -- gfx950 kernels built from the templates of ntt_core.h.
 *
 * fused_kernel : one workgroup transforms whole 2^LOGN blocks.  16 coefficients
 *                per thread live in VGPRs for up to four stages at a time; the
 *                block crosses LDS once per stage group and HBM exactly twice
 *                (one coalesced read, one coalesced write): 16 bytes of HBM
 *                traffic per coefficient per transform, the algorithmic minimum
 *                (SURVEY 8d).  No MFMA: 53/64-bit modular butterflies are
 *                element-wise VALU work.
 * column_kernel: strided passes for the leading stages of N > 2^14 and for tiny N.
 * fused_product_kernel: c = a * b in Z_q[X]/(X^N+1) with b never leaving the CU: forward transform of b, product
 *                with a^ in registers, inverse transform (N = 2^14 whole polynomials; block by block for
 *                N = 2^15..2^17); the inverse half reads the forward LDS twiddle table in mirrored order.
 * twophase_kernel: both passes of a 2^16 / 2^17 transform inside one workgroup (optional; fabric-bound, see
 *                DESIGN.md section 3).
 * team_kernel  : both passes of a 2^15 .. 2^17 transform as ITEMS of one persistent launch: per-XCD in-order queues
 *                (the XCD is read from HW_REG_XCC_ID), per-polynomial hand-off counters, the intermediate kept in the
 *                XCD's L2 / the Infinity Cache; team_product_kernel: the same scheme with three item kinds for a whole
 *                product (column stages of both operands, block products -- both blocks through their block stages,
 *                product, inverse block stages --, inverse column stages; or, a^ given, the b-chain only).
 * MULTI        : kernel variants that serve several RNS limbs in one launch (a LimbRec per limb in the kernel arguments).
 * The same kernels serve four arithmetic policies (ntt_arith.h): FP64 with a reduction schedule, FP64 for moduli up
 * to 2^52, the reference's integer radix-2 arithmetic and its radix-4 formulation.
 *
 * Launch geometry (wave64, 256 CUs): LOGN=14 -> one persistent 1024-thread
 * workgroup per CU (16 waves, 4 per SIMD, <=128 VGPRs, no scratch) using 158 KiB
 * of the CU's 160 KiB LDS (128.1 KiB exchange buffer + 30 KiB twiddle table);
 * 2^13 -> one 512-thread workgroup (128 KiB: every per-lane twiddle in LDS),
 * 2^12 -> four 256-thread workgroups (39.6 KiB each).  Blocks >= 2^12 run the
 * persistent loops below (register prefetch of the next block); smaller blocks
 * pack several per 256-thread workgroup, share LDS twiddle tables from 2^8 up and
 * rely on multiple resident workgroups instead of a prefetch.
 * Tuning history and rejected variants: profiles/r01/ablations.txt, profiles/r02/ablations.txt.
 */
#pragma once
#include <hip/hip_runtime.h>

#include "ntt_core.h"
#include "ntt_passplan.h"

namespace ntt {

/* a < b for block counts (both far below 2^63) as a subtraction and a sign test: the 64-bit unsigned comparison has no scalar
 * instruction, so the compiler copies b into two VGPRs for the whole kernel (v_cmp_lt_u64) -- in the 52-bit class's one-launch
 * product at 2^14 those were the two registers that spilled */
__device__ __forceinline__ bool below(uint64_t a, uint64_t b) { return (int64_t)(a - b) < 0; }
constexpr int kPreAlso = 12; /* forward: last group's twiddles register-resident at this size too (2^14 always) */
constexpr int kIpreMin = 12; /* inverse: first executed group's twiddles register-resident from this size up */

/* FLAVOR: 0 integer radix-2, 1 FP64 (compact twiddles, LDS tables, persistent inverse), 2 integer radix-4 (five-record
 * twiddle packs and 128-bit double products: it gets the register budget of two waves per SIMD where the
 * workgroup size allows) */
template <class A> constexpr int flavor_of() { return A::kCompact ? 1 : (A::kRadix4 ? 2 : 0); }

/* FLAVOR: 0 integer radix-2, 1 FP64 (compact twiddles), 2 integer radix-4, 3 FP64 inside fused_product_kernel (one block
 * per workgroup at every size: the two transforms of a product leave no registers for a second block's prefetch) */
template <int LOGN, bool INV, int FLAVOR> struct Geom {
  static constexpr bool COMPACT = FLAVOR == 1 || FLAVOR == 3;
  using P = Plan<LOGN>;
  /* one plan thread per hardware thread.  (Two per lane -- 512-thread workgroups with 256
   * VGPRs -- was measured at 14.5 vs 16.0 M NTT/s and removed: four waves per SIMD hide LDS
   * and L2 latencies better.) */
  /* 2^13 (FP64 policies): TWO blocks per 1024-thread workgroup, each half running the persistent loop on its own
   * exchange buffer and sharing the twiddle table -- the resource shape of the 2^14 kernel (16 waves per CU,
   * 158 KB of LDS).  One 512-thread workgroup per CU with every table in LDS (round 1) left two waves per SIMD.
   * Round 6 measured the alternative the review of round 5 proposed -- two INDEPENDENT 512-thread workgroups per CU (own barriers,
   * a table of stages 8..10 each, stage 11's twiddles from the L2 because two full tables do not fit, 1..16 workgroups per
   * resident slot) -- against this shape on one box, alternating: 0.570 (1 per slot) .. 0.590 (8 per slot) against 0.599-0.600 for
   * the lock-stepped halves: the halves' common barriers are not what holds 2^13 back (profiles/r06/ab_2p13_shapes.txt; the code was
   * removed again).  The half is a property of the wave: `sub` is made uniform, so block indices and addresses stay scalar. */
  static constexpr bool PERSIST2 = FLAVOR == 1 && LOGN == 13 && !INV;
  /* (inverse: measured 0.584 -> 0.48 in round 2 and again in round 3 (profiles/r03/ablations.txt) -- with the stage-12
   * twiddles register-resident the kernel needs 133 VGPRs (5 spilled); requested per block it fits in 122 without a
   * spill and is still 18 % slower: the two exchange buffers leave 1.8 KB of LDS, 128 bytes short of even the 1.9 KB
   * table of stages 4..7, so 23 per-lane twiddles per thread and block come from global memory where the one-block
   * shape reads all three tables (2 + 30 + 32 KB) from LDS) */
  static constexpr int WG  = PERSIST2 ? 1024 : (P::T < 256 ? 256 : P::T);      /* threads per workgroup */
  static constexpr int BPW = PERSIST2 ? 2 : (P::T < 256 ? 256 / P::T : 1);    /* blocks per workgroup  */
  static constexpr bool PERSISTENT = BPW == 1 || PERSIST2;                     /* persistent prefetching loops */
  /* Compact twiddles kept in LDS for the whole launch, one table per stage group whose stages
   * are all per-lane (entries; 0 = not used).  Which groups get one is a footprint decision:
   *   2^14: the second-to-last group (stages 8..11, 3840 doubles = 30 KB next to the 128.1 KB
   *         exchange buffer); the last group's 12288 entries do not fit and are preloaded;
   *   2^13: inverse: every per-lane group (stages 4..7, 8..11 and 12: 2 + 30 + 32 KB): one 130 KB
   *         workgroup per CU and no global twiddle loads at all; forward (two blocks per
   *         workgroup, PERSIST2): as 2^14 -- the second-to-last group's table, the last group's
   *         twiddles register-resident (measured +1.5 % over the inverse's scheme);
   *   2^12: the second-to-last group only (7.7 KB: 4 workgroups per CU, measured +10 %; the
   *         last group's 24 KB would halve the resident workgroups);
   *   2^8..2^11: every per-lane group (at most 16 KB per 256-thread workgroup of 2..64
   *         blocks): these sizes were texture-addresser-bound on their 27-31 per-lane global
   *         twiddle loads per thread (TA 91 % busy). */
  static constexpr bool group_is_per_lane(int g)
  {
    for(int j = 0; j < P::R(g); j++)
      if(P::TW_UNIFORM(g, j)) return false;
    return true;
  }
  static constexpr int TBL(int g)
  {
    if(!COMPACT || g < 0 || g >= P::NG || !group_is_per_lane(g)) return 0;
    bool on = false;
    if(LOGN == 14 || LOGN == 12 || (LOGN == 13 && !INV)) on = (g == P::NG - 2);
    if(LOGN >= 8 && LOGN <= 11) on = true; /* several blocks per workgroup share the tables (2^6, 2^7: measured no gain) */
    if(LOGN == 13 && INV) on = true;
    return on ? (((1 << P::R(g)) - 1) << P::S(g)) : 0;
  }
  /* first entry of group g's table behind the exchange buffer(s) */
  static constexpr int TBL_OFF(int g)
  {
    int o = 0;
    for(int h = 0; h < g; h++) o += TBL(h);
    return o;
  }
  static constexpr int LDS_TW = TBL_OFF(P::NG);
  static constexpr int LDS_BYTES  = (BPW * P::LDS_ELEMS + LDS_TW) * 8;
  static constexpr int WG_PER_CU0 = 163840 / LDS_BYTES;
  /* waves per SIMD the register allocator may assume (VGPR budget 512/x): what
   * the LDS footprint lets be resident, at most 4 */
  static constexpr int WPS0 = (WG_PER_CU0 * (WG / 64)) / 4;
  /* (FLAVOR 3 below 2^12: the product kernel keeps a^ (32 VGPRs) beside the transform's registers -- three
   * workgroups per CU, 170 VGPRs, as the LDS footprint dictates at 2^10 and 2^11 anyway; four would spill 2-5) */
  static constexpr int WPSC = FLAVOR == 2 ? (WG / 256 > 2 ? WG / 256 : 2) : (FLAVOR == 3 && LOGN < 12 ? 3 : 4);
  static constexpr int WPS  = WPS0 < 1 ? 1 : (WPS0 > WPSC ? WPSC : WPS0);
};

/* ------------------------------------------------------------------ */
/* kernel arguments: one launch may serve several RNS limbs             */
/* ------------------------------------------------------------------ */
/*
 * A launch carries one LimbRec per limb in its kernel arguments: one for an ordinary call, up to kMaxLimbs for an RNS set
 * (ntt_rns_*: limbs laid out [limb][batch][N], every limb its own prime, tables and constants -- SURVEY 8e).
 * Workgroup g of a launch serves limb g / wgs_per_limb and is block-id g % wgs_per_limb of that limb's share of the grid,
 * so a persistent workgroup never changes limb (its LDS tables and register-resident twiddles stay valid) and ONE launch
 * fills the chip even when a single limb's batch is a handful of polynomials (a ciphertext: few polynomials x tens of
 * primes).  The kernels below build their Params from their limb's record and are otherwise unchanged.
 */
template <class A> struct LimbRec {
  const typename A::tw * tw_f;  /* forward records                                   */
  const typename A::ctw *tw8_f; /* forward compact (FP64 policies; null otherwise)   */
  const typename A::tw * tw_i;  /* inverse records (+16 folded N^-1 records)         */
  const typename A::ctw *tw8_i;
  typename A::consts     c;
};

constexpr int kMaxLimbs = 16; /* limbs of one launch (16 records of 112 bytes in the 4 KiB kernel-argument segment); larger sets take several launches */

template <class A> struct KArgs {
  uint64_t *        a;            /* limb 0's coefficients                                        */
  uint64_t          limb_stride;  /* words between the slabs of consecutive limbs                 */
  uint64_t          poly_stride;  /* words between consecutive polynomials of one limb (N: the dense [batch][N] slab; a caller
                                   * that keeps [polynomial][limb][N] passes limb_stride = N, poly_stride = limbs * N) */
  uint32_t          wgs_per_limb; /* grid = wgs_per_limb * limbs                                  */
  uint32_t          logn, s0, wide, lastinv, lazy;
  uint64_t          nblocks;      /* per limb                                                     */
  const uint64_t *  ptab;         /* transform launches over a pointer batch: device table of per-polynomial word offsets from `a`
                                   * (ntt_core.h poly_offset; a = null, so an entry is address / 8); null otherwise */
  /* one record per limb, IN the kernel-argument segment: its loads are kernarg-relative scalar loads like those of a
   * single set of tables (the compiler re-loads them at will instead of holding or spilling them -- a table in global
   * memory cost the 2^14 inverse kernel 3-4 spilled VGPRs) */
  LimbRec<A>        limbs[kMaxLimbs];
};

/* the launch's Params for this workgroup's limb; bid = its block id inside the limb's share of the grid */
/* MULTI is a compile-time property of the kernel: with it off the limb is 0, every record field sits at a fixed offset of
 * the kernel-argument segment (the compiler re-loads such values at will instead of keeping them in registers) and the
 * kernel is the single-set kernel it always was; a run-time limb index costs the register-tight kernels 2-4 spilled VGPRs
 * (measured: the 2^14 inverse, the 2^13 forward), which is why the host only uses the MULTI variants when one limb's share
 * alone cannot fill the chip. */
template <class A, bool INV, bool MULTI>
__device__ __forceinline__ Params<A> limb_params(const KArgs<A> &k, uint32_t &bid, uint32_t &gdim, uint32_t &limb)
{
  if constexpr(MULTI) {
    /* A two-dimensional grid: blockIdx.y is the limb, blockIdx.x the block id inside the limb's share -- both arrive in scalar
     * registers.  Rounds 3 and 4 launched a flat grid and computed limb = blockIdx.x / wgs_per_limb: a 32-bit division by a
     * run-time value is expanded into VALU float instructions, its result lives in a VGPR, and everything derived from it --
     * the block id and the loop counters of the persistent loops, the limb's record address, the slab pointer, every buffer
     * descriptor (then a "waterfall" loop around each buffer_load) -- followed it there (a readfirstlane on a provably uniform
     * value is folded away by the compiler): that, not the run-time index as such, was what these variants spilled on. */
    limb = blockIdx.y;
    bid  = blockIdx.x;
    gdim = gridDim.x;
  } else {
    limb = 0;
    bid  = blockIdx.x;
    gdim = gridDim.x;
  }
  const LimbRec<A> &r = k.limbs[limb];
  Params<A>         p;
  p.a       = k.a + (uint64_t)limb * k.limb_stride;
  p.tw      = INV ? r.tw_i : r.tw_f;
  p.tw8     = INV ? r.tw8_i : r.tw8_f;
  p.c       = r.c;
  p.logn    = k.logn;
  p.s0      = k.s0;
  p.wide    = k.wide;
  p.lastinv = k.lastinv;
  p.lazy    = k.lazy;
  p.nblocks = k.nblocks;
  p.pstride = k.poly_stride;
  p.ptab    = k.ptab;
  return p;
}

/* word offset of block b of this launch (ntt_core.h block_offset: polynomial b >> s0 starts (b >> s0) * pstride words in) */
template <int LOGN, class A> __device__ __forceinline__ uint64_t blk_off(const Params<A> &p, uint64_t b)
{
  return block_offset<LOGN>(b, p.s0, p.pstride);
}
/* the same for the transform kernels, which also serve pointer batches (Params::ptab: the polynomial's start comes from a table) */
template <int LOGN, class A> __device__ __forceinline__ uint64_t blk_off_t(const Params<A> &p, uint64_t b)
{
  return block_offset<LOGN>(b, p.s0, p.pstride, p.ptab);
}


__device__ __forceinline__ void wave_sync()
{
  /* LDS operations of one wave execute in issue order; only the compiler has to
   * be told not to move accesses across the exchange */
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};

/* between(): runs between the two barriers of a cross-wave exchange -- every wave has left the previous stage
 * groups, nobody has entered the next ones: the place to rewrite LDS data shared by the whole workgroup
 * (the two-phase kernel refreshes its twiddle table there at no extra barrier) */
template <class A, int LOGN, int GW, int GR, class HOOK = NoHook>
__device__ __forceinline__ void exchange(typename A::val (&x)[kE], uint32_t t, typename A::val *lds, HOOK between = HOOK())
{
  using P = Plan<LOGN>;
  constexpr bool local = P::WAVE_LOCAL(GW, GR);
  if constexpr(local) {
    lds_scatter<A, LOGN, GW, GR>(x, t, lds);
    wave_sync();
    lds_gather<A, LOGN, GW, GR>(x, t, lds);
    wave_sync();
  } else {
    __syncthreads(); /* every wave has finished reading the previous layout */
    lds_scatter<A, LOGN, GW, GR>(x, t, lds);
    between();
    __syncthreads();
    lds_gather<A, LOGN, GW, GR>(x, t, lds);
  }
}

#ifndef NTT_LOAD_AUX
#  define NTT_LOAD_AUX 2 /* (A/B builds: tools/build_tu_variant.sh) */
#endif
constexpr int kLoadAux = NTT_LOAD_AUX; /* cache-policy bits of the coefficient loads: nt (measured +0.6..1 % over 0; sc0/sc1 no gain) */
/* A block seen through a buffer descriptor: the 16 row loads of a thread then share ONE
 * 32-bit lane offset (t*8) and take the row offset as a scalar operand, instead of a
 * 64-bit per-lane address each (two carry-chained VALU adds per row in the hot loop).
 * blk is wave-uniform (derived from blockIdx and the loop counter only). */
template <int LOGN> __device__ __forceinline__ __amdgpu_buffer_rsrc_t block_rsrc(const uint64_t *blk, bool live = true)
{
  /* live = false: a descriptor of zero records -- every load through it is out of range, returns 0 and moves no
   * data.  The persistent loops prefetch unconditionally (a branch around the prefetch costs registers); in a
   * workgroup's last iteration the descriptor is dead instead of re-reading a block (1/8 of the reads of a
   * 256 MiB chunk of a multi-pass transform, where a workgroup only sees 8 blocks per launch). */
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(blk), /*stride*/ 0, live ? (int)(8u << LOGN) : 0, 0x00020000);
}
/* cache-policy bits of the buffer instructions (gfx940+): 1 = sc0, 2 = nt, 16 = sc1 */
constexpr int kAuxNt = 2, kAuxSc1 = 16, kAuxSc0Sc1 = 17;
template <int AUX = kLoadAux> __device__ __forceinline__ uint64_t buffer_load_u64(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
  const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, AUX);
  return (uint64_t)v.x | ((uint64_t)v.y << 32);
}
template <int AUX = kLoadAux> __device__ __forceinline__ u64x2 buffer_load_u64x2(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
  const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, AUX);
  return u64x2{(uint64_t)v.x | ((uint64_t)v.y << 32), (uint64_t)v.z | ((uint64_t)v.w << 32)};
}

/* the inverse loop's final stores (slot e <-> index (e << LT) + t, 8 bytes per lane) through
 * the block descriptor: one lane offset, the row offset as a scalar operand */
template <int LOGN, int AUX = 0>
__device__ __forceinline__ void buffer_store_first_raw(const uint64_t (&u)[kE], uint32_t t, uint64_t *blk)
{
  using P                        = Plan<LOGN>;
  const __amdgpu_buffer_rsrc_t r = block_rsrc<LOGN>(blk);
  typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
  static_for<0, kE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    v2u32         v;
    v.x = (unsigned)u[E];
    v.y = (unsigned)(u[E] >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(v, r, (int)(t * 8u), (int)(((uint32_t)E << P::LT) * 8u), AUX);
  });
}

/* raw (unconverted) coefficients of the first-kind group: slot e <-> (e << LT) + t */
template <int LOGN, int AUX = kLoadAux>
__device__ __forceinline__ void prefetch_first(uint64_t (&raw)[kE], uint32_t t, const uint64_t *blk, bool live = true)
{
  using P = Plan<LOGN>;
  const __amdgpu_buffer_rsrc_t r = block_rsrc<LOGN>(blk, live);
  static_for<0, kE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    raw[E]          = buffer_load_u64<AUX>(r, t * 8u, ((uint32_t)E << P::LT) * 8u);
  });
}

/* raw coefficients in the last-kind layout (runs of 2^RL consecutive indices, 16-byte loads):
 * what the inverse transform's first group consumes */
template <int LOGN, int AUX = kLoadAux>
__device__ __forceinline__ void prefetch_last(uint64_t (&raw)[kE], uint32_t t, const uint64_t *blk, bool live = true)
{
  using P           = Plan<LOGN>;
  constexpr int G   = P::NG - 1;
  const uint32_t ib = P::IBASE(G, t);
  const __amdgpu_buffer_rsrc_t r = block_rsrc<LOGN>(blk, live);
  static_for<0, kE / 2>([&](auto hh) {
    constexpr int E = 2 * decltype(hh)::value;
    const u64x2   v = buffer_load_u64x2<AUX>(r, ib * 8u, P::IOFF(G, E) * 8u);
    raw[E]          = v.a;
    raw[E + 1]      = v.b;
  });
}

/* Final stores of the forward transform as WHOLE LINES.  In the last group a lane owns runs of four consecutive
 * coefficients (32 bytes): a 16-byte store instruction then writes half of every 32 bytes it touches -- 64
 * half-filled chunks over 2 KiB.  One v_permlane32_swap per dword exchanges slot bit 1 with lane bit 5 first:
 * lanes 0-31 then hold the even 16-byte chunks of a 1-KiB run and lanes 32-63 the odd ones, and every store
 * instruction of the wave covers one contiguous KiB (tools/skel.hip: 0.674 -> 0.694 of the HBM peak for the
 * memory skeleton).  This is the one place where a cross-lane move (north star: "wave64 shuffles") pays:
 * 16 single-issue VALU instructions per thread against 32 LDS operations for an LDS transpose. */
template <class A, int LOGN, bool LAZYT, int AUX = 0>
__device__ __forceinline__ void store_last_whole_lines(const typename A::val (&x)[kE], uint32_t t, uint64_t *blk,
                                                       const typename A::consts &c, bool lazy_rt)
{
  using P          = Plan<LOGN>;
  constexpr int G  = P::NG - 1;
  constexpr int HB = P::TB(G, 5); /* index bit held by lane bit 5 */
  static_assert(P::RL == 2 && P::NL == 6 && HB > 1, "needs runs of four coefficients per lane and full waves");
  uint64_t u[kE];
  static_for<0, kE>([&](auto ee) { u[decltype(ee)::value] = out_word<A, false, LAZYT>(x[decltype(ee)::value], lazy_rt, c); });
  /* slots E (bit 1 clear) and E|2: swap the upper-half lanes of the first with the lower-half lanes of the second */
  static_for<0, kE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    if constexpr((E & 2) == 0) {
      const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)u[E], (unsigned)u[E | 2], false, false);
      const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(u[E] >> 32), (unsigned)(u[E | 2] >> 32), false, false);
      u[E]          = (uint64_t)lo[0] | ((uint64_t)hi[0] << 32);
      u[E | 2]      = (uint64_t)lo[1] | ((uint64_t)hi[1] << 32);
    }
  });
  /* after the swap: slot bit 1 <-> index bit HB, lane bit 5 <-> index bit 1 */
  const uint32_t ib   = (P::IBASE(G, t) & ~(1u << HB)) | (((t >> 5) & 1u) << 1);
  const __amdgpu_buffer_rsrc_t r = block_rsrc<LOGN>(blk);
  typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
  static_for<0, kE / 2>([&](auto hh) {
    constexpr int      E   = 2 * decltype(hh)::value;
    constexpr uint32_t OFF = (P::IOFF(G, E) & ~2u) | ((uint32_t)((E >> 1) & 1) << HB);
    v4u32              v;
    v.x = (unsigned)u[E];
    v.y = (unsigned)(u[E] >> 32);
    v.z = (unsigned)u[E + 1];
    v.w = (unsigned)(u[E + 1] >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(ib * 8u), (int)(OFF * 8u), AUX);
  });
}

/* Makes the compiler complete the loads behind a prefetched block at this point.  Used
 * once, before a persistent loop is entered: the waits the compiler places inside the loop
 * are the merge of both ways into it, and anything still pending on the way in from the
 * prologue becomes an s_waitcnt vmcnt(n) that ALSO executes on every later iteration --
 * where the only pending operations are the previous block's stores, i.e. it would wait
 * for stores nobody needs. */
__device__ __forceinline__ void pin_raw(const uint64_t (&raw)[kE])
{
#pragma unroll
  for(int e = 0; e < kE; e++) asm volatile("" ::"v"(raw[e]));
}
/* makes the compiler complete the loads behind a preloaded twiddle set at this point */
template <class A, int LOGN, int G>
__device__ __forceinline__ void pin_preloaded(const typename A::ctw (&pre)[4][kE / 2])
{
  using P = Plan<LOGN>;
  /* plain unrolled loops: asm operands cannot name a reference captured by a lambda */
#pragma unroll
  for(int j = 0; j < P::R(G); j++) {
#pragma unroll
    for(int b = 0; b < kE / 2; b++) {
      if(P::BFLY_FIRST(G, j, b) == b) asm volatile("" ::"v"(pre[j][b]));
    }
  }
}

/* Fills the LDS twiddle tables of a workgroup (Geom::TBL).  A table depends on the block's
 * position inside its polynomial; a persistent workgroup keeps it for the whole launch, which
 * is valid because its stride over the blocks is a multiple of the blocks per polynomial
 * (launch_fused enforces it).  Stage J of group g is stored TRANSPOSED: slot
 * l = prefix * 2^J + u goes to (2^J - 1) * 2^S + u * 2^S + prefix (see load_stage_tw). */
template <class A, int LOGN, bool INV, class G = Geom<LOGN, INV, flavor_of<A>()>>
__device__ __forceinline__ void fill_lds_tables(typename A::ctw *tabl, const Params<A> &p, uint32_t blk0, uint32_t tid)
{
  using P = Plan<LOGN>;
  static_for<0, P::NG>([&](auto gg) {
    constexpr int GI = decltype(gg)::value;
    if constexpr(G::TBL(GI) > 0) {
      typename A::ctw *tg = tabl + G::TBL_OFF(GI);
      static_for<0, P::R(GI)>([&](auto jj) {
        constexpr int JJ  = decltype(jj)::value;
        constexpr int SG  = P::S(GI);
        constexpr int SLJ = SG + JJ;
        const typename A::ctw *src = p.tw8 + ((size_t)1 << (p.s0 + SLJ)) + ((size_t)blk0 << SLJ);
        for(uint32_t l = tid; l < (1u << SLJ); l += G::WG) {
          const uint32_t u = l & ((1u << JJ) - 1u), prefix = l >> JJ;
          tg[(((1u << JJ) - 1u) << SG) + (u << SG) + prefix] = src[l];
        }
      });
    }
  });
}

/* LAZY (forward, FP64 policy): outputs in [0,4q) instead of [0,q) -- a kernel variant of its own because
 * the reduction schedule has to bound the last stage (fused_mask); the integer policies take the run-time
 * flag Params::lazy instead. */
template <class A, int LOGN, bool INV, int KSH, bool LASTINV = false, bool LAZY = false, bool MULTI = false>
__global__ void __launch_bounds__((Geom<LOGN, INV, flavor_of<A>()>::WG), (Geom<LOGN, INV, flavor_of<A>()>::WPS)) fused_kernel(const KArgs<A> k)
{
  uint32_t        bid, gdim, limb_;
  const Params<A> p = limb_params<A, INV, MULTI>(k, bid, gdim, limb_);
  using P                 = Plan<LOGN>;
  using G                 = Geom<LOGN, INV, flavor_of<A>()>;
  static_assert(!LAZY || (!INV && A::kTracksBounds), "the LAZY variant exists for the FP64 forward kernels only");
  /* (kCanonInFlag: a transform kernel's inputs are canonical words, also after the fold of `wide` inputs -- ntt_core.h bfly_reduces) */
  constexpr uint32_t MASK = fused_mask<A, LOGN, INV, KSH, LAZY>() | (INV && LASTINV ? kLastInvFlag : 0u) | (INV && A::kWide52 ? kCanonInFlag : 0u);
  constexpr int LDS_TW = G::LDS_TW;
  __shared__ typename A::val lds_all[G::BPW * P::LDS_ELEMS + LDS_TW];

  const uint32_t     tid = threadIdx.x;
  /* (two blocks per 1024-thread workgroup, the A/B shape PERSIST2: the half is a property of the wave -- said so, the block index and
   * every address derived from it stay in scalar registers) */
  const uint32_t     sub = G::PERSIST2 ? uniform_u32(tid >> P::LT) : (tid >> P::LT);
  const uint32_t     t   = tid & (P::T - 1);
  typename A::val *  lds = lds_all + sub * P::LDS_ELEMS;
  const uint32_t     bmask = (1u << p.s0) - 1u;

  /* Persistent forward loop (one block per workgroup, grid = resident workgroups).
   * Ordering of the vector-memory queue is what matters here, because vmcnt
   * retires in order: per block the last group's twiddles are requested first,
   * then the NEXT block's 16 coefficient loads; the second-to-last group reads
   * its twiddles from an LDS-resident table (lgkmcnt) and the first two groups
   * through the scalar cache.  So no twiddle wait ever sits behind HBM loads,
   * and the prefetched block lands during ~10 stages of butterflies. */
  if constexpr(!INV && G::PERSISTENT && !A::kRadix4) {
    constexpr int  GL     = P::NG - 1;          /* last group                      */
    /* the last group's 12 per-lane twiddles (8-byte form) are requested well
     * ahead of their use; for whole-polynomial blocks they do not depend on the
     * block at all and stay in 24 VGPRs for the entire launch (LOGN 14 only:
     * smaller blocks have several workgroups per CU hiding that latency) */
    constexpr bool PRE    = A::kCompact && (LOGN == 14 || LOGN == 13 || LOGN == kPreAlso) && stage_is_compact<A, LOGN, false>(GL, 0) && G::TBL(GL) == 0;
    constexpr bool LTW    = LDS_TW > 0;
    /* BPW == 2 (2^13): each half of the workgroup owns the block b0 + sub; a half without a block (odd count)
     * shadows the last one and never stores.  BPW == 1: tt, ll, b are tid, lds_all, b0 -- unchanged code. */
    const uint32_t         tt     = G::BPW == 1 ? tid : t;
    typename A::val *const ll     = G::BPW == 1 ? lds_all : lds;
    const uint64_t         stride = (uint64_t)gdim * G::BPW;
    uint64_t               b0     = (uint64_t)bid * G::BPW;
    if(b0 >= p.nblocks) return;
    const uint64_t lastb = p.nblocks - 1;
    uint64_t       b     = G::BPW == 1 ? b0 : (b0 + sub < p.nblocks ? b0 + sub : lastb);
    /* twiddle tables of this workgroup, behind the exchange buffer */
    typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
    const lds_ctw_ptr<A>   ltw  = (lds_ctw_ptr<A>)tabl;
    if constexpr(LTW) {
      fill_lds_tables<A, LOGN, INV>(tabl, p, (uint32_t)b & bmask, tid);
      __syncthreads();
    }
    /* Block offsets run one block ahead of the loads: off_cur = the block being transformed, off_nxt = the block whose words are
     * requested during this iteration -- computed (a pointer batch: read from the table, Params::ptab, one scalar load) a whole
     * iteration before the prefetch that uses it, so no table latency ever sits in front of the coefficient loads */
    const auto next_blk = [&](uint64_t at) -> uint64_t {
      const uint64_t n0 = below(at + stride, p.nblocks) ? at + stride : at;
      return G::BPW == 1 ? n0 : (n0 + sub < p.nblocks ? n0 + sub : lastb);
    };
    uint64_t off_cur = blk_off_t<LOGN>(p, b), off_nxt = blk_off_t<LOGN>(p, next_blk(b0));
    uint64_t raw[kE];
    prefetch_first<LOGN>(raw, tt, p.a + off_cur);
    pin_raw(raw);
    /* The last group's per-lane twiddles (8-byte form, 24 VGPRs) stay in registers for the whole
     * launch: a workgroup always sees the same block position (its stride over the blocks is a
     * multiple of the blocks per polynomial, as for the LDS tables), so they never change.  This
     * spilled while the kernel needed more registers elsewhere; since the instruction-count work
     * it fits (120 VGPRs) and removes the per-block loads and the only vmcnt wait inside the
     * loop -- the prefetched block now has the entire iteration to arrive (measured +2 %). */
    typename A::ctw pre[4][kE / 2];
    if constexpr(PRE) {
      preload_group_tw<A, LOGN, GL>(pre, tt, (uint32_t)b & bmask, p);
      pin_preloaded<A, LOGN, GL>(pre);
    }
    for(; b0 < p.nblocks; b0 += stride) {
      const bool live = G::BPW == 1 || b0 + sub < p.nblocks;
      b               = live ? b0 + (G::BPW == 1 ? 0u : sub) : lastb;
      const uint32_t blk  = (uint32_t)b & bmask;
      uint64_t *     base = p.a + off_cur;
      typename A::val x[kE];
      convert_inputs<A, false>(x, raw, p.wide != 0, p.c);
      {
        /* request the next block as soon as this block's raw words have been consumed:
         * its HBM loads are then in flight for the whole iteration (measured best of
         * four placements: after the first exchange -4 %, inside the last group -3 %,
         * a quarter after every exchange -7 %; profiles/r01/ablations.txt) */
        const bool     more = b0 + stride < p.nblocks;
        prefetch_first<LOGN>(raw, tt, p.a + off_nxt, more);
        off_cur = off_nxt;
        off_nxt = blk_off_t<LOGN>(p, next_blk(more ? b0 + stride : b0)); /* the block after the next: used one iteration from now */
      }
      run_group<A, LOGN, 0, false, MASK>(x, tt, blk, p);
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        exchange<A, LOGN, GI, GI + 1>(x, tt, ll);
        if constexpr(PRE && GI + 1 == GL) {
          run_group_preloaded<A, LOGN, GL, MASK>(x, pre, p);
        } else if constexpr(G::TBL(GI + 1) > 0) {
          run_group<A, LOGN, GI + 1, false, MASK, true>(x, tt, blk, p, ltw + G::TBL_OFF(GI + 1));
        } else {
          run_group<A, LOGN, GI + 1, false, MASK>(x, tt, blk, p);
        }
      });
      /* whole-line stores: measured +0.6..0.9 % at 2^14, -0.5 % at 2^12 (profiles/r02/ablations.txt) */
      if constexpr(LOGN == 14) {
        store_last_whole_lines<A, LOGN, LAZY>(x, tid, base, p.c, p.lazy != 0);
      } else {
        if(live) global_store_last<A, LOGN, false, LAZY>(x, tt, base, p.c, p.lazy != 0);
      }
    }
    return;
  }
  /* Persistent inverse loop: the mirror image of the forward one.  Groups run
   * last -> first (Gentleman-Sande), the first group executed owns the per-lane
   * twiddles, the next one reads the LDS-resident table, the remaining stages are
   * wave-uniform; coefficients come in as 16-byte loads and leave as coalesced
   * 8-byte stores. */
  /* (FP64 policy only: with the integer policy's larger temporaries this loop spills 6-10
   * VGPRs and the plain loop below is 2-13 % faster -- measured, profiles/r01/ablations.txt) */
  if constexpr(INV && G::PERSISTENT && A::kCompact) {
    constexpr int  GL     = P::NG - 1;
    constexpr bool LTW    = LDS_TW > 0;
    const uint32_t         tt     = G::BPW == 1 ? tid : t;
    typename A::val *const ll     = G::BPW == 1 ? lds_all : lds;
    const uint64_t         stride = (uint64_t)gdim * G::BPW;
    uint64_t               b0     = (uint64_t)bid * G::BPW;
    if(b0 >= p.nblocks) return;
    const uint64_t lastb = p.nblocks - 1;
    uint64_t       b     = G::BPW == 1 ? b0 : (b0 + sub < p.nblocks ? b0 + sub : lastb);
    /* twiddle tables of this workgroup, behind the exchange buffer */
    typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
    const lds_ctw_ptr<A>   ltw  = (lds_ctw_ptr<A>)tabl;
    if constexpr(LTW) {
      fill_lds_tables<A, LOGN, INV>(tabl, p, (uint32_t)b & bmask, tid);
      __syncthreads();
    }
    /* the first executed group's per-lane twiddles do not change from block to block (the
     * workgroup always sees the same block position): loaded once, they stay in 24 VGPRs for
     * the whole launch (measured +5 % at 2^14 over re-requesting them every block) */
    /* (not for the 2^12 kernel of the q <= 2^50 class: its different reduction plan needs one register
     * more and would spill; it keeps the per-stage loads) */
    constexpr bool IPRE = A::kCompact && LOGN >= kIpreMin && stage_is_compact<A, LOGN, true>(GL, 0) && P::R(GL) < 4 && G::TBL(GL) == 0 &&
                          !(LOGN == 12 && KSH == 1);
    typename A::ctw pre[4][kE / 2];
    if constexpr(IPRE) preload_group_tw<A, LOGN, GL>(pre, tt, (uint32_t)b & bmask, p);
    /* (block offsets one block ahead of the loads, as in the forward loop) */
    const auto next_blk = [&](uint64_t at) -> uint64_t {
      const uint64_t n0 = below(at + stride, p.nblocks) ? at + stride : at;
      return G::BPW == 1 ? n0 : (n0 + sub < p.nblocks ? n0 + sub : lastb);
    };
    uint64_t off_cur = blk_off_t<LOGN>(p, b), off_nxt = blk_off_t<LOGN>(p, next_blk(b0));
    uint64_t raw[kE];
    prefetch_last<LOGN>(raw, tt, p.a + off_cur);
    pin_raw(raw);
    if constexpr(IPRE) pin_preloaded<A, LOGN, GL>(pre);
    for(; b0 < p.nblocks; b0 += stride) {
      const bool live = G::BPW == 1 || b0 + sub < p.nblocks;
      b               = live ? b0 + (G::BPW == 1 ? 0u : sub) : lastb;
      const uint32_t blk  = (uint32_t)b & bmask;
      uint64_t *     base = p.a + off_cur;
      typename A::val x[kE];
      convert_inputs<A, true>(x, raw, p.wide != 0, p.c);
      {
        /* (whole-line loads through the same lane swap as the forward stores: measured 17.74 vs 17.75 M, not kept) */
        const bool     more = b0 + stride < p.nblocks;
        prefetch_last<LOGN>(raw, tt, p.a + off_nxt, more);
        off_cur = off_nxt;
        off_nxt = blk_off_t<LOGN>(p, next_blk(more ? b0 + stride : b0));
      }
      if constexpr(IPRE) {
        run_group_preloaded<A, LOGN, GL, MASK, true>(x, pre, p);
      } else if constexpr(G::TBL(GL) > 0) {
        run_group<A, LOGN, GL, true, MASK, true>(x, tt, blk, p, ltw + G::TBL_OFF(GL));
      } else {
        run_group<A, LOGN, GL, true, MASK>(x, tt, blk, p);
      }
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = P::NG - 1 - decltype(gg)::value;
        exchange<A, LOGN, GI, GI - 1>(x, tt, ll);
        if constexpr(G::TBL(GI - 1) > 0) {
          run_group<A, LOGN, GI - 1, true, MASK, true>(x, tt, blk, p, ltw + G::TBL_OFF(GI - 1));
        } else {
          run_group<A, LOGN, GI - 1, true, MASK>(x, tt, blk, p);
        }
      });
      uint64_t out[kE];
      static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = A::store_inv(x[decltype(ee)::value], p.c); });
      if(live) buffer_store_first_raw<LOGN>(out, tt, base);
    }
    return;
  }

  /* generic loop: small blocks (several per workgroup), the integer policy's inverse */
  const lds_ctw_ptr<A> gtw = (lds_ctw_ptr<A>)reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
  if constexpr(LDS_TW > 0) {
    /* blocks below 2^14 are whole polynomials (ntt_passplan.h; launch_fused refuses anything
     * else here), so every block of the workgroup uses the same tables */
    fill_lds_tables<A, LOGN, INV>(reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS), p, 0u, tid);
    __syncthreads();
  }
  for(uint64_t b0 = (uint64_t)bid * G::BPW; b0 < p.nblocks; b0 += (uint64_t)gdim * G::BPW) {
    uint64_t   b    = b0 + sub;
    const bool live = b < p.nblocks;
    if(!live) b = p.nblocks - 1; /* idle lanes shadow a real block, never store */
    const uint32_t blk  = (uint32_t)b & bmask;
    /* (several blocks per workgroup: `sub` is per wave at most, the table entry of a pointer batch a per-lane load) */
    uint64_t *     base = p.a + (poly_offset<false>(b >> p.s0, p.pstride, p.ptab) + ((b & bmask) << LOGN));
    typename A::val x[kE];
    if constexpr(!INV) {
      global_load_first<A, LOGN, false>(x, t, base, p.wide != 0, p.c);
      run_group<A, LOGN, 0, false, MASK, (G::TBL(0) > 0)>(x, t, blk, p, gtw);
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        exchange<A, LOGN, GI, GI + 1>(x, t, lds);
        run_group<A, LOGN, GI + 1, false, MASK, (G::TBL(GI + 1) > 0)>(x, t, blk, p, gtw + G::TBL_OFF(GI + 1));
      });
      if(live) global_store_last<A, LOGN, false, LAZY>(x, t, base, p.c, p.lazy != 0);
    } else {
      global_load_last<A, LOGN, true>(x, t, base, p.wide != 0, p.c);
      run_group<A, LOGN, P::NG - 1, true, MASK, (G::TBL(P::NG - 1) > 0)>(x, t, blk, p, gtw + G::TBL_OFF(P::NG - 1));
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = P::NG - 1 - decltype(gg)::value;
        exchange<A, LOGN, GI, GI - 1>(x, t, lds);
        run_group<A, LOGN, GI - 1, true, MASK, (G::TBL(GI - 1) > 0)>(x, t, blk, p, gtw + G::TBL_OFF(GI - 1));
      });
      if(live) global_store_first<A, LOGN, true>(x, t, base, p.c, p.lazy != 0);
    }
  }
}

/* ------------------------------------------------------------------ */
/* N = 2^15 .. 2^17: both HBM passes of a polynomial inside ONE workgroup */
/* ------------------------------------------------------------------ */
/*
 * A transform larger than one fused block needs two passes over the polynomial: LEAD = m - 14 strided stages
 * (the columns) and the fused 2^14-point blocks.  As two launches over the whole batch every coefficient
 * crosses HBM four times (measured 0.33-0.35 of the 16*N roofline, profiles/r01/sweep_sizes.txt).  Here one
 * 1024-thread workgroup owns a whole polynomial (2/4/8 blocks = 256 KiB .. 1 MiB) and runs the column stages
 * and then its blocks back to back, so what the first pass wrote is read again by the same CU a few tens of
 * microseconds later -- at most 256 polynomials (256 MiB at 2^17) are in that state chip-wide, which the L2s
 * and the 256 MiB Infinity Cache absorb instead of HBM (tools/skel.hip "two-phase": 0.39-0.46 against 0.34 for
 * two launches, memory only).  No inter-workgroup synchronisation: the hand-off is a workgroup barrier.
 * Inside a workgroup the memory-bound column phase and the compute-bound block phase alternate; the CUs drift
 * apart, so chip-wide both kinds of work are in flight at any time.  Both phases keep their own pipelines full:
 *   columns: rounds of 16 values per thread, the next round's loads in flight while this one is computed
 *            (double buffer), wave-uniform twiddles through the scalar cache;
 *   blocks : the persistent loop of fused_kernel -- next block prefetched into registers, last group's
 *            twiddles requested early, second-to-last group's twiddles from an LDS table that is REFRESHED
 *            per block position between the two barriers of the cross-wave exchange (no extra barrier).
 * Reference precedent for "finish one sub-transform while its data is still close":
 * third_party/hexl/fwd-ntt-avx512.c:311-329 (depth-first recursion).
 */
template <int LEAD> __device__ __forceinline__ __amdgpu_buffer_rsrc_t poly_rsrc(const uint64_t *poly)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<uint64_t *>(poly), 0, (int)(8u << (kFusedLarge + LEAD)), 0x00020000);
}

/* One thread owns columns tid, tid + 1024, ... (16 of them); a round handles 16 >> LEAD columns = 16 values:
 * value v = cc * 2^LEAD + e is element e (index e * 2^14 + column) of column cc of the round. */
template <class A, int LEAD, bool INV, int KSH>
__device__ __forceinline__ void twophase_columns(uint64_t *base, uint32_t tid, const Params<A> &p, bool wide_in, bool lazy_out)
{
  constexpr int      NE    = 1 << LEAD;
  constexpr int      CPR   = kE / NE;
  constexpr int      NR    = kE / CPR;
  constexpr uint32_t CMASK = column_mask<A, LEAD, INV, KSH>();
  const __amdgpu_buffer_rsrc_t r = poly_rsrc<LEAD>(base);
  typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
  /* one raw buffer: the next round is requested as soon as this round's words have been converted, so its
   * loads are in flight during the butterflies and the stores (the same scheme as the block loops) */
  uint64_t raw[kE];
  auto load_round = [&](auto rr, uint64_t(&dst)[kE]) {
    constexpr int RD = decltype(rr)::value;
    static_for<0, kE>([&](auto vv) {
      constexpr int      V   = decltype(vv)::value;
      constexpr uint32_t OFF = ((uint32_t)(V % NE) << kFusedLarge) + (uint32_t)(RD * CPR + V / NE) * 1024u;
      dst[V]                 = buffer_load_u64(r, tid * 8u, OFF * 8u);
    });
  };
  load_round(std::integral_constant<int, 0>{}, raw);
  static_for<0, NR>([&](auto rr) {
    constexpr int RD = decltype(rr)::value;
    typename A::val x[kE];
    convert_inputs<A, INV>(x, raw, wide_in, p.c);
    if constexpr(RD + 1 < NR) load_round(std::integral_constant<int, RD + 1>{}, raw);
    static_for<0, LEAD>([&](auto jj) {
      constexpr int  J   = INV ? (LEAD - 1 - decltype(jj)::value) : decltype(jj)::value;
      constexpr int  AB  = LEAD - 1 - J;
      constexpr int  POS = INV ? (LEAD - 1 - J) : J;
      constexpr bool RED = (CMASK >> POS) & 1u;
      static_for<0, kE>([&](auto vv) {
        constexpr int V  = decltype(vv)::value;
        constexpr int E0 = V % NE;
        if constexpr(((E0 >> AB) & 1) == 0) {
          constexpr int V1 = V | (1 << AB);
          if constexpr(INV && J == 0) {
            A::inv_bfly_last(x[V], x[V1], p.c); /* global stage 0 ends the inverse transform: N^-1 folded in */
          } else {
            const typename A::tw w = load_tw<A, true>(p.tw, (1u << J) + (uint32_t)(E0 >> (LEAD - J)));
            if constexpr(INV) {
              A::template inv_bfly<RED>(x[V], x[V1], w, p.c);
            } else {
              A::template fwd_bfly<RED>(x[V], x[V1], w, p.c);
            }
          }
        }
      });
    });
    static_for<0, kE>([&](auto vv) {
      constexpr int      V   = decltype(vv)::value;
      constexpr uint32_t OFF = ((uint32_t)(V % NE) << kFusedLarge) + (uint32_t)(RD * CPR + V / NE) * 1024u;
      const uint64_t     u   = out_word<A, INV, false>(x[V], lazy_out, p.c);
      v2u32              w2;
      w2.x = (unsigned)u;
      w2.y = (unsigned)(u >> 32);
      __builtin_amdgcn_raw_buffer_store_b64(w2, r, (int)(tid * 8u), (int)(OFF * 8u), 0);
    });
  });
}

/* second-to-last group's LDS twiddle table for block position blk: values into registers (at most 5 per thread),
 * later written to LDS in the transposed layout of fill_lds_tables */
template <class A, int LOGN, bool INV> struct TableRegs {
  using P                  = Plan<LOGN>;
  using G                  = Geom<LOGN, INV, flavor_of<A>()>;
  static constexpr int GI  = P::NG - 2;
  static constexpr int SG  = P::S(GI);
  static constexpr int R   = P::R(GI);
  static constexpr int CNT(int jj) { return ((1 << (SG + jj)) + G::WG - 1) / G::WG; }
  static constexpr int TOTAL()
  {
    int n = 0;
    for(int j = 0; j < R; j++) n += CNT(j);
    return n;
  }
  typename A::ctw v[TOTAL() > 0 ? TOTAL() : 1];

  __device__ __forceinline__ void load(const Params<A> &p, uint32_t blk, uint32_t tid)
  {
    asm volatile("" : "+v"(tid)); /* as in store(): keep the lane offsets out of the loop-invariant set */
    int k = 0;
    static_for<0, R>([&](auto jj) {
      constexpr int JJ  = decltype(jj)::value;
      constexpr int SLJ = SG + JJ;
      const typename A::ctw *src = p.tw8 + ((size_t)1 << (p.s0 + SLJ)) + ((size_t)blk << SLJ);
      static_for<0, CNT(JJ)>([&](auto cc) {
        const uint32_t l = tid + (uint32_t)decltype(cc)::value * G::WG;
        v[k]             = l < (1u << SLJ) ? at32(src, l) : typename A::ctw{};
        k++;
      });
    });
  }
  __device__ __forceinline__ void store(typename A::ctw *tabl, uint32_t tid) const
  {
    /* recomputed per block on purpose: hoisted out of the block loop these five LDS addresses would occupy
     * registers for the whole launch (they were the kernel's only spills) */
    asm volatile("" : "+v"(tid));
    typename A::ctw *tg = tabl + G::TBL_OFF(GI);
    int              k  = 0;
    static_for<0, R>([&](auto jj) {
      constexpr int JJ  = decltype(jj)::value;
      constexpr int SLJ = SG + JJ;
      static_for<0, CNT(JJ)>([&](auto cc) {
        const uint32_t l = tid + (uint32_t)decltype(cc)::value * G::WG;
        if(l < (1u << SLJ)) {
          const uint32_t u = l & ((1u << JJ) - 1u), prefix = l >> JJ;
          tg[(((1u << JJ) - 1u) << SG) + (u << SG) + prefix] = v[k];
        }
        k++;
      });
    });
  }
};

/* forward block loop: the exchange after which the last group's twiddles (kTpPreAt) and the next block
 * (kTpPfAt) are requested -- as late as their latency allows, so that the registers carry them only then */
constexpr int kTpPreAt = 1;
constexpr int kTpPfAt  = 2;
template <class A, int LEAD, bool INV, int KSH>
__global__ void __launch_bounds__(1024, 4) twophase_kernel(const KArgs<A> k)
{
  uint32_t        bid, gdim, limb_;
  const Params<A> pin = limb_params<A, INV, false>(k, bid, gdim, limb_);
  constexpr int LOGN = kFusedLarge;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, INV, flavor_of<A>()>;
  static_assert(G::BPW == 1 && P::T == 1024, "two-phase kernel is built on the 2^14 block");
  constexpr uint32_t MASK = fused_mask<A, LOGN, INV, KSH>(); /* forward: the transform's last pass; inverse: not its last */
  constexpr int      NBLK = 1 << LEAD;
  constexpr int      GL   = P::NG - 1;
  constexpr bool     LTW  = G::LDS_TW > 0;
  /* per-lane twiddles of the last group (forward) / first executed group (inverse) in registers, requested early */
  constexpr bool     PRE  = A::kCompact && stage_is_compact<A, LOGN, INV>(GL, 0) && G::TBL(GL) == 0 && P::R(GL) < 4;
  __shared__ typename A::val lds_all[P::LDS_ELEMS + G::LDS_TW];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + P::LDS_ELEMS);
  const lds_ctw_ptr<A>   ltw  = (lds_ctw_ptr<A>)tabl;
  const uint32_t         tid  = threadIdx.x;
  Params<A>              p    = pin;
  p.s0                        = LEAD;
  /* words exchanged between the two phases: the integer policies keep the reference's lazy ranges, the FP64
   * policy canonical words (out_word ignores the flag for it) */
  constexpr bool MID_LAZY = !A::kTracksBounds;

  for(uint64_t poly = bid; poly < p.nblocks; poly += gdim) {
    uint64_t *const base = p.a + poly_offset<true>(poly, p.pstride, p.ptab);
    if constexpr(!INV) {
      twophase_columns<A, LEAD, false, KSH>(base, tid, p, p.wide != 0, MID_LAZY);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      uint64_t raw[kE];
      prefetch_first<LOGN>(raw, tid, base);
#pragma unroll 1
      for(uint32_t blk = 0; blk < (uint32_t)NBLK; blk++) {
        uint64_t *const       bb = base + ((uint64_t)blk << LOGN);
        TableRegs<A, LOGN, false> tr;
        typename A::val x[kE];
        convert_inputs<A, false>(x, raw, false, p.c);
        run_group<A, LOGN, 0, false, MASK>(x, tid, blk, p);
        /* this block's table entries: requested only now (x and the prefetch are the only live values), they
         * arrive while the waves meet at the exchange's first barrier and scatter */
        __builtin_amdgcn_sched_barrier(0);
        if constexpr(LTW) tr.load(p, blk, tid);
        __builtin_amdgcn_sched_barrier(0);
        typename A::ctw pre[4][kE / 2];
        static_for<0, P::NG - 1>([&](auto gg) {
          constexpr int GI = decltype(gg)::value;
          if constexpr(GI == 0 && LTW) {
            exchange<A, LOGN, GI, GI + 1>(x, tid, lds_all, [&]() { tr.store(tabl, tid); });
          } else {
            exchange<A, LOGN, GI, GI + 1>(x, tid, lds_all);
          }
          if constexpr(GI == kTpPreAt && PRE) preload_group_tw<A, LOGN, GL>(pre, tid, blk, p);
          if constexpr(GI == kTpPfAt) {
            /* the next block comes from the L2 / Infinity Cache (this workgroup's column phase wrote it): requested
             * here, it has the rest of the block to arrive, and the register file is not carrying it during the
             * first three groups.  Always issued (the last block re-requests itself): a conditional refill would
             * keep the old words alive. */
            const uint32_t nb = blk + 1 < (uint32_t)NBLK ? blk + 1 : blk;
            prefetch_first<LOGN>(raw, tid, base + ((uint64_t)nb << LOGN));
          }
          if constexpr(PRE && GI + 1 == GL) {
            run_group_preloaded<A, LOGN, GL, MASK>(x, pre, p);
          } else if constexpr(G::TBL(GI + 1) > 0) {
            run_group<A, LOGN, GI + 1, false, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI + 1));
          } else {
            run_group<A, LOGN, GI + 1, false, MASK>(x, tid, blk, p);
          }
        });
        store_last_whole_lines<A, LOGN, false>(x, tid, bb, p.c, p.lazy != 0);
      }
    } else {
      /* blocks first (their inputs come from HBM), then the columns */
      uint64_t raw[kE];
      prefetch_last<LOGN>(raw, tid, base);
      typename A::ctw pre[4][kE / 2];
      {
        /* per-polynomial prologue: its lane offsets are recomputed here rather than kept in registers (or
         * scratch) for the whole launch */
        uint32_t tp = tid;
        asm volatile("" : "+v"(tp));
        if constexpr(PRE) preload_group_tw<A, LOGN, GL>(pre, tp, 0u, p);
        if constexpr(LTW) {
          __syncthreads(); /* the previous polynomial's readers of the table are done */
          fill_lds_tables<A, LOGN, true>(tabl, p, 0u, tp);
          __syncthreads();
        }
      }
#pragma unroll 1
      for(uint32_t blk = 0; blk < (uint32_t)NBLK; blk++) {
        uint64_t *const bb = base + ((uint64_t)blk << LOGN);
        typename A::val x[kE];
        convert_inputs<A, true>(x, raw, p.wide != 0, p.c);
        const uint32_t nb = blk + 1 < (uint32_t)NBLK ? blk + 1 : blk; /* clamped: every refill below is unconditional */
        prefetch_last<LOGN>(raw, tid, base + ((uint64_t)nb << LOGN));
        if constexpr(PRE) {
          run_group_preloaded<A, LOGN, GL, MASK, true>(x, pre, p);
        } else {
          run_group<A, LOGN, GL, true, MASK>(x, tid, blk, p);
        }
        TableRegs<A, LOGN, true> tr;
        if constexpr(LTW) tr.load(p, nb, tid);
        static_for<0, P::NG - 1>([&](auto gg) {
          constexpr int GI = P::NG - 1 - decltype(gg)::value;
          if constexpr(GI == 1 && LTW) {
            exchange<A, LOGN, GI, GI - 1>(x, tid, lds_all, [&]() { tr.store(tabl, tid); });
            /* next block's first-group twiddles: their registers are free from here on */
            if constexpr(PRE) preload_group_tw<A, LOGN, GL>(pre, tid, nb, p);
          } else {
            exchange<A, LOGN, GI, GI - 1>(x, tid, lds_all);
          }
          if constexpr(G::TBL(GI - 1) > 0) {
            run_group<A, LOGN, GI - 1, true, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI - 1));
          } else {
            run_group<A, LOGN, GI - 1, true, MASK>(x, tid, blk, p);
          }
        });
        uint64_t out[kE];
        static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = out_word<A, true, false>(x[decltype(ee)::value], MID_LAZY, p.c); });
        buffer_store_first_raw<LOGN>(out, tid, bb);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      twophase_columns<A, LEAD, true, KSH>(base, tid, p, false, p.lazy != 0);
    }
  }
}

/* ------------------------------------------------------------------ */
/* N = 2^15 in ONE pass: the whole polynomial in the registers of one workgroup */
/* ------------------------------------------------------------------ */
/*
 * Round 6 (review r05 item 5; tools/skel15.hip, profiles/r06/skel15.txt).  A 2^15-point transform is one stage on pairs 2^14
 * apart and two independent 2^14-point transforms with the twiddles of block positions 0 and 1 -- the shape the 2^14 block
 * kernel already runs below a column pass.  A 1024-thread workgroup that owns BOTH halves of a polynomial (32 words per thread,
 * 256 KiB of registers per CU) runs that first stage thread-locally -- slot e of half 0 and slot e of half 1 are index (e << 10) + t
 * and 2^14 + (e << 10) + t -- so every coefficient crosses HBM exactly twice: 16N bytes, where the two-pass forms move 24N..32N across
 * the fabric (team_kernel 0.43 of the roofline forward, per-pass launches 0.38 inverse; the skeleton with this kernel's VALU count
 * and exchanges: 0.60).  Registers decide the schedule: the polynomial is 64 VGPRs, so the NEXT polynomial cannot be prefetched
 * whole.  Forward: half A's 16 words are requested when half A has been stored, half B's when half B has been stored (the first
 * stage of the next polynomial needs both: half B's latency is exposed once per polynomial -- the skeleton prices that at 3 %).
 * Inverse (blocks first, the pair stage last, N^-1 folded): half B of the SAME polynomial arrives while half A runs its fourteen
 * stages; the next polynomial's half A is requested behind the final stores.  The blocks are the body of twophase_kernel's block
 * loops: second-to-last group's twiddles from an LDS table REFRESHED per half between the two barriers of the cross-wave exchange,
 * last group's twiddles requested per half (two halves = two sets: they cannot stay resident as in the 2^14 kernel).
 * Reduction schedule (FP64): ONE schedule over all fifteen forward stages (onepass_fwd_mask) -- the pair stage is its bit 0, the
 * blocks take the rest; inverse: the blocks' per-slot plan, then both inputs of the pair stage reduced (3 exact instructions each;
 * the plan's bound at a block's end is not an input the last butterfly's 2B <= LIM argument covers).
 * Reference precedent: src/ntt_radix4x4.c:54-78 (several stages on values held close), third_party/hexl/fwd-ntt-avx512.c:311-329.
 */
/* one half (block position blk of a 2^15-point polynomial, s0 = 1) through the forward block stages and out to memory */
template <class A, uint32_t MASK>
__device__ __forceinline__ void onepass_block_fwd(typename A::val (&x)[kE], uint32_t blk, uint32_t tid, uint64_t *bb, const Params<A> &p,
                                                  typename A::val *lds_all, typename A::ctw *tabl)
{
  constexpr int LOGN = kFusedLarge;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, flavor_of<A>()>;
  constexpr int  GL  = P::NG - 1;
  constexpr bool LTW = G::LDS_TW > 0;
  constexpr bool PRE = A::kCompact && stage_is_compact<A, LOGN, false>(GL, 0) && G::TBL(GL) == 0 && P::R(GL) < 4;
  const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  TableRegs<A, LOGN, false> tr;
  run_group<A, LOGN, 0, false, MASK>(x, tid, blk, p);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr(LTW) tr.load(p, blk, tid);
  __builtin_amdgcn_sched_barrier(0);
  typename A::ctw pre[4][kE / 2];
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = decltype(gg)::value;
    if constexpr(GI == 0 && LTW) {
      exchange<A, LOGN, GI, GI + 1>(x, tid, lds_all, [&]() { tr.store(tabl, tid); });
    } else {
      exchange<A, LOGN, GI, GI + 1>(x, tid, lds_all);
    }
    if constexpr(GI == kTpPreAt && PRE) preload_group_tw<A, LOGN, GL>(pre, tid, blk, p);
    if constexpr(PRE && GI + 1 == GL) {
      run_group_preloaded<A, LOGN, GL, MASK>(x, pre, p);
    } else if constexpr(G::TBL(GI + 1) > 0) {
      run_group<A, LOGN, GI + 1, false, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI + 1));
    } else {
      run_group<A, LOGN, GI + 1, false, MASK>(x, tid, blk, p);
    }
  });
  store_last_whole_lines<A, LOGN, false>(x, tid, bb, p.c, false);
}

/* one half through the inverse block stages (not the transform's last pass): x holds the block's results, unreduced, in the
 * first-kind layout.  nblk: the half whose LDS table is fetched for the NEXT call (the refresh sits between this call's cross-wave
 * barriers).  The first group's per-lane twiddles are fetched stage by stage, a stage ahead (run_group's pipelining): requested a
 * half ahead, as twophase_kernel does, their 24 registers next to the waiting half cost 20-44 spilled VGPRs and 3 % (measured:
 * profiles/r06/onepass_2p15.txt). */
template <class A, uint32_t MASK>
__device__ __forceinline__ void onepass_block_inv(typename A::val (&x)[kE], uint32_t blk, uint32_t nblk, uint32_t tid, const Params<A> &p,
                                                  typename A::val *lds_all, typename A::ctw *tabl)
{
  constexpr int LOGN = kFusedLarge;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, true, flavor_of<A>()>;
  constexpr int  GL  = P::NG - 1;
  constexpr bool LTW = G::LDS_TW > 0;
  const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  run_group<A, LOGN, GL, true, MASK>(x, tid, blk, p);
  TableRegs<A, LOGN, true> tr;
  if constexpr(LTW) tr.load(p, nblk, tid);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    if constexpr(GI == 1 && LTW) {
      exchange<A, LOGN, GI, GI - 1>(x, tid, lds_all, [&]() { tr.store(tabl, tid); });
    } else {
      exchange<A, LOGN, GI, GI - 1>(x, tid, lds_all);
    }
    if constexpr(G::TBL(GI - 1) > 0) {
      run_group<A, LOGN, GI - 1, true, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI - 1));
    } else {
      run_group<A, LOGN, GI - 1, true, MASK>(x, tid, blk, p);
    }
  });
}

template <class A, bool INV, int KSH, bool MULTI = false>
__global__ void __launch_bounds__(1024, 4) onepass_kernel(const KArgs<A> k)
{
  uint32_t  bid, gdim, limb_;
  Params<A> p = limb_params<A, INV, MULTI>(k, bid, gdim, limb_);
  constexpr int LOGN = kFusedLarge;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, INV, flavor_of<A>()>;
  static_assert(A::kCompact && A::kTracksBounds && G::BPW == 1 && P::T == 1024, "built for the FP64 policies on the 2^14 block");
  __shared__ typename A::val lds_all[P::LDS_ELEMS + G::LDS_TW];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + P::LDS_ELEMS);
  const uint32_t         tid  = threadIdx.x;
  p.s0                        = 1;
  constexpr uint64_t HALF     = 1ull << LOGN;
  uint64_t poly = bid;
  if(!below(poly, p.nblocks)) return;
  /* polynomial offsets one polynomial ahead of the loads, as in fused_kernel's loops (a pointer batch reads them from a table) */
  const auto next_poly = [&](uint64_t at) -> uint64_t { return below(at + gdim, p.nblocks) ? at + gdim : at; };
  uint64_t off_cur = poly_offset<true>(poly, p.pstride, p.ptab), off_nxt = poly_offset<true>(next_poly(poly), p.pstride, p.ptab);
  /* ONE copy of the block body per direction: the two halves are the iterations of a loop that is not unrolled, the half that waits
   * (forward: half B's values; inverse: half B's raw words, then half A's results) parked as bit patterns in `hold` -- two inlined
   * copies let the compiler hoist either copy's lane offsets and LDS addresses out of the polynomial loop, into registers the
   * halves need (79 spilled VGPRs in the first version of this kernel) */
  uint64_t hold[kE], rb[kE];
  const auto bits = [](typename A::val v) -> uint64_t { return __builtin_bit_cast(uint64_t, v); };
  const auto vals = [](uint64_t u) -> typename A::val { return __builtin_bit_cast(typename A::val, u); };
  if constexpr(!INV) {
    constexpr uint32_t M15  = onepass_fwd_mask<A, KSH>();
    constexpr uint32_t MASK = M15 >> 1;
    constexpr bool     RED0 = (M15 & 1u) != 0;
    prefetch_first<LOGN>(hold, tid, p.a + off_cur);
    prefetch_first<LOGN>(rb, tid, p.a + off_cur + HALF);
    for(; below(poly, p.nblocks); poly += gdim) {
      uint64_t *const base = p.a + off_cur;
      const bool      more = below(poly + gdim, p.nblocks);
      uint64_t *const nxt  = p.a + off_nxt;
      off_cur              = off_nxt;
      off_nxt              = poly_offset<true>(next_poly(more ? poly + gdim : poly), p.pstride, p.ptab);
      typename A::val x[kE];
      {
        typename A::val xb[kE];
        convert_inputs<A, false>(x, hold, p.wide != 0, p.c);
        convert_inputs<A, false>(xb, rb, p.wide != 0, p.c);
        onepass_pairs_fwd<A, RED0>(x, xb, p); /* global stage 0, thread-local (ntt_core.h) */
        static_for<0, kE>([&](auto ee) { hold[decltype(ee)::value] = bits(xb[decltype(ee)::value]); });
      }
#pragma unroll 1
      for(uint32_t h = 0; h < 2u; h++) {
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl)); /* per-half lane offsets: recomputed, not carried in registers through the launch */
        onepass_block_fwd<A, MASK>(x, h, tl, base + (h ? HALF : 0), p, lds_all, tabl);
        if(h == 0) {
          static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = vals(hold[decltype(ee)::value]); });
          prefetch_first<LOGN>(hold, tl, nxt, more); /* half A has been stored: the next polynomial's first half */
        }
      }
      prefetch_first<LOGN>(rb, tid, nxt + HALF, more); /* ... and its second half, behind half B's stores */
    }
  } else {
    constexpr uint32_t MASK = fused_mask<A, LOGN, true, KSH>() | (A::kWide52 ? kCanonInFlag : 0u); /* the blocks do not end the transform; canonical inputs */
    constexpr bool     LTW  = G::LDS_TW > 0;
    prefetch_last<LOGN>(rb, tid, p.a + off_cur);
    if constexpr(LTW) {
      fill_lds_tables<A, LOGN, true>(tabl, p, 0u, tid);
      __syncthreads();
    }
    for(; below(poly, p.nblocks); poly += gdim) {
      uint64_t *const base = p.a + off_cur;
      const bool      more = below(poly + gdim, p.nblocks);
      uint64_t *const nxt  = p.a + off_nxt;
      off_cur              = off_nxt;
      off_nxt              = poly_offset<true>(next_poly(more ? poly + gdim : poly), p.pstride, p.ptab);
      typename A::val x[kE];
      convert_inputs<A, true>(x, rb, p.wide != 0, p.c);
      prefetch_last<LOGN>(hold, tid, base + HALF); /* this polynomial's second half arrives during the first half's stages */
#pragma unroll 1
      for(uint32_t h = 0; h < 2u; h++) {
        uint32_t tl = tid;
        asm volatile("" : "+v"(tl));
        onepass_block_inv<A, MASK>(x, h, h ^ 1u, tl, p, lds_all, tabl);
        if(h == 0) {
          /* half A's results wait as bit patterns where half B's raw words were; half B's words become the values */
          /* (slot by slot: a second array of sixteen values between the two would not fit) */
          const auto swap_in = [&](auto wide_c) {
            static_for<0, kE>([&](auto ee) {
              constexpr int         E = decltype(ee)::value;
              const typename A::val v = A::template load<true, decltype(wide_c)::value>(hold[E], p.c);
              hold[E]                 = bits(x[E]);
              x[E]                    = v;
            });
          };
          if(p.wide != 0) swap_in(std::true_type{});
          else swap_in(std::false_type{});
        }
      }
      /* global stage 0 with N^-1 folded in: both inputs reduced first (see the header comment), then the pair's two products,
       * stored pair by pair (slot e <-> index (e << 10) + t of either half: coalesced 8-byte rows) so that no second copy of the
       * polynomial is ever live.  The next polynomial's first half is requested IN FRONT of the stage, whose arithmetic and stores hide
       * part of its way from HBM (96 VGPRs of data for the length of the stage; measured on one box, alternating: requested behind the
       * stage 0.396 of the roofline, in front of pair 8 0.417, in front of the stage 0.438 -- profiles/r06/onepass_inverse_prefetch_position.txt) */
      {
        const __amdgpu_buffer_rsrc_t r0 = block_rsrc<LOGN>(base), r1 = block_rsrc<LOGN>(base + HALF);
        typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
        static_for<0, kE>([&](auto ee) {
          constexpr int E = decltype(ee)::value;
          if constexpr(E == 0) {
            /* (inside the unrolled sequence on purpose: issued in front of it, as a statement of its own, the same request made the
             * register allocator spill 12-16 VGPRs) */
            prefetch_last<LOGN>(rb, tid, nxt, more);
            sched_fence();
          }
          typename A::val va = vals(hold[E]), vb = x[E];
          onepass_pair_inv<A>(va, vb, p.c); /* ntt_core.h */
          const uint64_t ua = A::store_inv(va, p.c), ub = A::store_inv(vb, p.c);
          v2u32          wa, wb;
          wa.x = (unsigned)ua, wa.y = (unsigned)(ua >> 32);
          wb.x = (unsigned)ub, wb.y = (unsigned)(ub >> 32);
          __builtin_amdgcn_raw_buffer_store_b64(wa, r0, (int)(tid * 8u), (int)(((uint32_t)E << P::LT) * 8u), 0);
          __builtin_amdgcn_raw_buffer_store_b64(wb, r1, (int)(tid * 8u), (int)(((uint32_t)E << P::LT) * 8u), 0);
          /* (two pairs at a time: sixteen interleaved pairs would need their temporaries all at once) */
          if constexpr(E % 2 == 1) sched_fence();
        });
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* N = 2^15 .. 2^17: both passes in one launch, intermediate kept in the XCD's L2 */
/* ------------------------------------------------------------------ */
/*
 * The two passes of a large transform -- LEAD = m - 12 strided column stages, then the 2^12-point blocks -- as ITEMS of
 * one persistent launch instead of two launches per 256 MiB chunk.  A column item is 256 adjacent columns of one
 * polynomial (column_pass_thread's work: 2^LEAD values per thread in registers, no exchange, every access a contiguous
 * 2 KiB row segment), a row item one 2^12-point block (the fused block kernel's work).  What makes it worth a kernel:
 *   - a workgroup reads which XCD it runs on (HW_REG_XCC_ID) and pulls items from THAT XCD's queue, so all items of a
 *     polynomial run on one XCD whatever the dispatcher does: the intermediate is written by plain stores into that
 *     XCD's 4 MiB L2 and read back from it -- measured FETCH_SIZE 1.0x the data instead of 2.0x (profiles/r03,
 *     skel_pmc_fabric_traffic.txt).  Final stores are write-through (sc1: the line leaves the L2 at once) and input loads
 *     sc0 sc1, so the streaming sides do not push the waiting intermediates out of the L2;
 *   - queue order per XCD: first-pass items of polynomial j, then second-pass items of polynomial j - LAG (forward:
 *     columns then rows; inverse: rows then columns).  A second-pass item waits on a per-polynomial counter that the
 *     first-pass items bump once their stores have completed (s_waitcnt vmcnt(0), workgroup barrier, agent-scope
 *     atomic).  First-pass items never wait and items are handed out in order, so every item somebody waits for is
 *     already in the hands of a running workgroup: no deadlock whatever the residency;
 *   - polynomials are dealt to the eight queues statically (p mod 8); a queue is processed only by the XCD that owns it
 *     (compare-and-swap on first touch: normally its namesake; an XCD that finds its own queue finished or foreign
 *     adopts queues nobody has claimed), so exactly one L2 sees all items of a polynomial even on a device that exposes
 *     fewer XCDs than eight.
 * Correctness never rests on placement assumptions: the XCD is read, and a hand-off only happens inside one XCD.
 * MEMORY-ORDER INVARIANT of the hand-off (team_kernel, team_product_kernel, team_dot kernels).  The signal is a relaxed
 * agent-scope atomic behind s_waitcnt vmcnt(0) + a workgroup barrier, the poll a relaxed agent-scope load in front of a
 * workgroup barrier; there is deliberately NO release/acquire fence (an agent-scope release is buffer_wbl2: it writes back
 * every dirty L2 line of the XCD, the neighbours' results included -- 8 us per hand-off) and no buffer_inv on the consumer.
 * That is sound because producer and consumer sit on ONE XCD, i.e. behind one L2, which is the point of coherence for
 * them (stores are complete in that L2 once vmcnt reaches 0; the per-CU vector cache is write-through), PROVIDED that no
 * CU's vector cache (TCP) can hold a stale copy of a line the consumer reads:
 *   (1) a line that a later pass of the same launch overwrites from ANOTHER CU is only ever read with sc0 sc1 loads,
 *       which do not allocate in the TCP (kAuxSc0Sc1: the inputs of the first pass);
 *   (2) every other load (nt: may allocate) of a line that is overwritten later in the launch is issued by the very item
 *       that overwrites it: items are block-aligned (a row item reads and writes exactly its own 2^12-point block, a
 *       column item its own 2 KiB row segments), so the only CU that may cache the old contents is the one whose own
 *       write-through stores replace them;
 *   (3) TCPs start a launch invalid, and the launch never reads a final output again.
 * Changing a cache policy or making items overlap in lines breaks this silently; tests/test_gpu_parity.py
 * (test_xcd_local_*) and tools/soak.py compare every polynomial with the per-pass path for that reason.
 * Reference precedent for finishing a sub-transform while its data is close: third_party/hexl/fwd-ntt-avx512.c:311-329.
 */
static_assert(kAuxSc0Sc1 == 17 && kAuxNt == 2 && kAuxSc1 == 16, "cache-policy encodings the hand-off invariant is written for");
struct TeamCtl {
  unsigned next[8][32];  /* per queue: next item; one 128-byte line each */
  unsigned owner[8][32]; /* per queue: 0 = unclaimed, else 1 + the XCD that processes it */
  unsigned done[1];      /* [polynomials] first-pass items finished (flexible) */
};

template <class A> struct KTeam {
  KArgs<A> k;      /* a, limbs[], limb_stride, logn; nblocks = polynomials PER LIMB */
  TeamCtl *ctl;    /* zeroed before the launch */
  uint32_t lag;    /* polynomials between a first-pass item and the second-pass items of the same queue */
  uint32_t nlimbs; /* MULTI kernels: limbs of the launch (polynomial v of the queues = limb * batch + polynomial) */
  uint64_t split_rcp; /* MULTI kernels: floor(2^64 / D) + 1 for the divisor D of team_split (batch, or nlimbs when poly_major): the
                       * quotient as a scalar multiply-high, exact for every v (v D < 2^64) -- a division would run in the VALU */
  uint32_t poly_major; /* MULTI kernels: v = polynomial * nlimbs + limb instead -- the numbering that follows the ADDRESSES when a
                        * polynomial's limbs lie side by side ([batch][limb][N]): the queues then walk memory as they do in the
                        * limb-major layout (which polynomials are in flight together decides the HBM channel mix) */
};
/* queue polynomial v -> (limb, polynomial inside the limb) */
__device__ __forceinline__ void team_split(uint32_t v, uint32_t batch, uint32_t nlimbs, uint32_t poly_major, uint64_t rcp, uint32_t &limb, uint32_t &pl)
{
  /* (v is wave-uniform: the multiply-high stays in scalar registers, see limb_params; rcp = floor(2^64 / D) + 1 gives the exact quotient) */
  const uint32_t quo = rcp ? (uint32_t)__umul64hi((uint64_t)v, rcp) : v; /* (rcp == 0: divisor 1) */
  if(poly_major) {
    pl   = quo;
    limb = v - pl * nlimbs;
  } else {
    limb = quo;
    pl   = v - limb * batch;
  }
}
inline uint64_t team_split_rcp(uint64_t divisor) { return divisor > 1 ? ~0ull / divisor + 1 : 0; /* (D = 1: the quotient is v itself, see callers) */ }

/* MULTI (several RNS limbs in one launch): the item's limb picks the tables, the constants and the slab.  The items of these
 * kernels fetch their tables per item anyway, so a limb that changes from item to item costs scalar loads only. */
template <class A, bool INV> __device__ __forceinline__ void team_limb(Params<A> &p, const KArgs<A> &k, uint32_t limb)
{
  const LimbRec<A> &r = k.limbs[limb];
  p.a   = k.a + (uint64_t)limb * k.limb_stride;
  p.tw  = INV ? r.tw_i : r.tw_f;
  p.tw8 = INV ? r.tw8_i : r.tw8_f;
  p.c   = r.c;
}

__device__ __forceinline__ uint32_t xcc_id()
{
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 7u;
}

constexpr int kTeamBlock = 12;  /* log2 of the row items (the block size below the column stages) */
constexpr int kTeamCols  = 256; /* adjacent columns of a column item = threads of a workgroup */

/* column item: leading stages [0, R) of a 2^logn-point polynomial on columns col of the 2^R x 2^(logn-R) view; the
 * thread's 2^R values sit 2^(logn-R) apart.  column_pass_thread (ntt_core.h) with S = 0, through a buffer descriptor so
 * that loads and stores carry a cache policy, twiddles through the scalar cache (their slots are compile-time here). */
template <class A, int R, bool INV, uint32_t MASK, int LDAUX, int STAUX>
__device__ __forceinline__ void team_column_item(uint64_t *poly, uint32_t col, uint32_t logn, const Params<A> &p, bool lazy_out)
{
  constexpr int  NE  = 1 << R;
  const uint32_t lsp = logn - R;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(poly, 0, (int)(8u << logn), 0x00020000);
  typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
  uint64_t raw[NE];
  static_for<0, NE>([&](auto ee) {
    constexpr int E = decltype(ee)::value;
    raw[E]          = buffer_load_u64<LDAUX>(r, col * 8u, ((uint32_t)E << lsp) * 8u);
  });
  typename A::val x[NE];
  static_for<0, NE>([&](auto ee) { x[decltype(ee)::value] = A::template load<INV, false>(raw[decltype(ee)::value], p.c); });
  static_for<0, R>([&](auto jj) {
    constexpr int  J   = INV ? (R - 1 - decltype(jj)::value) : decltype(jj)::value;
    constexpr int  AB  = R - 1 - J;
    constexpr int  POS = INV ? (R - 1 - J) : J;
    constexpr bool RED = (MASK >> POS) & 1u;
    static_for<0, NE>([&](auto ee) {
      constexpr int E0 = decltype(ee)::value;
      if constexpr(((E0 >> AB) & 1) == 0) {
        constexpr int E1 = E0 | (1 << AB);
        if constexpr(INV && J == 0) {
          A::inv_bfly_last(x[E0], x[E1], p.c); /* global stage 0 ends the inverse transform: N^-1 folded in */
        } else {
          const typename A::tw w = load_tw<A, true>(p.tw, (1u << J) + (uint32_t)(E0 >> (R - J)));
          if constexpr(INV) {
            A::template inv_bfly<RED>(x[E0], x[E1], w, p.c);
          } else {
            A::template fwd_bfly<RED>(x[E0], x[E1], w, p.c);
          }
        }
      }
    });
  });
  static_for<0, NE>([&](auto ee) {
    constexpr int  E = decltype(ee)::value;
    const uint64_t u = out_word<A, INV, false>(x[E], lazy_out, p.c);
    v2u32          w2;
    w2.x = (unsigned)u;
    w2.y = (unsigned)(u >> 32);
    __builtin_amdgcn_raw_buffer_store_b64(w2, r, (int)(col * 8u), (int)(((uint32_t)E << lsp) * 8u), STAUX);
  });
}

/* row item, forward: one 2^12-point block at position blk of its polynomial (the body of fused_kernel's persistent
 * loop without the prefetch: the table of the second-to-last group and the last group's twiddles are per position,
 * so they are fetched per item -- from the L2, the whole batch shares them) */
template <class A, int KSH, int LDAUX, int STAUX>
__device__ __forceinline__ void team_row_item_fwd(uint64_t *base, uint32_t blk, uint32_t tid, const Params<A> &p,
                                                  typename A::val *lds, typename A::ctw *tabl)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, flavor_of<A>()>;
  constexpr int GL   = P::NG - 1;
  constexpr uint32_t MASK = fused_mask<A, LOGN, false, KSH>();
  static_assert(G::TBL(GL - 1) > 0 && G::TBL(GL) == 0, "twiddle placement this item assumes");
  const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  uint64_t raw[kE];
  prefetch_first<LOGN, LDAUX>(raw, tid, base);
  typename A::ctw pre[4][kE / 2];
  preload_group_tw<A, LOGN, GL>(pre, tid, blk, p);
  fill_lds_tables<A, LOGN, false>(tabl, p, blk, tid);
  /* (the table is read in the second-to-last group; the cross-wave exchange in front of it has two workgroup barriers) */
  static_assert(!P::WAVE_LOCAL(0, 1), "the first exchange must cross waves: its barriers publish the LDS table");
  typename A::val x[kE];
  convert_inputs<A, false>(x, raw, false, p.c);
  run_group<A, LOGN, 0, false, MASK>(x, tid, blk, p);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = decltype(gg)::value;
    exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
    if constexpr(GI + 1 == GL) {
      run_group_preloaded<A, LOGN, GL, MASK>(x, pre, p);
    } else if constexpr(G::TBL(GI + 1) > 0) {
      run_group<A, LOGN, GI + 1, false, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI + 1));
    } else {
      run_group<A, LOGN, GI + 1, false, MASK>(x, tid, blk, p);
    }
  });
  /* whole 128-byte lines per store instruction: a write-through store of half a line costs a full line's write */
  store_last_whole_lines<A, LOGN, false, STAUX>(x, tid, base, p.c, p.lazy != 0);
}

/* row item, inverse: the mirror image (the block pass comes first in the inverse transform) */
template <class A, int KSH, int LDAUX, int STAUX>
__device__ __forceinline__ void team_row_item_inv(uint64_t *base, uint32_t blk, uint32_t tid, const Params<A> &p,
                                                  typename A::val *lds, typename A::ctw *tabl, bool mid_lazy)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, true, flavor_of<A>()>;
  constexpr int GL   = P::NG - 1;
  constexpr uint32_t MASK = fused_mask<A, LOGN, true, KSH>() | (A::kWide52 ? kCanonInFlag : 0u); /* not the pass that ends the transform; canonical inputs */
  const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  uint64_t raw[kE];
  prefetch_last<LOGN, LDAUX>(raw, tid, base);
  constexpr bool IPRE = stage_is_compact<A, LOGN, true>(GL, 0) && P::R(GL) < 4 && G::TBL(GL) == 0 && KSH != 1;
  typename A::ctw pre[4][kE / 2];
  if constexpr(IPRE) preload_group_tw<A, LOGN, GL>(pre, tid, blk, p);
  fill_lds_tables<A, LOGN, true>(tabl, p, blk, tid);
  /* (read after the exchange between the last two groups -- wave-local -- so the table needs its own barrier here) */
  __syncthreads();
  typename A::val x[kE];
  convert_inputs<A, true>(x, raw, p.wide != 0, p.c);
  if constexpr(IPRE) {
    run_group_preloaded<A, LOGN, GL, MASK, true>(x, pre, p);
  } else {
    run_group<A, LOGN, GL, true, MASK>(x, tid, blk, p);
  }
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    exchange<A, LOGN, GI, GI - 1>(x, tid, lds);
    if constexpr(G::TBL(GI - 1) > 0) {
      run_group<A, LOGN, GI - 1, true, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI - 1));
    } else {
      run_group<A, LOGN, GI - 1, true, MASK>(x, tid, blk, p);
    }
  });
  uint64_t out[kE];
  static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = out_word<A, true, false>(x[decltype(ee)::value], mid_lazy, p.c); });
  buffer_store_first_raw<LOGN, STAUX>(out, tid, base);
}

/* the row items of a policy without compact twiddles and LDS tables (the wide integer policy): the block body of
 * fused_kernel's plain loop -- 16-byte records through the scalar cache and the L1/L2, which the whole batch shares -- with
 * the cache policies of the items above (the launch's memory-order invariant does not depend on the arithmetic) */
template <class A, int KSH, int LDAUX, int STAUX>
__device__ __forceinline__ void team_row_item_fwd_plain(uint64_t *base, uint32_t blk, uint32_t tid, const Params<A> &p, typename A::val *lds)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, false, KSH>();
  uint64_t raw[kE];
  prefetch_first<LOGN, LDAUX>(raw, tid, base);
  typename A::val x[kE];
  convert_inputs<A, false>(x, raw, false, p.c);
  run_group<A, LOGN, 0, false, MASK>(x, tid, blk, p);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = decltype(gg)::value;
    exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
    run_group<A, LOGN, GI + 1, false, MASK>(x, tid, blk, p);
  });
  store_last_whole_lines<A, LOGN, false, STAUX>(x, tid, base, p.c, p.lazy != 0);
}
template <class A, int KSH, int LDAUX, int STAUX>
__device__ __forceinline__ void team_row_item_inv_plain(uint64_t *base, uint32_t blk, uint32_t tid, const Params<A> &p, typename A::val *lds,
                                                        bool mid_lazy)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, true, KSH>(); /* not the pass that ends the transform */
  uint64_t raw[kE];
  prefetch_last<LOGN, LDAUX>(raw, tid, base);
  typename A::val x[kE];
  convert_inputs<A, true>(x, raw, p.wide != 0, p.c);
  run_group<A, LOGN, P::NG - 1, true, MASK>(x, tid, blk, p);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    exchange<A, LOGN, GI, GI - 1>(x, tid, lds);
    run_group<A, LOGN, GI - 1, true, MASK>(x, tid, blk, p);
  });
  uint64_t out[kE];
  static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = out_word<A, true, false>(x[decltype(ee)::value], mid_lazy, p.c); });
  buffer_store_first_raw<LOGN, STAUX>(out, tid, base);
}

template <class A, int LEAD, bool INV, int KSH, bool MULTI = false>
__global__ void __launch_bounds__(256, 4) team_kernel(const KTeam<A> kt)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, INV, flavor_of<A>()>;
  static_assert((A::kCompact || A::kIntWide) && P::T == kTeamCols && LEAD >= 3 && LEAD <= 5,
                "built for the FP64 policies and the wide integer policy on 2^12-point blocks, N = 2^15..2^17");
  __shared__ typename A::val lds[P::LDS_ELEMS + G::LDS_TW];
  __shared__ unsigned        s_k, s_k2[2];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds + P::LDS_ELEMS);
  const uint32_t         tid  = threadIdx.x;
  uint32_t               bid_, gdim_, limb_;
  Params<A>              p = limb_params<A, INV, false>(kt.k, bid_, gdim_, limb_);
  p.s0                     = LEAD;
  constexpr uint32_t CMASK = column_mask<A, LEAD, INV, KSH>();
  constexpr bool     MID_LAZY = !A::kTracksBounds; /* words between the passes: canonical for the FP64 policies */
  const uint32_t logn  = LOGN + LEAD;
  const uint32_t batch = (uint32_t)p.nblocks;      /* polynomials per limb */
  const uint32_t total = MULTI ? batch * kt.nlimbs : batch;
  constexpr uint32_t NCOL = 1u << (LOGN - 8);      /* column items per polynomial: 2^(m - LEAD) columns / 256 */
  constexpr uint32_t NROW = 1u << LEAD;            /* row items per polynomial */
  constexpr uint32_t NA   = INV ? NROW : NCOL;     /* first-pass items */
  constexpr uint32_t NB   = INV ? NCOL : NROW;
  TeamCtl *const ctl = kt.ctl;
  const uint32_t lag = kt.lag;
  const uint32_t my  = xcc_id();
  for(uint32_t qq = 0; qq < 8; qq++) {
    const uint32_t q = (my + qq) & 7u; /* own queue first, then whatever nobody claimed */
    if(tid == 0) {
      const unsigned prev = atomicCAS(&ctl->owner[q][0], 0u, my + 1u);
      s_k                 = (prev == 0u || prev == my + 1u) ? 1u : 0u;
    }
    __syncthreads();
    const bool mine = s_k != 0;
    __syncthreads();
    if(!mine) continue;
    /* Every lane-0 block of this loop is followed at once by a workgroup barrier.  A lane-0 block at the END of the body
     * (the completion signal used to sit there) ends up next to the loop's back edge, and the compiler then lets lane 0
     * leave the loop "early" while lanes 1-63 of its wave wait at the next iteration's barrier for the item only lane 0
     * can fetch: the first version of this kernel hung on its first items exactly like that.  So the signal of a finished
     * first-pass item is carried into the next iteration and issued by the same lane-0 block that fetches the next item. */
    constexpr uint32_t kNoSignal = 0xffffffffu;
    uint32_t           sig       = kNoSignal;
    for(uint32_t it = 0;; it ^= 1u) {
      /* the item index travels through one of two LDS words in turn, so that one barrier per fetch is enough (lane 0
       * writes the other word next time: nobody can still be reading it, everybody has passed this barrier since) */
      if(tid == 0) {
        if(sig != kNoSignal) __hip_atomic_fetch_add(&ctl->done[sig], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_k2[it] = atomicAdd(&ctl->next[q][0], 1u);
      }
      sig = kNoSignal;
      __syncthreads();
      /* queue entry -> (pass, item, polynomial): ntt_core.h team_decode, the function tests/test_team_protocol.py simulates */
      const TeamItem ti = team_decode(uniform_u32(s_k2[it]), q, total, lag, NA, NB, 0u);
      if(ti.stop) break;
      if(!ti.valid) continue;
      const bool     second = ti.pass != 0;
      const uint32_t item   = ti.item;
      const uint32_t pidx   = ti.v;
      uint32_t       pl     = pidx; /* the polynomial inside its limb */
      if constexpr(MULTI) {
        uint32_t limb;
        team_split(pidx, batch, kt.nlimbs, kt.poly_major, kt.split_rcp, limb, pl);
        team_limb<A, INV>(p, kt.k, limb);
      }
      uint64_t *poly = p.a + poly_offset<true>((uint64_t)pl, p.pstride, p.ptab);
      if(second) {
        if(tid == 0) {
#ifdef NTT_TEAM_WATCHDOG
          /* development builds: a wait that lasts longer than about a second is recorded (owner[q][1..3]) and abandoned,
           * so that a protocol error shows up as a wrong result with a diagnosis instead of a hung GPU */
          unsigned spins = 0;
          while(__hip_atomic_load(&ctl->done[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NA) {
            __builtin_amdgcn_s_sleep(8);
            if(++spins > (1u << 15)) {
              ctl->owner[q][1] = pidx + 1u;
              ctl->owner[q][2] = __hip_atomic_load(&ctl->done[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              ctl->owner[q][3] = s_k2[it];
              break;
            }
          }
#else
          while(__hip_atomic_load(&ctl->done[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NA) __builtin_amdgcn_s_sleep(8);
#endif
        }
        __syncthreads(); /* (also keeps every later load of the workgroup behind the poll) */
      }
      const bool row = second != INV;
      if(!row) {
        /* forward: inputs -> intermediate (kept dirty in the L2); inverse: intermediate -> final (write-through) */
        if constexpr(INV) team_column_item<A, LEAD, true, CMASK, kAuxNt, kAuxSc1>(poly, item * kTeamCols + tid, logn, p, p.lazy != 0);
        else team_column_item<A, LEAD, false, CMASK, kAuxSc0Sc1, 0>(poly, item * kTeamCols + tid, logn, p, MID_LAZY);
      } else {
        uint64_t *base = poly + ((uint64_t)item << LOGN);
        /* (inverse inputs arrive as 16-byte loads in runs of four coefficients per lane: two instructions share every 128-byte
         * line, so these loads must be allowed to hit the L2 -- nt; with the cache-bypassing policy of the forward
         * inputs every line crossed the fabric twice) */
        if constexpr(!A::kCompact) {
          (void)tabl;
          if constexpr(INV) team_row_item_inv_plain<A, KSH, kAuxNt, 0>(base, item, tid, p, lds, MID_LAZY);
          else team_row_item_fwd_plain<A, KSH, kAuxNt, kAuxSc1>(base, item, tid, p, lds);
        } else if constexpr(INV) team_row_item_inv<A, KSH, kAuxNt, 0>(base, item, tid, p, lds, tabl, MID_LAZY);
        else team_row_item_fwd<A, KSH, kAuxNt, kAuxSc1>(base, item, tid, p, lds, tabl);
      }
      if(!second) {
        /* the item's stores have completed (every wave waits for its own, the barrier collects the waves) before the
         * counter moves -- at the top of the next iteration */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sig = pidx;
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* fused product: c = a * b in Z_q[X]/(X^N+1), b never leaves the CU     */
/* ------------------------------------------------------------------ */
/*
 * The caller-side step either side of the path.  a^ = fwd(a) is in HBM (one ordinary forward
 * launch).  This kernel then does, per polynomial and without touching HBM in between:
 *     load b -> forward transform (14 stages) -> times a^ (read once, in the layout the last forward group
 *     already has) -> inverse transform (14 stages, N^-1 folded) -> store c
 * The forward transform ends and the inverse begins in the same thread <-> index layout (runs of four
 * consecutive coefficients per lane), so the product needs no exchange.  HBM traffic of a product: 16N (fwd a)
 * + 24N (this kernel) = 40N bytes instead of 72N for four launches (and 3 -> 2 launches); the kernel itself is
 * bound by its 2 x 14 stages of butterflies, the 24N bytes hide behind them.
 * Twiddles: forward half as fused_kernel (scalar cache / LDS table / registers); the inverse half reads its
 * per-lane group from the SAME LDS table in mirrored order (load_stage_tw MIRROR: w^-1[2^s+j] = -w[2^(s+1)-1-j]),
 * so no second table is needed in LDS; the first inverse group's 12 twiddles are requested while the forward
 * half finishes and reuse the registers of the forward half's last group.
 * Reference primitive this generalises: fast_mul_mod_q (include/internal/fast_mul_operators.h:56-60).
 */
template <class A> struct ProdParams {
  Params<A>              f;      /* forward tables; a = b's coefficients in, nblocks = polynomials */
  const typename A::tw * tw_i;   /* inverse full records (+16 folded N^-1 records) */
  const typename A::ctw *tw8_i;  /* inverse compact */
  const uint64_t *       ahat;   /* fwd(a): canonical, or lazy [0,4q) when a_lazy */
  uint64_t *             out;    /* c; may alias b or ahat */
  uint32_t               a_lazy;
};
/* kernel argument of the product kernels (limb 0's pointers; KArgs::limb_stride separates the limbs of all three slabs) */
template <class A> struct KProd {
  KArgs<A>        f;
  const uint64_t *ahat;
  uint64_t *      out;
  uint32_t        a_lazy;
};
template <class A, bool MULTI> __device__ __forceinline__ ProdParams<A> limb_prod_params(const KProd<A> &k, uint32_t &bid, uint32_t &gdim)
{
  uint32_t      limb;
  ProdParams<A> pp;
  pp.f = limb_params<A, false, MULTI>(k.f, bid, gdim, limb);
  pp.tw_i   = k.f.limbs[limb].tw_i;
  pp.tw8_i  = k.f.limbs[limb].tw8_i;
  pp.ahat   = k.ahat + (uint64_t)limb * k.f.limb_stride;
  pp.out    = k.out + (uint64_t)limb * k.f.limb_stride;
  pp.a_lazy = k.a_lazy;
  return pp;
}

/* WHOLE: the block is the whole polynomial (N = 2^14).  !WHOLE: the blocks of a larger transform (N = 2^15..2^17,
 * pp.f.s0 = log2 N - 14 leading stages done by column passes before and after this launch): the product is
 * element-wise, so it fuses block by block just the same -- per limb col(a), blocks(a), col(b), THIS, col^-1(c):
 * 88N bytes instead of 120N.  A workgroup then always sees the same block position (its stride is a multiple of the
 * blocks per polynomial), whose forward table slice it keeps in LDS; the mirrored read would need the slice of the
 * complementary position, so the inverse half takes that one group's twiddles from global memory instead. */
/* BOTH: pp.ahat holds a's COEFFICIENTS (blocks of a larger product: a after its column passes) and the kernel takes them through the forward stages too -- a^
 * waits, as doubles, in the 32 VGPRs that hold the prefetched a^ words otherwise, so the register budget is the same; a^
 * never exists in memory (24N instead of 40N bytes per product, one launch instead of two) and a is left untouched. */
template <class A, int LOGN, int KSH, bool ALAZY, bool WHOLE, bool MULTI = false, bool BOTH = false>
__global__ void __launch_bounds__((Geom<LOGN, false, 3>::WG), (Geom<LOGN, false, 3>::WPS))
  fused_product_kernel(const KProd<A> kp)
{
  /* (!WHOLE && BOTH: the blocks of a larger product; pp.ahat then holds what a's column passes left, as pf.a does for b) */
  uint32_t            bid, gdim;
  const ProdParams<A> pp = limb_prod_params<A, MULTI>(kp, bid, gdim);
  using P = Plan<LOGN>;
  using G = Geom<LOGN, false, 3>;
  static_assert(A::kCompact && G::BPW == 1 && (LOGN == 14 || LOGN == 12 || (LOGN == 13 && WHOLE)),
                "built for the FP64 policy on blocks of 2^12 and 2^14 points and whole polynomials of 2^13");
  constexpr uint32_t MASKF = fused_mask<A, LOGN, false, KSH>();
  constexpr uint32_t MASKI = fused_mask<A, LOGN, true, KSH>() | (WHOLE ? kLastInvFlag : 0u);
  constexpr int      GL    = P::NG - 1;
  static_assert(G::TBL(GL - 1) > 0 && G::TBL(GL) == 0, "twiddle placement this kernel assumes");
  __shared__ typename A::val lds_all[P::LDS_ELEMS + G::LDS_TW];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + P::LDS_ELEMS);
  const lds_ctw_ptr<A>   ltw  = (lds_ctw_ptr<A>)tabl;
  const uint32_t         tid  = threadIdx.x;
  /* whole polynomials: the stage offset and size are compile-time constants (as run-time kernel arguments they
   * cost scalar registers the two halves do not have) */
  Params<A> pf = pp.f;
  if constexpr(WHOLE) {
    pf.s0   = 0;
    pf.logn = LOGN;
  }
  pf.wide      = 0;
  pf.lazy      = 0;
  Params<A> pi = pf;
  pi.tw                       = pp.tw_i;
  pi.tw8                      = pp.tw8_i;
  pi.lastinv                  = WHOLE ? 1 : 0;
  const uint64_t stride = gdim;
  uint64_t       b      = bid;
  if(b >= pf.nblocks) return;
  /* (the loop's comparisons as scalar subtractions where the two VGPRs of a vector comparison are the ones that spill: `below`) */
  constexpr bool SCMP = BOTH && A::kWide52 && LOGN == 14;
  const uint32_t blk = WHOLE ? 0u : ((uint32_t)b & ((1u << pf.s0) - 1u)); /* the same for every block of this workgroup */
  fill_lds_tables<A, LOGN, false, G>(tabl, pf, blk, tid);
  __syncthreads();
  uint64_t raw[kE];
  prefetch_first<LOGN>(raw, tid, (BOTH ? pp.ahat : pf.a) + blk_off<LOGN>(pf, b));
  pin_raw(raw);
  for(; SCMP ? below(b, pf.nblocks) : b < pf.nblocks; b += stride) {
    /* The two sets of 12 per-lane twiddles (forward half's last group, inverse half's first group) share one set
     * of registers and are therefore requested per block.  They do not depend on the block, so the
     * compiler would hoist both sets (and their 24 lane offsets) out of the loop and spill; the opaque copy of
     * the thread id ties them -- and every other lane-dependent address of the two halves (six exchanges, two
     * prefetches, the stores): hoisted, those were spilled and reloaded from scratch behind the HBM prefetch --
     * to the iteration. */
    uint32_t tl = tid;
    asm volatile("" : "+v"(tl));
    typename A::ctw pre[4][kE / 2];
    preload_group_tw<A, LOGN, GL>(pre, tl, blk, pf);
    /* (52-bit class, BOTH at 2^14: b's words are requested behind a's first stage group instead of in front of it -- the
     * reduce-both-operands butterflies of that group need the registers: 2 spilled VGPRs otherwise) */
    constexpr bool LATE_B = BOTH && A::kWide52 && LOGN == 14;
    const auto forward = [&](typename A::val(&v)[kE], auto late) {
      run_group<A, LOGN, 0, false, MASKF>(v, tl, blk, pf);
      if constexpr(decltype(late)::value) {
        __builtin_amdgcn_sched_barrier(0);
        prefetch_first<LOGN>(raw, tl, pf.a + blk_off<LOGN>(pf, b));
      }
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        exchange<A, LOGN, GI, GI + 1>(v, tl, lds_all);
        if constexpr(GI + 1 == GL) {
          run_group_preloaded<A, LOGN, GL, MASKF>(v, pre, pf);
        } else if constexpr(G::TBL(GI + 1) > 0) {
          run_group<A, LOGN, GI + 1, false, MASKF, true>(v, tl, blk, pf, ltw + G::TBL_OFF(GI + 1));
        } else {
          run_group<A, LOGN, GI + 1, false, MASKF>(v, tl, blk, pf);
        }
      });
    };
    typename A::val x[kE];
    typename A::val xa[BOTH ? kE : 1];
    if constexpr(BOTH) {
      convert_inputs<A, false>(xa, raw, false, pf.c);
      if constexpr(!LATE_B) prefetch_first<LOGN>(raw, tl, pf.a + blk_off<LOGN>(pf, b)); /* b's words travel during a's forward stages */
      forward(xa, std::integral_constant<bool, LATE_B>{});
      /* (b's words are converted after a's last stage, not before: interleaved by the scheduler, x, xa and the raw words
       * lived side by side and spilled) */
      __builtin_amdgcn_sched_barrier(0);
      convert_inputs<A, false>(x, raw, false, pf.c);
    } else {
      convert_inputs<A, false>(x, raw, false, pf.c);
      /* a^ in the last group's layout: requested now, used after the 14 forward stages */
      prefetch_last<LOGN>(raw, tl, pp.ahat + blk_off<LOGN>(pf, b));
    }
    forward(x, std::false_type{});
    /* the inverse's first group: its twiddles land while the product is computed */
    asm volatile("" : "+v"(tl));
    preload_group_tw<A, LOGN, GL>(pre, tl, blk, pi);
    if constexpr(BOTH) {
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::product_rr(x[decltype(ee)::value], xa[decltype(ee)::value], pf.c); });
    } else {
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::template product_in_domain<ALAZY>(x[decltype(ee)::value], raw[decltype(ee)::value], pf.c); });
    }
    /* the next block's loads reuse a^'s registers: not before the last product has read them (interleaved by
     * the scheduler, the two lived side by side and spilled) */
    __builtin_amdgcn_sched_barrier(0);
    {
      const bool     more = SCMP ? below(b + stride, pf.nblocks) : b + stride < pf.nblocks;
      const uint64_t nb   = more ? b + stride : b;
      prefetch_first<LOGN>(raw, tl, (BOTH ? pp.ahat : pf.a) + blk_off<LOGN>(pf, nb), more);
    }
    run_group_preloaded<A, LOGN, GL, MASKI, true>(x, pre, pi);
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int GI = P::NG - 1 - decltype(gg)::value;
      exchange<A, LOGN, GI, GI - 1>(x, tl, lds_all);
      if constexpr(WHOLE && G::TBL(GI - 1) > 0) {
        run_group<A, LOGN, GI - 1, true, MASKI, true, true>(x, tl, blk, pi, ltw + G::TBL_OFF(GI - 1));
      } else {
        /* (per-lane twiddles from global memory in the !WHOLE form: keep their requests behind the exchange --
         * hoisted above it by the scheduler they occupied 30 registers during the previous group and spilled) */
        if constexpr(!WHOLE && G::TBL(GI - 1) > 0) __builtin_amdgcn_sched_barrier(0);
        run_group<A, LOGN, GI - 1, true, MASKI>(x, tl, blk, pi);
      }
    });
    uint64_t out[kE];
    static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = A::store_inv(x[decltype(ee)::value], pf.c); });
    buffer_store_first_raw<LOGN>(out, tl, pp.out + blk_off<LOGN>(pf, b));
  }
}

/* ------------------------------------------------------------------ */
/* products at N = 2^15 .. 2^17: all passes as items of one launch      */
/* ------------------------------------------------------------------ */
/*
 * c = a * b with a^ = fwd(a) already in HBM: the remaining chain -- column stages of b, per block forward x a^ -> inverse,
 * inverse column stages of c -- as the three item kinds of ONE launch in team_kernel's scheme (per-XCD in-order queues,
 * per-polynomial hand-off counters, intermediates kept in the XCD's L2 / Infinity Cache): first-pass items of polynomial j,
 * second-pass items of polynomial j - lag, third-pass items of polynomial j - 2 lag.  An item only ever waits for items
 * handed out earlier in its queue, and first-pass items never wait: no deadlock whatever the residency.  Five launches per
 * 256 MiB chunk become one launch per batch; the fabric carries 40N bytes for this chain instead of 56N while the
 * intermediates stay on chip.  The blocks are 2^12 points at every size (2^17: five column stages in one item, where the
 * per-pass path needs 2^14-point blocks to get by with one column launch).
 * c may alias a or b exactly as in fused_product_kernel: a block's a^ and b words are read by the item that overwrites them.
 */
template <class A, int KSH, int LDAUX_B, int LDAUX_A, int STAUX>
__device__ __forceinline__ void team_product_item(uint64_t *bblk, const uint64_t *ablk, uint64_t *cblk, uint32_t blk, uint32_t tid0,
                                                  const Params<A> &pf, const Params<A> &pi, typename A::val *lds, typename A::ctw *tabl)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, 3>;
  constexpr uint32_t MASKF = fused_mask<A, LOGN, false, KSH>();
  constexpr uint32_t MASKI = fused_mask<A, LOGN, true, KSH>();
  constexpr int      GL    = P::NG - 1;
  static_assert(G::TBL(GL - 1) > 0 && G::TBL(GL) == 0 && !P::WAVE_LOCAL(0, 1), "twiddle placement / barrier this item assumes");
  const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  /* (an opaque copy of the thread id ties every lane-dependent address of the item to the item: hoisted out of the item
   * loop they live in registers -- or scratch -- for the whole launch; fused_product_kernel does the same) */
  uint32_t tl = tid0;
  asm volatile("" : "+v"(tl));
  const uint32_t tid = tl;
  uint64_t raw[kE];
  prefetch_first<LOGN, LDAUX_B>(raw, tid, bblk);
  typename A::ctw pre[4][kE / 2];
  preload_group_tw<A, LOGN, GL>(pre, tid, blk, pf);
  fill_lds_tables<A, LOGN, false, G>(tabl, pf, blk, tid); /* (published by the first exchange's barriers) */
  typename A::val x[kE];
  convert_inputs<A, false>(x, raw, false, pf.c);
  prefetch_last<LOGN, LDAUX_A>(raw, tid, ablk); /* a^ in the last group's layout: used after the twelve forward stages */
  run_group<A, LOGN, 0, false, MASKF>(x, tid, blk, pf);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = decltype(gg)::value;
    exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
    if constexpr(GI + 1 == GL) {
      run_group_preloaded<A, LOGN, GL, MASKF>(x, pre, pf);
    } else if constexpr(G::TBL(GI + 1) > 0) {
      run_group<A, LOGN, GI + 1, false, MASKF, true>(x, tid, blk, pf, ltw + G::TBL_OFF(GI + 1));
    } else {
      run_group<A, LOGN, GI + 1, false, MASKF>(x, tid, blk, pf);
    }
  });
  uint32_t t2 = tid;
  asm volatile("" : "+v"(t2));
  preload_group_tw<A, LOGN, GL>(pre, t2, blk, pi); /* the inverse's first group: lands while the product is computed */
  static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::template product_in_domain<true>(x[decltype(ee)::value], raw[decltype(ee)::value], pf.c); });
  run_group_preloaded<A, LOGN, GL, MASKI, true>(x, pre, pi);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    exchange<A, LOGN, GI, GI - 1>(x, tid, lds);
    if constexpr(G::TBL(GI - 1) > 0) __builtin_amdgcn_sched_barrier(0); /* (as in fused_product_kernel: keep the global twiddle requests behind the exchange) */
    run_group<A, LOGN, GI - 1, true, MASKI>(x, tid, blk, pi);
  });
  uint64_t out[kE];
  static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = A::store_inv(x[decltype(ee)::value], pf.c); });
  buffer_store_first_raw<LOGN, STAUX>(out, tid, cblk);
}

/* The same item with BOTH forward transforms inside (team_product_kernel<..., FOUR = true>): a's block comes in as the
 * intermediate of a's column pass, is taken through the twelve block stages first and waits in 32 VGPRs -- the registers
 * that hold the prefetched a^ words in the item above -- while b's block follows; a^ never exists in memory: 16N bytes
 * fewer across the fabric per product (no write-through store of a^, no read of it) and one launch less. */
template <class A, int KSH, int LDAUX, int STAUX>
__device__ __forceinline__ void team_product_item2(uint64_t *bblk, const uint64_t *ablk, uint64_t *cblk, uint32_t blk, uint32_t tid0,
                                                   const Params<A> &pf, const Params<A> &pi, typename A::val *lds, typename A::ctw *tabl)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, 3>;
  constexpr uint32_t MASKF = fused_mask<A, LOGN, false, KSH>();
  constexpr uint32_t MASKI = fused_mask<A, LOGN, true, KSH>();
  constexpr int      GL    = P::NG - 1;
  static_assert(G::TBL(GL - 1) > 0 && G::TBL(GL) == 0 && !P::WAVE_LOCAL(0, 1), "twiddle placement / barrier this item assumes");
  const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  uint32_t tl = tid0;
  asm volatile("" : "+v"(tl));
  const uint32_t tid = tl;
  uint64_t raw[kE];
  prefetch_first<LOGN, LDAUX>(raw, tid, ablk);
  typename A::ctw pre[4][kE / 2];
  preload_group_tw<A, LOGN, GL>(pre, tid, blk, pf);
  fill_lds_tables<A, LOGN, false, G>(tabl, pf, blk, tid); /* (published by the first exchange's barriers) */
  const auto forward = [&](typename A::val(&x)[kE]) {
    run_group<A, LOGN, 0, false, MASKF>(x, tid, blk, pf);
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int GI = decltype(gg)::value;
      exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
      if constexpr(GI + 1 == GL) {
        run_group_preloaded<A, LOGN, GL, MASKF>(x, pre, pf);
      } else if constexpr(G::TBL(GI + 1) > 0) {
        run_group<A, LOGN, GI + 1, false, MASKF, true>(x, tid, blk, pf, ltw + G::TBL_OFF(GI + 1));
      } else {
        run_group<A, LOGN, GI + 1, false, MASKF>(x, tid, blk, pf);
      }
    });
  };
  typename A::val xa[kE];
  convert_inputs<A, false>(xa, raw, false, pf.c);
  prefetch_first<LOGN, LDAUX>(raw, tid, bblk); /* b's words travel during a's twelve stages */
  forward(xa);
  typename A::val x[kE];
  convert_inputs<A, false>(x, raw, false, pf.c);
  forward(x);
  uint32_t t2 = tid;
  asm volatile("" : "+v"(t2));
  preload_group_tw<A, LOGN, GL>(pre, t2, blk, pi); /* the inverse's first group: lands while the product is computed */
  static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::product_rr(x[decltype(ee)::value], xa[decltype(ee)::value], pf.c); });
  run_group_preloaded<A, LOGN, GL, MASKI, true>(x, pre, pi);
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    exchange<A, LOGN, GI, GI - 1>(x, tid, lds);
    if constexpr(G::TBL(GI - 1) > 0) __builtin_amdgcn_sched_barrier(0);
    run_group<A, LOGN, GI - 1, true, MASKI>(x, tid, blk, pi);
  });
  uint64_t out[kE];
  static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = A::store_inv(x[decltype(ee)::value], pf.c); });
  buffer_store_first_raw<LOGN, STAUX>(out, tid, cblk);
}

struct TeamProdCtl {
  unsigned next[8][32];
  unsigned owner[8][32];
  unsigned done[1]; /* [2][polynomials]: first- and second-pass items finished */
};

template <class A> struct KTeamProd {
  KProd<A>     k;      /* f.a = b, ahat, out = c (limb 0's slabs; f.limb_stride apart); f.nblocks = polynomials PER LIMB */
  TeamProdCtl *ctl;    /* zeroed before the launch */
  uint32_t     lag;
  uint32_t     nlimbs; /* MULTI kernels: limbs of the launch */
  uint64_t     split_rcp;  /* as KTeam::split_rcp */
  uint32_t     poly_major; /* as KTeam::poly_major */
};

template <class A, int LEAD, int KSH, bool FOUR = false, bool MULTI = false>
__global__ void __launch_bounds__(256, 4) team_product_kernel(const KTeamProd<A> kt)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, 3>;
  static_assert(A::kCompact && P::T == kTeamCols && LEAD >= 3 && LEAD <= 5, "FP64 policies, N = 2^15..2^17");
  __shared__ typename A::val lds[P::LDS_ELEMS + G::LDS_TW];
  __shared__ unsigned        s_k, s_k2[2];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds + P::LDS_ELEMS);
  const uint32_t         tid  = threadIdx.x;
  uint32_t               bid_, gdim_;
  const ProdParams<A>    pp = limb_prod_params<A, false>(kt.k, bid_, gdim_);
  Params<A>              pf = pp.f;
  pf.s0   = LEAD;
  pf.wide = 0;
  pf.lazy = 0;
  Params<A> pi = pf;
  pi.tw        = pp.tw_i;
  pi.tw8       = pp.tw8_i;
  pi.lastinv   = 1;
  constexpr uint32_t CMASKF = column_mask<A, LEAD, false, KSH>();
  constexpr uint32_t CMASKI = column_mask<A, LEAD, true, KSH>();
  constexpr bool     MID_LAZY = !A::kTracksBounds;
  const uint32_t logn  = LOGN + LEAD;
  const uint32_t batch = (uint32_t)pf.nblocks; /* polynomials per limb */
  const uint32_t total = MULTI ? batch * kt.nlimbs : batch;
  constexpr uint32_t NCOL = 1u << (LOGN - 8), NROW = 1u << LEAD;
  /* FOUR: the first pass takes the column tiles of BOTH operands (b's, then a's) and the product item transforms both blocks */
  constexpr uint32_t NFIRST = FOUR ? 2u * NCOL : NCOL;
  TeamProdCtl *const ctl = kt.ctl;
  const uint32_t lag = kt.lag;
  const uint32_t my  = xcc_id();
  uint64_t       loff = 0; /* the item's limb: word offset of its slabs (MULTI) */
  for(uint32_t qq = 0; qq < 8; qq++) {
    const uint32_t q = (my + qq) & 7u;
    if(tid == 0) {
      const unsigned prev = atomicCAS(&ctl->owner[q][0], 0u, my + 1u);
      s_k                 = (prev == 0u || prev == my + 1u) ? 1u : 0u;
    }
    __syncthreads();
    const bool mine = s_k != 0;
    __syncthreads();
    if(!mine) continue;
    constexpr uint32_t kNoSignal = 0xffffffffu;
    uint32_t           sig       = kNoSignal; /* index into done[]: polynomial, or total + polynomial for the second pass */
    for(uint32_t it = 0;; it ^= 1u) {
      /* (lane-0 blocks are followed at once by a workgroup barrier: see team_kernel) */
      if(tid == 0) {
        if(sig != kNoSignal) __hip_atomic_fetch_add(&ctl->done[sig], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_k2[it] = atomicAdd(&ctl->next[q][0], 1u);
      }
      sig = kNoSignal;
      __syncthreads();
      const TeamItem ti = team_decode(uniform_u32(s_k2[it]), q, total, lag, NFIRST, NROW, NCOL);
      if(ti.stop) break;
      if(!ti.valid) continue;
      const uint32_t pass = ti.pass, item = ti.item, pidx = ti.v;
      uint32_t       pl   = pidx; /* the polynomial inside its limb */
      if constexpr(MULTI) {
        uint32_t limb;
        team_split(pidx, batch, kt.nlimbs, kt.poly_major, kt.split_rcp, limb, pl);
        loff                = (uint64_t)limb * kt.k.f.limb_stride;
        const LimbRec<A> &r = kt.k.f.limbs[limb];
        pf.tw  = r.tw_f;
        pf.tw8 = r.tw8_f;
        pf.c   = r.c;
        pi.tw  = r.tw_i;
        pi.tw8 = r.tw8_i;
        pi.c   = r.c;
      }
      if(pass > 0) {
        const uint32_t need = pass == 1 ? NFIRST : NROW;
        const uint32_t slot = pass == 1 ? pidx : total + pidx;
        if(tid == 0) {
          while(__hip_atomic_load(&ctl->done[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
      }
      const uint64_t  poff  = loff + (uint64_t)pl * pf.pstride;
      uint64_t *      bpoly = pf.a + poff;
      const uint64_t *apoly = pp.ahat + poff;
      uint64_t *      cpoly = pp.out + poff;
      /* NTT_TEAMPROD_ONLY (diagnostic builds, like NTT_STAMPS; never in the shipped library): which item types do their work -- bit 0
       * b's column items, 3 a's column items, 1 the block products, 2 c's inverse column items; the others only run the queue protocol.
       * Wrong results; one --pmc pass per build gives an item type's FETCH / WRITE bytes by themselves (profiles/r06/config5_bytes_by_item.txt) */
#ifndef NTT_TEAMPROD_ONLY
#  define NTT_TEAMPROD_ONLY 15
#endif
      constexpr uint32_t kOnly = NTT_TEAMPROD_ONLY;
      if(pass == 0) {
        uint64_t *const src = FOUR && item >= NCOL ? const_cast<uint64_t *>(apoly) : bpoly; /* (a is an operand buffer of the caller's: written here) */
        if((kOnly & 9u) == 9u || (kOnly & (FOUR && item >= NCOL ? 8u : 1u)) != 0)
          team_column_item<A, LEAD, false, CMASKF, kAuxSc0Sc1, 0>(src, (item & (NCOL - 1u)) * kTeamCols + tid, logn, pf, MID_LAZY);
      } else if((kOnly & (pass == 1 ? 2u : 4u)) == 0) {
        /* (switched off in this diagnostic build) */
      } else if(pass == 1) {
        if constexpr(FOUR) {
          /* (NTT_TEAMPROD_FAKEBLK, diagnostic builds: every block product reads block 0's twiddles -- wrong results, the same loads,
           * all of them L2-hot: what the twiddle reads of the product items cost at the fabric) */
#ifndef NTT_TEAMPROD_FAKEBLK
#  define NTT_TEAMPROD_FAKEBLK 0
#endif
          team_product_item2<A, KSH, kAuxNt, 0>(bpoly + ((uint64_t)item << LOGN), apoly + ((uint64_t)item << LOGN),
                                                cpoly + ((uint64_t)item << LOGN), NTT_TEAMPROD_FAKEBLK ? 0u : item, tid, pf, pi, lds, tabl);
        } else {
          team_product_item<A, KSH, kAuxNt, kAuxNt, 0>(bpoly + ((uint64_t)item << LOGN), apoly + ((uint64_t)item << LOGN),
                                                         cpoly + ((uint64_t)item << LOGN), item, tid, pf, pi, lds, tabl);
        }
      } else {
        team_column_item<A, LEAD, true, CMASKI, kAuxNt, kAuxSc1>(cpoly, item * kTeamCols + tid, logn, pi, false);
      }
      if(pass < 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sig = pass == 0 ? pidx : total + pidx;
      }
    }
  }
}

/* The same product for whole polynomials of 2^8 .. 2^11 points, where a 256-thread workgroup holds several blocks
 * (Geom::BPW) that share the LDS twiddle tables: the plain (non-persistent) loop of fused_kernel's small-block path with
 * the product and the inverse half appended.  Every per-lane group has its forward table in LDS at these sizes, and the
 * inverse half reads all of them mirrored, so the kernel issues no per-lane global twiddle load at all; a^ arrives in
 * the last group's layout as 16-byte loads.  40N bytes per product instead of 72N. */
template <class A, int LOGN, int KSH, bool MULTI = false, bool BOTH = false>
__global__ void __launch_bounds__((Geom<LOGN, false, 3>::WG), (Geom<LOGN, false, 3>::WPS))
  fused_product_small_kernel(const KProd<A> kp)
{
  uint32_t            bid, gdim;
  const ProdParams<A> pp = limb_prod_params<A, MULTI>(kp, bid, gdim);
  using P = Plan<LOGN>;
  using G = Geom<LOGN, false, 3>;
  static_assert(A::kCompact && G::BPW > 1 && LOGN >= 8 && LOGN <= 11, "whole polynomials of 2^8..2^11 points");
  constexpr uint32_t MASKF = fused_mask<A, LOGN, false, KSH>();
  constexpr uint32_t MASKI = fused_mask<A, LOGN, true, KSH>() | kLastInvFlag;
  __shared__ typename A::val lds_all[G::BPW * P::LDS_ELEMS + G::LDS_TW];
  const uint32_t         tid = threadIdx.x;
  const uint32_t         sub = tid >> P::LT;
  const uint32_t         t   = tid & (P::T - 1);
  typename A::val *const lds = lds_all + sub * P::LDS_ELEMS;
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
  const lds_ctw_ptr<A>   gtw  = (lds_ctw_ptr<A>)tabl;
  Params<A> pf = pp.f;
  pf.s0        = 0;
  pf.logn      = LOGN;
  pf.wide      = 0;
  pf.lazy      = 0;
  Params<A> pi = pf;
  pi.tw        = pp.tw_i;
  pi.tw8       = pp.tw8_i;
  pi.lastinv   = 1;
  fill_lds_tables<A, LOGN, false, G>(tabl, pf, 0u, tid);
  __syncthreads();
  for(uint64_t b0 = (uint64_t)bid * G::BPW; b0 < pf.nblocks; b0 += (uint64_t)gdim * G::BPW) {
    uint64_t   b    = b0 + sub;
    const bool live = b < pf.nblocks;
    if(!live) b = pf.nblocks - 1; /* idle sub-blocks shadow a real polynomial (barriers are workgroup-wide), never store */
    const auto forward = [&](typename A::val(&v)[kE]) {
      run_group<A, LOGN, 0, false, MASKF, (G::TBL(0) > 0)>(v, t, 0u, pf, gtw);
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        exchange<A, LOGN, GI, GI + 1>(v, t, lds);
        run_group<A, LOGN, GI + 1, false, MASKF, (G::TBL(GI + 1) > 0)>(v, t, 0u, pf, gtw + G::TBL_OFF(GI + 1));
      });
    };
    typename A::val x[kE];
    uint64_t        raw[kE];
    if constexpr(BOTH) {
      /* a's coefficients through the same forward stages first; a^ waits in registers (the ones a^'s words occupy otherwise) */
      typename A::val xa[kE];
      global_load_first<A, LOGN, false>(xa, t, pp.ahat + blk_off<LOGN>(pf, b), false, pf.c);
      prefetch_first<LOGN>(raw, t, pf.a + blk_off<LOGN>(pf, b));
      forward(xa);
      convert_inputs<A, false>(x, raw, false, pf.c);
      forward(x);
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::product_rr(x[decltype(ee)::value], xa[decltype(ee)::value], pf.c); });
    } else {
      global_load_first<A, LOGN, false>(x, t, pf.a + blk_off<LOGN>(pf, b), false, pf.c);
      prefetch_last<LOGN>(raw, t, pp.ahat + blk_off<LOGN>(pf, b));
      forward(x);
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = A::template product_in_domain<true>(x[decltype(ee)::value], raw[decltype(ee)::value], pf.c); });
    }
    constexpr int GL = P::NG - 1;
    run_group<A, LOGN, GL, true, MASKI, (G::TBL(GL) > 0), (G::TBL(GL) > 0)>(x, t, 0u, pi, gtw + G::TBL_OFF(GL));
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int GI = P::NG - 1 - decltype(gg)::value;
      exchange<A, LOGN, GI, GI - 1>(x, t, lds);
      run_group<A, LOGN, GI - 1, true, MASKI, (G::TBL(GI - 1) > 0), (G::TBL(GI - 1) > 0)>(x, t, 0u, pi, gtw + G::TBL_OFF(GI - 1));
    });
    if(live) global_store_first<A, LOGN, true>(x, t, pp.out + blk_off<LOGN>(pf, b), pf.c, false);
  }
}

/* ------------------------------------------------------------------ */
/* products of operands that ARE in the NTT domain                      */
/* ------------------------------------------------------------------ */
/*
 * c = inv( sum_{i<k} a_i^ (.) b_i^ ): the other half of SURVEY 8(f) f1 ("fusing the multiply into the inverse's first load
 * saves 16N bytes").  Keys, ciphertexts and plaintexts of an FHE caller live in the NTT domain; what it issues is the
 * pointwise product of two transformed operands (k = 1) or the inner product of a digit-decomposed ciphertext with a
 * key (key switching, k = 2 .. tens) followed by ONE inverse transform.  This kernel is the inverse block kernel with its
 * input conversion replaced: where fused_kernel<.., INV> turns the 16 words of a thread into values, this one reads the 16
 * words of every a_i^ and b_i^ in the same layout (the product is element-wise, so the inverse's first group's layout
 * serves), forms the k products and their sum in registers (A::dot_term / dot_acc / dot_fold) and runs the inverse stages
 * on the sum.  HBM traffic: 16kN bytes in, 8N out -- 24N for a plain product instead of 40N for pointwise + inverse, no
 * intermediate ever written; with B_BCAST the b_i^ are ONE polynomial each, shared by the batch (a key: read from the
 * L2), and the traffic is 8kN + 8N.  Blocks of a larger transform (LASTINV = false: N > 2^14, the column stages of the
 * inverse follow as launches of their own) work the same way: the product rides in the first pass of the inverse.
 * Reference primitive this generalises: fast_mul_mod_q (include/internal/fast_mul_operators.h:56-60).
 */
constexpr int kMaxDot = 32; /* operand pairs of one launch (2 x 32 pointers in the kernel arguments) */

template <class A> struct KDot {
  KArgs<A>        k;             /* k.a = c (limb 0), nblocks / s0 / logn as for an inverse block pass */
  uint32_t        npairs;        /* 1 .. kMaxDot */
  uint32_t        lazy_in;       /* operand words may be lazy: anywhere in [0,4q) */
  uint32_t        b_bcast;       /* every b_i^ is one polynomial per limb, shared by the whole batch */
  uint64_t        b_limb_stride; /* words between consecutive limbs of a b operand */
  const uint64_t *a[kMaxDot];
  const uint64_t *b[kMaxDot];
};

/* The b operand of a pair goes through the caches (no nt hint): when ONE polynomial serves the whole batch (a key) every
 * block re-reads it and it must stay in the L2 -- with nt loads the broadcast form measured no faster than the
 * per-polynomial one (profiles/r04/domain_bench_first.txt).  The hint is an instruction bit, and a run-time branch between
 * two sets of loads makes the register allocator keep both sets apart (77 spilled VGPRs): one policy for both forms. */
constexpr int kDotAuxB = 0;
/* tuning knobs of the persistent loop (A/B builds: tools/build_tu_variant.sh) */
#ifndef NTT_DOT_AUX_A
#  define NTT_DOT_AUX_A 0 /* cache policy of the a operand's loads: plain, like b's (measured +3 % over nt at k = 1, +10 % with a broadcast key at k = 8: profiles/r04/ab_dot.txt) */
#endif
#ifndef NTT_DOT_A_AT
#  define NTT_DOT_A_AT 1 /* the next block's a words are requested behind the exchange into this group (1 = the last one) */
#endif
#ifndef NTT_DOT_LOOP_CHUNK
#  define NTT_DOT_LOOP_CHUNK 2 /* products in flight inside the pair loop */
#endif
template <int LOGN> __device__ __forceinline__ void prefetch_last_b(uint64_t (&raw)[kE], uint32_t t, const uint64_t *blk, bool live = true)
{
  prefetch_last<LOGN, kDotAuxB>(raw, t, blk, live);
}

template <class A, int LOGN, int KSH, bool LASTINV, bool MULTI = false>
__global__ void __launch_bounds__((Geom<LOGN, true, flavor_of<A>()>::WG), (Geom<LOGN, true, flavor_of<A>()>::WPS)) dot_inv_kernel(const KDot<A> kd)
{
  uint32_t        bid, gdim, limb;
  const Params<A> p = limb_params<A, true, MULTI>(kd.k, bid, gdim, limb);
  using P = Plan<LOGN>;
  using G = Geom<LOGN, true, flavor_of<A>()>;
  constexpr uint32_t MASK   = fused_mask<A, LOGN, true, KSH>() | (LASTINV ? kLastInvFlag : 0u);
  constexpr int      LDS_TW = G::LDS_TW;
  __shared__ typename A::val lds_all[G::BPW * P::LDS_ELEMS + LDS_TW];
  const uint32_t   tid   = threadIdx.x;
  const uint32_t   sub   = tid >> P::LT;
  const uint32_t   t     = tid & (P::T - 1);
  typename A::val *lds   = lds_all + sub * P::LDS_ELEMS;
  const uint32_t   bmask = (1u << p.s0) - 1u;
  const uint32_t   np    = kd.npairs;
  const bool       lazy  = kd.lazy_in != 0;
  const bool       bc    = kd.b_bcast != 0;
  const uint64_t   aoff  = (uint64_t)limb * kd.k.limb_stride; /* (MULTI off: limb == 0, both offsets fold away) */
  const uint64_t   boff  = (uint64_t)limb * kd.b_limb_stride;

  /* (MULTI -- several limbs in one launch -- exists for batches that cannot fill the chip: a workgroup sees one or two blocks,
   * there is nothing to prefetch across, and the plain loop below needs fewer registers next to the run-time limb's constants) */
  if constexpr(G::PERSISTENT && A::kCompact && !MULTI) {
    static_assert(G::BPW == 1, "the persistent inverse loop owns one block per workgroup");
    constexpr int  GL  = P::NG - 1;
    constexpr bool LTW = LDS_TW > 0;
    const uint64_t stride = gdim;
    uint64_t       b      = bid;
    if(b >= p.nblocks) return;
    typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + P::LDS_ELEMS);
    const lds_ctw_ptr<A>   ltw  = (lds_ctw_ptr<A>)tabl;
    if constexpr(LTW) {
      fill_lds_tables<A, LOGN, true>(tabl, p, (uint32_t)b & bmask, tid);
      __syncthreads();
    }
    /* Register budget (2^14: 128 VGPRs at four waves per SIMD).  The transform kernel keeps the first executed group's
     * twelve per-lane twiddles resident (24 VGPRs) next to ONE prefetched block (32); here the next block's FIRST PAIR is
     * two blocks of words (64), so the twiddles are requested per block instead (from the L2, in front of the products that
     * hide their latency) and the prefetch is issued after the last exchange, when the LDS addresses and the per-lane
     * twiddles of the middle groups are dead: the words then have the last group, the stores and the next block's
     * twiddle request to arrive. */
    constexpr bool IPRE = stage_is_compact<A, LOGN, true>(GL, 0) && P::R(GL) < 4 && G::TBL(GL) == 0;
    uint64_t ra[kE], rb[kE];
    prefetch_last<LOGN, NTT_DOT_AUX_A>(ra, tid, kd.a[0] + aoff + blk_off<LOGN>(p, b));
    prefetch_last_b<LOGN>(rb, tid, kd.b[0] + boff + (bc ? ((b & bmask) << LOGN) : blk_off<LOGN>(p, b)));
    pin_raw(ra);
    pin_raw(rb);
    for(; b < p.nblocks; b += stride) {
      const uint32_t blk  = (uint32_t)b & bmask;
      const uint64_t offa = blk_off<LOGN>(p, b);                    /* operands a_i^ and c: the launch's layout */
      const uint64_t offb = bc ? ((uint64_t)blk << LOGN) : offa;    /* a broadcast b_i^ is one dense polynomial */
      uint64_t *     base = p.a + offa;
      /* (an opaque copy of the thread id ties the per-block twiddle request and every lane-dependent address to the
       * iteration: hoisted, they would stay in registers -- or scratch -- for the whole launch; see fused_product_kernel) */
      uint32_t tl = tid;
      asm volatile("" : "+v"(tl));
      typename A::val x[kE];
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = typename A::val{}; });
      /* every pair but the last: add its products, request the next pair */
#pragma unroll 1
      for(uint32_t i = 0; i + 1 < np; i++) {
        if(i != 0 && i % (uint32_t)A::kDotEvery == 0) dot_fold_tile<A>(x, p.c);
        dot_tile<A, 0, kE, NTT_DOT_LOOP_CHUNK>(x, ra, rb, lazy, p.c);
        prefetch_last<LOGN, NTT_DOT_AUX_A>(ra, tl, kd.a[i + 1] + aoff + offa);
        prefetch_last_b<LOGN>(rb, tl, kd.b[i + 1] + boff + offb);
        sched_fence();
      }
      if(np > 1 && (np - 1) % (uint32_t)A::kDotEvery == 0) dot_fold_tile<A>(x, p.c);
      /* the last pair.  The first executed group's twiddles are requested half way through its products -- which hide
       * most of their L2 latency --, when half of the 64 registers of words are free again (any earlier and they would
       * have to live next to all of them) */
      dot_tile<A, 0, kE / 2>(x, ra, rb, lazy, p.c);
      typename A::ctw pre[4][kE / 2];
      if constexpr(IPRE) preload_group_tw<A, LOGN, GL>(pre, tl, blk, p);
      dot_tile<A, kE / 2, kE>(x, ra, rb, lazy, p.c);
      if(np > 1) dot_fold_tile<A>(x, p.c);
      if constexpr(IPRE) {
        run_group_preloaded<A, LOGN, GL, MASK, true>(x, pre, p);
      } else if constexpr(G::TBL(GL) > 0) {
        run_group<A, LOGN, GL, true, MASK, true>(x, tl, blk, p, ltw + G::TBL_OFF(GL));
      } else {
        run_group<A, LOGN, GL, true, MASK>(x, tl, blk, p);
      }
      const bool     more = b + stride < p.nblocks;
      const uint64_t nb   = more ? b + stride : b;
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = P::NG - 1 - decltype(gg)::value;
        exchange<A, LOGN, GI, GI - 1>(x, tl, lds_all);
        if constexpr(GI == NTT_DOT_A_AT) {
          /* the next block's first pair, operand a: always issued (a dead descriptor moves no data past the end) */
          uint32_t t2 = tid;
          asm volatile("" : "+v"(t2));
          sched_fence();
          prefetch_last<LOGN, NTT_DOT_AUX_A>(ra, t2, kd.a[0] + aoff + blk_off<LOGN>(p, nb), more);
          sched_fence();
        }
        if constexpr(G::TBL(GI - 1) > 0) {
          run_group<A, LOGN, GI - 1, true, MASK, true>(x, tl, blk, p, ltw + G::TBL_OFF(GI - 1));
        } else {
          run_group<A, LOGN, GI - 1, true, MASK>(x, tl, blk, p);
        }
      });
      {
        /* ... operand b: behind the last group's butterflies, whose temporaries do not fit next to 64 registers of words */
        uint32_t t3 = tid;
        asm volatile("" : "+v"(t3));
        sched_fence();
        prefetch_last_b<LOGN>(rb, t3, kd.b[0] + boff + (bc ? ((nb & bmask) << LOGN) : blk_off<LOGN>(p, nb)), more);
        sched_fence();
      }
      uint64_t out[kE];
      static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = out_word<A, true, false>(x[decltype(ee)::value], !LASTINV, p.c); });
      buffer_store_first_raw<LOGN>(out, tl, base);
    }
    return;
  } else {
    /* small blocks (several per workgroup, sharing the LDS tables), the integer policies, several limbs: the plain loop */
    const lds_ctw_ptr<A> gtw = (lds_ctw_ptr<A>)reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
    if constexpr(LDS_TW > 0) {
      /* blocks below 2^12 are whole polynomials; larger ones may be blocks of a bigger transform: a workgroup then always
       * sees the same block position (its stride is a multiple of the blocks per polynomial: launch_dot_blocks) */
      fill_lds_tables<A, LOGN, true>(reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS), p,
                                     G::BPW == 1 ? ((uint32_t)bid & bmask) : 0u, tid);
      __syncthreads();
    }
    for(uint64_t b0 = (uint64_t)bid * G::BPW; b0 < p.nblocks; b0 += (uint64_t)gdim * G::BPW) {
      uint64_t   b    = b0 + sub;
      const bool live = b < p.nblocks;
      if(!live) b = p.nblocks - 1; /* idle sub-blocks shadow a real block (barriers are workgroup-wide), never store */
      const uint32_t blk  = (uint32_t)b & bmask;
      const uint64_t offa = blk_off<LOGN>(p, b);
      const uint64_t offb = bc ? ((uint64_t)blk << LOGN) : offa;
      uint64_t *     base = p.a + offa;
      typename A::val x[kE];
      static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = typename A::val{}; });
#pragma unroll 1
      for(uint32_t i = 0; i < np; i++) {
        if(i != 0 && i % (uint32_t)A::kDotEvery == 0) dot_fold_tile<A>(x, p.c);
        /* half a tile at a time: 32 registers of words next to the 32 running sums */
        static_for<0, 2>([&](auto hh) {
          constexpr int H = decltype(hh)::value;
          uint64_t      ra[kE], rb[kE];
          sched_fence();
          load_last_raw<LOGN, 8 * H, 8 * H + 8>(ra, t, kd.a[i] + aoff + offa);
          load_last_raw<LOGN, 8 * H, 8 * H + 8>(rb, t, kd.b[i] + boff + offb);
          dot_tile<A, 8 * H, 8 * H + 8>(x, ra, rb, lazy, p.c);
        });
      }
      if(np > 1) dot_fold_tile<A>(x, p.c);
      run_group<A, LOGN, P::NG - 1, true, MASK, (G::TBL(P::NG - 1) > 0)>(x, t, blk, p, gtw + G::TBL_OFF(P::NG - 1));
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = P::NG - 1 - decltype(gg)::value;
        exchange<A, LOGN, GI, GI - 1>(x, t, lds);
        run_group<A, LOGN, GI - 1, true, MASK, (G::TBL(GI - 1) > 0)>(x, t, blk, p, gtw + G::TBL_OFF(GI - 1));
      });
      /* (a pass that does not end the transform keeps the integer policies' lazy range, as fused_kernel's does) */
      if(live) global_store_first<A, LOGN, true>(x, t, base, p.c, !LASTINV);
    }
  }
}

/* ------------------------------------------------------------------ */
/* NTT-domain products at N = 2^15 .. 2^17 as items of ONE launch       */
/* ------------------------------------------------------------------ */
/*
 * c = inv( sum_i a_i^ (.) b_i^ ) for polynomials larger than a block: per 128 / 256 MiB chunk the library used to launch
 * dot_inv_kernel over the blocks (the products ride in the inverse's first pass) and then the inverse's column pass -- at
 * 4 GB per operand sixty launches of some 60 us, every one with its own ramp and tail.  Here both passes are the ITEMS of one
 * persistent launch in team_kernel's scheme (per-XCD in-order queues, a per-polynomial counter between the passes, a later
 * pass `lag` polynomials behind: see team_kernel for the protocol and its memory-order invariant):
 *   pass 0  row item   : one 2^12-point block -- the k operand pairs' words of the block, products and their sum in registers
 *                        (A::dot_term / dot_acc / dot_fold), the twelve inverse block stages, intermediate words to c;
 *   pass 1  column item: 256 adjacent columns of c through the LEAD leading inverse stages, N^-1 folded in, final stores.
 * The row item reads its operands' block and writes c's block at the same position, so c may alias an operand for k = 1
 * exactly as in dot_inv_kernel (the item that overwrites a line is the only one that ever read it).  A broadcast b_i^ (one
 * polynomial per limb shared by the batch) is read from the L2 by every item.
 * Reference primitive: fast_mul_mod_q (include/internal/fast_mul_operators.h:56-60) in front of inv_ntt_* (src/ntt_reference.c:33-66).
 */
template <class A> struct KTeamDot {
  KDot<A>  d;       /* d.k.a = c (limb 0), d.k.nblocks = polynomials PER LIMB; operand pointers, flags, strides as for dot_inv_kernel */
  TeamCtl *ctl;     /* zeroed before the launch */
  uint64_t split_rcp;
  uint32_t lag, nlimbs, poly_major;
};

/* row item: block `blk` of one polynomial; offa / offb = word offsets of the block inside the a-like operands (and c) / the b operands */
template <class A, int KSH>
__device__ __forceinline__ void team_dot_row_item(uint64_t *cblk, uint64_t offa, uint64_t offb, uint32_t blk, uint32_t tid0, const Params<A> &p,
                                                  const KDot<A> &kd, uint64_t aoff, uint64_t boff, typename A::val *lds, typename A::ctw *tabl)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, true, flavor_of<A>()>;
  constexpr int GL   = P::NG - 1;
  constexpr uint32_t MASK = fused_mask<A, LOGN, true, KSH>(); /* not the pass that ends the transform; inputs are products, not canonical words */
  constexpr bool     MID_LAZY = !A::kTracksBounds;
  /* (an opaque copy of the thread id ties every lane-dependent address to the item: see team_product_item) */
  uint32_t tl = tid0;
  asm volatile("" : "+v"(tl));
  const uint32_t tid  = tl;
  const uint32_t np   = kd.npairs;
  const bool     lazy = kd.lazy_in != 0;
  [[maybe_unused]] const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
  constexpr bool IPRE = A::kCompact && stage_is_compact<A, LOGN, true>(GL, 0) && P::R(GL) < 4 && G::TBL(GL) == 0 && KSH != 1;
  [[maybe_unused]] typename A::ctw pre[4][kE / 2];
  if constexpr(A::kCompact) fill_lds_tables<A, LOGN, true>(tabl, p, blk, tid);
  typename A::val x[kE];
  static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = typename A::val{}; });
#pragma unroll 1
  for(uint32_t i = 0; i < np; i++) {
    if(i != 0 && i % (uint32_t)A::kDotEvery == 0) dot_fold_tile<A>(x, p.c);
    /* half a tile at a time: 32 registers of words next to the 32 running sums (dot_inv_kernel's plain loop) */
    static_for<0, 2>([&](auto hh) {
      constexpr int H = decltype(hh)::value;
      uint64_t      ra[kE], rb[kE];
      sched_fence();
      load_last_raw<LOGN, 8 * H, 8 * H + 8, false>(ra, tid, kd.a[i] + aoff + offa);
      load_last_raw<LOGN, 8 * H, 8 * H + 8>(rb, tid, kd.b[i] + boff + offb);
      dot_tile<A, 8 * H, 8 * H + 8>(x, ra, rb, lazy, p.c);
    });
  }
  if(np > 1) dot_fold_tile<A>(x, p.c);
  /* (the first group's twiddles: requested behind the products, whose 64 registers of words are free again) */
  if constexpr(A::kCompact && IPRE) preload_group_tw<A, LOGN, GL>(pre, tid, blk, p);
  if constexpr(A::kCompact) __syncthreads(); /* the LDS table is read after a wave-local exchange: it needs a barrier of its own */
  if constexpr(A::kCompact && IPRE) {
    run_group_preloaded<A, LOGN, GL, MASK, true>(x, pre, p);
  } else {
    run_group<A, LOGN, GL, true, MASK>(x, tid, blk, p);
  }
  static_for<0, P::NG - 1>([&](auto gg) {
    constexpr int GI = P::NG - 1 - decltype(gg)::value;
    exchange<A, LOGN, GI, GI - 1>(x, tid, lds);
    if constexpr(A::kCompact && G::TBL(GI - 1) > 0) {
      run_group<A, LOGN, GI - 1, true, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI - 1));
    } else {
      run_group<A, LOGN, GI - 1, true, MASK>(x, tid, blk, p);
    }
  });
  uint64_t out[kE];
  static_for<0, kE>([&](auto ee) { out[decltype(ee)::value] = out_word<A, true, false>(x[decltype(ee)::value], MID_LAZY, p.c); });
  buffer_store_first_raw<LOGN, 0>(out, tid, cblk);
}

template <class A, int LEAD, int KSH, bool MULTI = false>
__global__ void __launch_bounds__(256, 4) team_dot_kernel(const KTeamDot<A> kt)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, true, flavor_of<A>()>;
  static_assert((A::kCompact || A::kIntWide) && P::T == kTeamCols && LEAD >= 3 && LEAD <= 5,
                "built for the FP64 policies and the wide integer policy on 2^12-point blocks, N = 2^15..2^17");
  __shared__ typename A::val lds[P::LDS_ELEMS + G::LDS_TW];
  __shared__ unsigned        s_k, s_k2[2];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds + P::LDS_ELEMS);
  const uint32_t         tid  = threadIdx.x;
  uint32_t               bid_, gdim_, limb_;
  Params<A>              p = limb_params<A, true, false>(kt.d.k, bid_, gdim_, limb_);
  p.s0                     = LEAD;
  constexpr uint32_t CMASK = column_mask<A, LEAD, true, KSH>();
  const uint32_t logn  = LOGN + LEAD;
  const uint32_t batch = (uint32_t)p.nblocks; /* polynomials per limb */
  const uint32_t total = MULTI ? batch * kt.nlimbs : batch;
  constexpr uint32_t NCOL = 1u << (LOGN - 8), NROW = 1u << LEAD;
  TeamCtl *const ctl = kt.ctl;
  const uint32_t lag = kt.lag;
  const uint32_t my  = xcc_id();
  const bool     bc  = kt.d.b_bcast != 0;
  uint64_t       aoff = 0, boff = 0; /* the item's limb: word offsets of its slabs (MULTI) */
  for(uint32_t qq = 0; qq < 8; qq++) {
    const uint32_t q = (my + qq) & 7u;
    if(tid == 0) {
      const unsigned prev = atomicCAS(&ctl->owner[q][0], 0u, my + 1u);
      s_k                 = (prev == 0u || prev == my + 1u) ? 1u : 0u;
    }
    __syncthreads();
    const bool mine = s_k != 0;
    __syncthreads();
    if(!mine) continue;
    constexpr uint32_t kNoSignal = 0xffffffffu;
    uint32_t           sig       = kNoSignal;
    for(uint32_t it = 0;; it ^= 1u) {
      /* (lane-0 blocks are followed at once by a workgroup barrier: see team_kernel) */
      if(tid == 0) {
        if(sig != kNoSignal) __hip_atomic_fetch_add(&ctl->done[sig], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_k2[it] = atomicAdd(&ctl->next[q][0], 1u);
      }
      sig = kNoSignal;
      __syncthreads();
      const TeamItem ti = team_decode(uniform_u32(s_k2[it]), q, total, lag, NROW, NCOL, 0u);
      if(ti.stop) break;
      if(!ti.valid) continue;
      const bool     second = ti.pass != 0;
      const uint32_t item = ti.item, pidx = ti.v;
      uint32_t       pl   = pidx;
      if constexpr(MULTI) {
        uint32_t limb;
        team_split(pidx, batch, kt.nlimbs, kt.poly_major, kt.split_rcp, limb, pl);
        team_limb<A, true>(p, kt.d.k, limb);
        aoff = (uint64_t)limb * kt.d.k.limb_stride;
        boff = (uint64_t)limb * kt.d.b_limb_stride;
      }
      const uint64_t poff = (uint64_t)pl * p.pstride; /* the polynomial inside its limb: c and every operand laid out like it */
      uint64_t *     poly = p.a + poff;
      if(second) {
        if(tid == 0) {
          while(__hip_atomic_load(&ctl->done[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NROW) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
        team_column_item<A, LEAD, true, CMASK, kAuxNt, kAuxSc1>(poly, item * kTeamCols + tid, logn, p, false);
      } else {
        const uint64_t offa = poff + ((uint64_t)item << LOGN);
        const uint64_t offb = bc ? ((uint64_t)item << LOGN) : offa;
        team_dot_row_item<A, KSH>(poly + ((uint64_t)item << LOGN), offa, offb, item, tid, p, kt.d, aoff, boff, lds, tabl);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sig = pidx;
      }
    }
  }
}

/* ------------------------------------------------------------------ */
/* forward transform with the product at its output: c^ = fwd(a) (.) b^ (+ c^) */
/* ------------------------------------------------------------------ */
/*
 * The counterpart of dot_inv_kernel on the other side of the path: the operand comes in as coefficients, the result STAYS in
 * the NTT domain -- a plaintext or key factor b^ kept transformed is multiplied in where the forward block kernel would
 * reduce and store its outputs, optionally added to what c^ already holds (the multiply-accumulate of a key-switching inner
 * product, digit by digit).  24N bytes (16N with a broadcast b^) instead of 40N for forward transform + pointwise product;
 * with the accumulator 32N (24N) instead of 48N.  The forward transform ends in the layout the element-wise product needs
 * (runs of four consecutive coefficients per lane), so b^ and c^ are read and c^ written as 16-byte words by the lane that
 * owns them: c may alias a or b^.  Blocks of a larger transform (N > 2^14: the block pass is the forward transform's LAST
 * pass) work the same way.  Reference primitive: fast_mul_mod_q (include/internal/fast_mul_operators.h:56-60).
 * Registers (2^14: 128 VGPRs): the next block's words are in flight during the whole iteration (32), so the last group's
 * twiddles are requested per block instead of staying resident (24), b^ is fetched in two halves -- the first one in front of
 * the last group, the second one while the first half's products run -- and the accumulator words right where each half is
 * finished.
 */
template <class A> struct KMul {
  KArgs<A>        k;             /* k.a = a (coefficients, limb 0); nblocks / s0 / logn as for a forward block pass */
  const uint64_t *b;             /* b^ (limb 0) */
  uint64_t *      out;           /* c^ (limb 0) */
  uint64_t        b_limb_stride; /* words between consecutive limbs of b^ */
  uint32_t        lazy_in;       /* words of b^ may be lazy: anywhere in [0,4q) */
  uint32_t        b_bcast;       /* b^ is one polynomial per limb, shared by the whole batch */
  uint32_t        accumulate;    /* c^ += ... (c^ canonical on entry) */
};

/* live = false: a descriptor of zero records -- the loads return 0 and move no data (the accumulator words of a call that
 * does not accumulate: a run-time branch around the loads would make the register allocator keep two sets apart) */
template <int LOGN, int E0, int E1, int AUX = 0>
__device__ __forceinline__ void prefetch_last_range(uint64_t (&raw)[kE], uint32_t t, const uint64_t *blk, bool live = true)
{
  using P           = Plan<LOGN>;
  constexpr int G   = P::NG - 1;
  const uint32_t ib = P::IBASE(G, t);
  const __amdgpu_buffer_rsrc_t r = block_rsrc<LOGN>(blk, live);
  static_for<E0 / 2, E1 / 2>([&](auto hh) {
    constexpr int E = 2 * decltype(hh)::value;
    const u64x2   v = buffer_load_u64x2<AUX>(r, ib * 8u, P::IOFF(G, E) * 8u);
    raw[E]          = v.a;
    raw[E + 1]      = v.b;
  });
}
template <int LOGN, int E0, int E1>
__device__ __forceinline__ void buffer_store_last_range(const uint64_t (&u)[kE], uint32_t t, uint64_t *blk)
{
  using P           = Plan<LOGN>;
  constexpr int G   = P::NG - 1;
  const uint32_t ib = P::IBASE(G, t);
  const __amdgpu_buffer_rsrc_t r = block_rsrc<LOGN>(blk);
  typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
  static_for<E0 / 2, E1 / 2>([&](auto hh) {
    constexpr int E = 2 * decltype(hh)::value;
    v4u32         v;
    v.x = (unsigned)u[E];
    v.y = (unsigned)(u[E] >> 32);
    v.z = (unsigned)u[E + 1];
    v.w = (unsigned)(u[E + 1] >> 32);
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)(ib * 8u), (int)(P::IOFF(G, E) * 8u), 0);
  });
}

template <class A, int LOGN, int KSH, bool MULTI = false>
__global__ void __launch_bounds__((Geom<LOGN, false, flavor_of<A>()>::WG), (Geom<LOGN, false, flavor_of<A>()>::WPS)) fwd_mul_kernel(const KMul<A> km)
{
  uint32_t        bid, gdim, limb;
  const Params<A> p = limb_params<A, false, MULTI>(km.k, bid, gdim, limb);
  using P = Plan<LOGN>;
  using G = Geom<LOGN, false, flavor_of<A>()>;
  constexpr uint32_t MASK   = fused_mask<A, LOGN, false, KSH>();
  constexpr int      LDS_TW = G::LDS_TW;
  __shared__ typename A::val lds_all[G::BPW * P::LDS_ELEMS + LDS_TW];
  const uint32_t   tid   = threadIdx.x;
  const uint32_t   sub   = tid >> P::LT;
  const uint32_t   t     = tid & (P::T - 1);
  typename A::val *lds   = lds_all + sub * P::LDS_ELEMS;
  const uint32_t   bmask = (1u << p.s0) - 1u;
  const bool       lazy  = km.lazy_in != 0;
  const bool       bc    = km.b_bcast != 0;
  const bool       acc   = km.accumulate != 0;
  const uint64_t * bptr  = km.b + (uint64_t)limb * km.b_limb_stride; /* (MULTI off: limb == 0) */
  uint64_t *       cptr  = km.out + (uint64_t)limb * km.k.limb_stride;

  if constexpr(G::PERSISTENT && A::kCompact && !MULTI) {
    constexpr int  GL  = P::NG - 1;
    constexpr bool PRE = stage_is_compact<A, LOGN, false>(GL, 0) && G::TBL(GL) == 0;
    constexpr bool LTW = LDS_TW > 0;
    static_assert(P::NG >= 3, "the persistent blocks have at least three stage groups");
    const uint32_t         tt     = G::BPW == 1 ? tid : t;
    typename A::val *const ll     = G::BPW == 1 ? lds_all : lds;
    const uint64_t         stride = (uint64_t)gdim * G::BPW;
    uint64_t               b0     = (uint64_t)bid * G::BPW;
    if(b0 >= p.nblocks) return;
    const uint64_t lastb = p.nblocks - 1;
    uint64_t       b     = G::BPW == 1 ? b0 : (b0 + sub < p.nblocks ? b0 + sub : lastb);
    typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
    const lds_ctw_ptr<A>   ltw  = (lds_ctw_ptr<A>)tabl;
    if constexpr(LTW) {
      fill_lds_tables<A, LOGN, false>(tabl, p, (uint32_t)b & bmask, tid);
      __syncthreads();
    }
    uint64_t raw[kE];
    prefetch_first<LOGN>(raw, tt, p.a + blk_off<LOGN>(p, b));
    pin_raw(raw);
    for(; b0 < p.nblocks; b0 += stride) {
      const bool live = G::BPW == 1 || b0 + sub < p.nblocks;
      b               = live ? b0 + (G::BPW == 1 ? 0u : sub) : lastb;
      const uint32_t  blk   = (uint32_t)b & bmask;
      const uint64_t *bblk  = bptr + (bc ? ((uint64_t)blk << LOGN) : blk_off<LOGN>(p, b));
      uint64_t *      cblk  = cptr + blk_off<LOGN>(p, b);
      uint32_t        tl    = tt;
      asm volatile("" : "+v"(tl)); /* ties the per-block requests to the iteration (see dot_inv_kernel) */
      typename A::val x[kE];
      convert_inputs<A, false>(x, raw, false, p.c);
      {
        const bool     more = b0 + stride < p.nblocks;
        const uint64_t nb0  = more ? b0 + stride : b0;
        const uint64_t nb   = G::BPW == 1 ? nb0 : (nb0 + sub < p.nblocks ? nb0 + sub : lastb);
        prefetch_first<LOGN>(raw, tl, p.a + blk_off<LOGN>(p, nb), more);
      }
      run_group<A, LOGN, 0, false, MASK>(x, tl, blk, p);
      typename A::ctw pre[4][kE / 2];
      uint64_t        rb[kE], rc[kE];
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        if constexpr(PRE && GI + 1 == GL) {
          /* the last group's twiddles: requested in front of the exchange into it (its LDS round trip hides part of the L2
           * latency; a whole group ahead they would live through the table group next to the prefetched block: spills) */
          sched_fence();
          preload_group_tw<A, LOGN, GL>(pre, tl, blk, p);
          sched_fence();
        }
        exchange<A, LOGN, GI, GI + 1>(x, tl, ll);
        if constexpr(PRE && GI + 1 == GL) {
          run_group_preloaded<A, LOGN, GL, MASK>(x, pre, p);
        } else if constexpr(G::TBL(GI + 1) > 0) {
          run_group<A, LOGN, GI + 1, false, MASK, true>(x, tl, blk, p, ltw + G::TBL_OFF(GI + 1));
        } else {
          run_group<A, LOGN, GI + 1, false, MASK>(x, tl, blk, p);
        }
      });
      /* the products, a quarter of the tile at a time, the next quarter's words in flight meanwhile: at most eight 16-byte
       * words of b^ and c^ per thread live next to the 32 values and the 32 prefetched words of the next block */
      uint64_t u[kE];
      uint32_t t2 = tt; /* (a fresh opaque copy: the lane offsets of this phase are computed here, not carried through the groups) */
      asm volatile("" : "+v"(t2));
      sched_fence();
      prefetch_last_range<LOGN, 0, 4>(rb, t2, bblk);
      static_for<0, 4>([&](auto qq) {
        constexpr int Q = decltype(qq)::value;
        sched_fence();
        /* this quarter's accumulator words (zeros, and no traffic, when the call does not accumulate) and the next
         * quarter's b^ words */
        prefetch_last_range<LOGN, 4 * Q, 4 * Q + 4>(rc, t2, cblk, acc);
        if constexpr(Q < 3) prefetch_last_range<LOGN, 4 * Q + 4, 4 * Q + 8>(rb, t2, bblk);
        sched_fence();
        mul_out_tile<A, 4 * Q, 4 * Q + 4, 1>(u, x, rb, rc, lazy, p.c); /* (one product at a time: two need 14 registers more than there are) */
        if(live) buffer_store_last_range<LOGN, 4 * Q, 4 * Q + 4>(u, t2, cblk);
        sched_fence();
      });
    }
    return;
  } else {
    /* small blocks (several per workgroup), the integer policies, several limbs: the plain loop */
    const lds_ctw_ptr<A> gtw = (lds_ctw_ptr<A>)reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS);
    if constexpr(LDS_TW > 0) {
      fill_lds_tables<A, LOGN, false>(reinterpret_cast<typename A::ctw *>(lds_all + G::BPW * P::LDS_ELEMS), p,
                                      G::BPW == 1 ? ((uint32_t)bid & bmask) : 0u, tid);
      __syncthreads();
    }
    for(uint64_t b0 = (uint64_t)bid * G::BPW; b0 < p.nblocks; b0 += (uint64_t)gdim * G::BPW) {
      uint64_t   b    = b0 + sub;
      const bool live = b < p.nblocks;
      if(!live) b = p.nblocks - 1;
      const uint32_t  blk  = (uint32_t)b & bmask;
      const uint64_t *bblk = bptr + (bc ? ((uint64_t)blk << LOGN) : blk_off<LOGN>(p, b));
      uint64_t *      cblk = cptr + blk_off<LOGN>(p, b);
      /* (an opaque copy of the thread id per block: the integer policy's per-lane twiddle addresses would otherwise be
       * computed once for the launch and sit in registers -- or scratch -- throughout) */
      uint32_t tg = t;
      asm volatile("" : "+v"(tg));
      typename A::val x[kE];
      global_load_first<A, LOGN, false>(x, tg, p.a + blk_off<LOGN>(p, b), false, p.c);
      run_group<A, LOGN, 0, false, MASK, (G::TBL(0) > 0)>(x, tg, blk, p, gtw);
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int GI = decltype(gg)::value;
        exchange<A, LOGN, GI, GI + 1>(x, tg, lds);
        run_group<A, LOGN, GI + 1, false, MASK, (G::TBL(GI + 1) > 0)>(x, tg, blk, p, gtw + G::TBL_OFF(GI + 1));
      });
      static_for<0, 4>([&](auto qq) {
        constexpr int Q = decltype(qq)::value;
        uint64_t      rb[kE], rc[kE], u[kE];
        sched_fence();
        load_last_raw<LOGN, 4 * Q, 4 * Q + 4>(rb, tg, bblk);
        /* (the accumulator words: c^ itself, or zeros when the call does not accumulate) */
        if(acc) load_last_raw<LOGN, 4 * Q, 4 * Q + 4>(rc, tg, cblk);
        else static_for<4 * Q, 4 * Q + 4>([&](auto ee) { rc[decltype(ee)::value] = 0; });
        mul_out_tile<A, 4 * Q, 4 * Q + 4, 1>(u, x, rb, rc, lazy, p.c);
        if(live) store_last_raw<LOGN, 4 * Q, 4 * Q + 4>(u, tg, cblk);
        sched_fence();
      });
    }
  }
}

/* ------------------------------------------------------------------ */
/* c^ = fwd(a) (.) b^ (+ c^) at N = 2^15 .. 2^17 as ONE launch            */
/* ------------------------------------------------------------------ */
/*
 * team_kernel's forward scheme (column items of polynomial j, row items of polynomial j - lag in the same XCD's queue, a
 * per-polynomial counter between them) with fwd_mul_kernel's epilogue in the row items: where the forward block stages would
 * reduce and store their outputs, b^ (and the accumulator) are read by the lane that owns the words and c^ is written --
 * instead of a column launch and a block launch per 256 MiB chunk.  a is scratch (its column stages run in place), c^ may
 * alias a (a block's words are consumed before its products are stored; not when accumulating) or b^.
 */
template <class A> struct KTeamMul {
  KMul<A>  m;       /* m.k.a = a (limb 0), m.k.nblocks = polynomials PER LIMB; b^, c^, flags and strides as for fwd_mul_kernel */
  TeamCtl *ctl;     /* zeroed before the launch */
  uint64_t split_rcp;
  uint32_t lag, nlimbs, poly_major;
};

template <class A, int KSH, int LDAUX>
__device__ __forceinline__ void team_mul_row_item(const uint64_t *ablk, const uint64_t *bblk, uint64_t *cblk, uint32_t blk, uint32_t tid0,
                                                  const Params<A> &p, bool lazy, bool acc, typename A::val *lds, typename A::ctw *tabl)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, flavor_of<A>()>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, false, KSH>();
  /* (an opaque copy of the thread id ties every lane-dependent address to the item: see team_product_item) */
  uint32_t tl = tid0;
  asm volatile("" : "+v"(tl));
  const uint32_t tid = tl;
  uint64_t raw[kE];
  prefetch_first<LOGN, LDAUX>(raw, tid, ablk);
  typename A::val x[kE];
  if constexpr(A::kCompact) {
    constexpr int GL = P::NG - 1;
    static_assert(G::TBL(GL - 1) > 0 && G::TBL(GL) == 0 && !P::WAVE_LOCAL(0, 1), "twiddle placement / barrier this item assumes");
    const lds_ctw_ptr<A> ltw = (lds_ctw_ptr<A>)tabl;
    typename A::ctw      pre[4][kE / 2];
    preload_group_tw<A, LOGN, GL>(pre, tid, blk, p);
    fill_lds_tables<A, LOGN, false>(tabl, p, blk, tid); /* (published by the first exchange's barriers) */
    convert_inputs<A, false>(x, raw, false, p.c);
    run_group<A, LOGN, 0, false, MASK>(x, tid, blk, p);
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int GI = decltype(gg)::value;
      exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
      if constexpr(GI + 1 == GL) {
        run_group_preloaded<A, LOGN, GL, MASK>(x, pre, p);
      } else if constexpr(G::TBL(GI + 1) > 0) {
        run_group<A, LOGN, GI + 1, false, MASK, true>(x, tid, blk, p, ltw + G::TBL_OFF(GI + 1));
      } else {
        run_group<A, LOGN, GI + 1, false, MASK>(x, tid, blk, p);
      }
    });
  } else {
    (void)tabl;
    convert_inputs<A, false>(x, raw, false, p.c);
    run_group<A, LOGN, 0, false, MASK>(x, tid, blk, p);
    static_for<0, P::NG - 1>([&](auto gg) {
      constexpr int GI = decltype(gg)::value;
      exchange<A, LOGN, GI, GI + 1>(x, tid, lds);
      run_group<A, LOGN, GI + 1, false, MASK>(x, tid, blk, p);
    });
  }
  /* the products, a quarter of the tile at a time (fwd_mul_kernel's plain loop) */
  uint32_t t2 = tid0;
  asm volatile("" : "+v"(t2));
  static_for<0, 4>([&](auto qq) {
    constexpr int Q = decltype(qq)::value;
    uint64_t      rb[kE], rc[kE], u[kE];
    sched_fence();
    load_last_raw<LOGN, 4 * Q, 4 * Q + 4>(rb, t2, bblk);
    if(acc) load_last_raw<LOGN, 4 * Q, 4 * Q + 4>(rc, t2, cblk);
    else static_for<4 * Q, 4 * Q + 4>([&](auto ee) { rc[decltype(ee)::value] = 0; });
    mul_out_tile<A, 4 * Q, 4 * Q + 4, 1>(u, x, rb, rc, lazy, p.c);
    store_last_raw<LOGN, 4 * Q, 4 * Q + 4>(u, t2, cblk);
    sched_fence();
  });
}

template <class A, int LEAD, int KSH, bool MULTI = false>
__global__ void __launch_bounds__(256, 4) team_mul_kernel(const KTeamMul<A> kt)
{
  constexpr int LOGN = kTeamBlock;
  using P            = Plan<LOGN>;
  using G            = Geom<LOGN, false, flavor_of<A>()>;
  static_assert((A::kCompact || A::kIntWide) && P::T == kTeamCols && LEAD >= 3 && LEAD <= 5,
                "built for the FP64 policies and the wide integer policy on 2^12-point blocks, N = 2^15..2^17");
  __shared__ typename A::val lds[P::LDS_ELEMS + G::LDS_TW];
  __shared__ unsigned        s_k, s_k2[2];
  typename A::ctw *const tabl = reinterpret_cast<typename A::ctw *>(lds + P::LDS_ELEMS);
  const uint32_t         tid  = threadIdx.x;
  uint32_t               bid_, gdim_, limb_;
  Params<A>              p = limb_params<A, false, false>(kt.m.k, bid_, gdim_, limb_);
  p.s0                     = LEAD;
  constexpr uint32_t CMASK    = column_mask<A, LEAD, false, KSH>();
  constexpr bool     MID_LAZY = !A::kTracksBounds; /* words between the passes: canonical for the FP64 policies */
  const uint32_t logn  = LOGN + LEAD;
  const uint32_t batch = (uint32_t)p.nblocks; /* polynomials per limb */
  const uint32_t total = MULTI ? batch * kt.nlimbs : batch;
  constexpr uint32_t NCOL = 1u << (LOGN - 8), NROW = 1u << LEAD;
  TeamCtl *const ctl = kt.ctl;
  const uint32_t lag = kt.lag;
  const uint32_t my  = xcc_id();
  const bool     bc  = kt.m.b_bcast != 0, lazy = kt.m.lazy_in != 0, acc = kt.m.accumulate != 0;
  uint64_t       boff = 0, coff = 0; /* the item's limb: word offsets of its b^ and c^ slabs (MULTI) */
  for(uint32_t qq = 0; qq < 8; qq++) {
    const uint32_t q = (my + qq) & 7u;
    if(tid == 0) {
      const unsigned prev = atomicCAS(&ctl->owner[q][0], 0u, my + 1u);
      s_k                 = (prev == 0u || prev == my + 1u) ? 1u : 0u;
    }
    __syncthreads();
    const bool mine = s_k != 0;
    __syncthreads();
    if(!mine) continue;
    constexpr uint32_t kNoSignal = 0xffffffffu;
    uint32_t           sig       = kNoSignal;
    for(uint32_t it = 0;; it ^= 1u) {
      /* (lane-0 blocks are followed at once by a workgroup barrier: see team_kernel) */
      if(tid == 0) {
        if(sig != kNoSignal) __hip_atomic_fetch_add(&ctl->done[sig], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_k2[it] = atomicAdd(&ctl->next[q][0], 1u);
      }
      sig = kNoSignal;
      __syncthreads();
      const TeamItem ti = team_decode(uniform_u32(s_k2[it]), q, total, lag, NCOL, NROW, 0u);
      if(ti.stop) break;
      if(!ti.valid) continue;
      const bool     second = ti.pass != 0;
      const uint32_t item = ti.item, pidx = ti.v;
      uint32_t       pl   = pidx;
      if constexpr(MULTI) {
        uint32_t limb;
        team_split(pidx, batch, kt.nlimbs, kt.poly_major, kt.split_rcp, limb, pl);
        team_limb<A, false>(p, kt.m.k, limb);
        boff = (uint64_t)limb * kt.m.b_limb_stride;
        coff = (uint64_t)limb * kt.m.k.limb_stride;
      }
      const uint64_t poff = (uint64_t)pl * p.pstride; /* the polynomial inside its limb: a, c^ and a per-polynomial b^ alike */
      uint64_t *     poly = p.a + poff;
      if(!second) {
        /* inputs -> intermediate (kept dirty in the L2), as in team_kernel */
        team_column_item<A, LEAD, false, CMASK, kAuxSc0Sc1, 0>(poly, item * kTeamCols + tid, logn, p, MID_LAZY);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        sig = pidx;
      } else {
        if(tid == 0) {
          while(__hip_atomic_load(&ctl->done[pidx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NCOL) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
        const uint64_t ioff = (uint64_t)item << LOGN;
        team_mul_row_item<A, KSH, kAuxNt>(poly + ioff, kt.m.b + boff + (bc ? ioff : poff + ioff), kt.m.out + coff + poff + ioff, item, tid, p, lazy,
                                           acc, lds, tabl);
      }
    }
  }
}

template <class A, int R, bool INV, int KSH, bool MULTI = false>
__global__ void __launch_bounds__(256) column_kernel(const KArgs<A> k)
{
  /* k.nblocks = polynomials per limb, k.s0 = first global stage of the pass */
  uint32_t           bid, gdim, limb_;
  const Params<A>    p     = limb_params<A, INV, MULTI>(k, bid, gdim, limb_);
  constexpr uint32_t MASK  = column_mask<A, R, INV, KSH>();
  const uint32_t     lcols = p.logn - R;
  const uint64_t     total = p.nblocks << lcols;
  for(uint64_t g = (uint64_t)bid * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gdim * blockDim.x) {
    const uint64_t poly = g >> lcols;
    const uint32_t col  = (uint32_t)(g & ((1ull << lcols) - 1));
    if constexpr(A::kRadix4) {
      /* (even stage count: launch_pass refuses anything else for this policy) */
      if constexpr(R % 2 == 0) column_pass_thread_r4<A, R, INV>(p.a + poly_offset<false>(poly, p.pstride, p.ptab), col, p.logn, p.s0, p.tw, p.c, p.lazy != 0);
    } else {
      column_pass_thread<A, R, INV, MASK>(p.a + poly_offset<false>(poly, p.pstride, p.ptab), col, p.logn, p.s0, p.wide != 0, p.lastinv != 0, p.tw, p.c, p.lazy != 0);
    }
  }
}

/* ------------------------------------------------------------------ */
/* type-erased launch interface (one translation unit per policy/class) */
/* ------------------------------------------------------------------ */
struct PassArgs {
  uint64_t *  a;
  const void *limbs;       /* HOST array of LimbRec<A>, one per limb (copied into the kernel arguments) */
  int         nlimbs;      /* >= 1 */
  uint64_t    limb_stride; /* words between consecutive limbs' slabs   */
  uint64_t    poly_stride; /* words between consecutive polynomials of a limb (0 = dense: N) */
  const uint64_t *ptab;    /* pointer batch: DEVICE table of per-polynomial word offsets (a = null: entries are addresses / 8), `batch` entries;
                            * null = the progression above */
  uint64_t    batch;       /* polynomials per limb                     */
  uint32_t    logn;   /* whole transform                   */
  int         fused;  /* Pass::fused; 2 = both passes of a 2^16 / 2^17 transform in one workgroup (r = m - 14); 3 = both passes as
                       * items of one launch with the intermediate kept in the XCD's L2 (team_kernel, r = m - 12); 4 = a 2^15-point
                       * transform in one pass, the polynomial in the registers of one workgroup (onepass_kernel) */
  int         r;      /* Pass::r                           */
  int         s;      /* Pass::s                           */
  int         inverse;
  int         wide;
  int         lastinv;
  int         lazy;     /* the caller asked for lazy outputs of the whole transform */
  int         ends;     /* this pass is the last one of the transform */
  int         max_grid; /* cap on workgroups (0 = default) */
  int         num_cus;  /* compute units of the device     */
  int         oversub;  /* persistent block kernels: workgroups per resident slot (0 = block_oversub's default) */
  void *      team_ctl; /* fused == 3: device memory for the queues and counters (TeamCtl + batch counters) */
  int         team_lag, team_wpc;
  hipStream_t stream;
};

template <class A, int KSH> hipError_t launch_pass(const PassArgs &pa);

/* fused product (fused_product_kernel): c = inv(fwd(b) * ahat), whole polynomials of 2^14 points */
struct ProdArgs {
  uint64_t *      b;
  const uint64_t *ahat;
  uint64_t *      out;
  const void *    limbs;       /* HOST array of LimbRec<A> */
  int             nlimbs;
  uint64_t        limb_stride;
  uint64_t        poly_stride; /* words between consecutive polynomials of a limb, the same for all three operands (0 = dense: N) */
  uint64_t        batch;       /* per limb */
  uint32_t        logn;
  uint32_t        block_log; /* N > 2^14: log2 of the blocks (12, 13 or 14); the column passes around the launch cover logn - block_log stages */
  int             a_lazy;
  int             max_grid, num_cus;
  int             oversub;  /* as PassArgs::oversub */
  void *          team_ctl; /* launch_team_product: device memory for the queues and 2 * batch counters */
  int             team_lag, team_wpc;
  int             four; /* launch_team_product: ahat holds a's COEFFICIENTS; the launch transforms both operands */
  int             both; /* launch_product, N <= 2^14: the same for the fused product kernels */
  hipStream_t     stream;
};
template <class A, int KSH> hipError_t launch_product(const ProdArgs &pa);
template <class A, int KSH> hipError_t launch_team_product(const ProdArgs &pa);

/* c = inverse block pass of sum_i a_i^ (.) b_i^ (dot_inv_kernel) */
struct DotArgs {
  uint64_t *             out;
  const uint64_t *const *a; /* HOST arrays of npairs device pointers (limb 0's slabs) */
  const uint64_t *const *b;
  int                    npairs;
  int                    lazy_in, b_bcast;
  const void *           limbs; /* HOST array of LimbRec<A> */
  int                    nlimbs;
  uint64_t               limb_stride, b_limb_stride;
  uint64_t               poly_stride; /* words between consecutive polynomials of a limb: every a_i^, c, and every b_i^ that is not broadcast (0 = dense: N) */
  uint64_t               batch;     /* per limb */
  uint32_t               logn;
  uint32_t               block_log; /* N > 2^14: log2 of the blocks (12 or 14); the inverse column passes follow as launches of their own */
  int                    max_grid, num_cus;
  int                    oversub; /* as PassArgs::oversub */
  void *                 team_ctl; /* N = 2^15..2^17: non-null = both passes as items of ONE launch (team_dot_kernel); TeamCtl + nlimbs * batch counters */
  int                    team_lag, team_wpc;
  hipStream_t            stream;
};
template <class A, int KSH> hipError_t launch_dot(const DotArgs &da);

/* c^ = forward block pass of a, times b^ (+ c^) (fwd_mul_kernel) */
struct MulArgs {
  uint64_t *      a;   /* coefficients (N > 2^14: after the forward column passes) */
  const uint64_t *b;   /* b^ */
  uint64_t *      out; /* c^ */
  int             lazy_in, b_bcast, accumulate;
  const void *    limbs; /* HOST array of LimbRec<A> */
  int             nlimbs;
  uint64_t        limb_stride, b_limb_stride;
  uint64_t        poly_stride; /* words between consecutive polynomials of a limb: a, c^, and b^ unless broadcast (0 = dense: N) */
  uint64_t        batch;
  uint32_t        logn;
  uint32_t        block_log; /* N > 2^14: log2 of the blocks (12 or 14) */
  int             max_grid, num_cus;
  int             oversub; /* as PassArgs::oversub */
  void *          team_ctl; /* N = 2^15..2^17: non-null = column items and row items with the product as ONE launch (team_mul_kernel); a = the caller's coefficients */
  int             team_lag, team_wpc;
  hipStream_t     stream;
};
template <class A, int KSH> hipError_t launch_fwd_mul(const MulArgs &ma);

/* What a pass stores: the last pass of a transform honours the caller's lazy flag; every earlier pass of an
 * integer policy keeps the reference's lazy ranges in HBM (no reduction between stages, as in
 * src/ntt_reference.c:17-30 -- which also makes the final lazy values the reference's bit for bit).  The FP64
 * policy ignores the run-time flag (its passes exchange canonical words). */
inline int pass_lazy(const PassArgs &pa) { return pa.ends ? pa.lazy : 1; }

/* the MULTI kernel variants exist for the scheduled FP64 policy, its 52-bit form and the wide integer policy (RNS bases of
 * 54..60-bit primes: a ciphertext is a few polynomials x tens of such limbs -- one launch instead of one chain per prime) */
template <class A> constexpr bool multi_limb_built() { return A::kCompact || A::kIntWide; }

/* Workgroups launched per resident slot of a persistent block kernel.  One workgroup per slot (the r01..r04 grids) lets the four
 * 256-thread workgroups that share a CU at 2^12 run IN PHASE for the whole launch: they start together, do identical work and
 * meet at the memory system, the LDS pipe and their barriers at the same time.  With several times as many workgroups as
 * slots a slot is refilled whenever its workgroup runs out of blocks, at a time of its own, and the phases of a CU's workgroups
 * decorrelate: measured (profiles/r05/grid_sweep.txt, three alternating repetitions on one box) 2^12 forward 0.592 -> 0.622 of
 * the roofline at 8 workgroups per slot, inverse 0.617 -> 0.641, flat from 8 to 16, 0.61 with one block per workgroup (no
 * prefetch across blocks left); the 1024-thread kernels (2^13, 2^14: one workgroup per CU, 16 waves in step by construction)
 * measured no gain (0.591 at 1, 2, 4 per slot, 0.586 at 8) and keep one.  requested > 0 (NTT_OPT_BLOCK_OVERSUB) overrides. */
template <int LOGN, int WG> constexpr int block_oversub_default(bool whole_polynomials)
{
  return (LOGN == 12 && WG == 256 && whole_polynomials) ? 8 : 1;
}
template <int LOGN, int WG> inline uint64_t block_oversub(int requested, bool whole_polynomials)
{
  return (uint64_t)(requested > 0 ? requested : block_oversub_default<LOGN, WG>(whole_polynomials));
}

template <class A> KArgs<A> make_kargs(const PassArgs &pa)
{
  KArgs<A> k{};
  k.a            = pa.a;
  const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(pa.limbs);
  for(int l = 0; l < (pa.nlimbs > 0 ? pa.nlimbs : 1) && l < kMaxLimbs; l++) k.limbs[l] = recs[l];
  k.limb_stride  = pa.limb_stride;
  k.poly_stride  = pa.poly_stride ? pa.poly_stride : (1ull << pa.logn);
  k.wgs_per_limb = 1;
  k.logn         = pa.logn;
  k.s0           = 0;
  k.wide         = (uint32_t)pa.wide;
  k.lastinv      = (uint32_t)pa.lastinv;
  k.lazy         = (uint32_t)pa.lazy;
  k.nblocks      = pa.batch;
  k.ptab         = pa.ptab;
  return k;
}

template <class A, int LOGN, bool INV, int KSH> hipError_t launch_fused(const PassArgs &pa)
{
  using G = Geom<LOGN, INV, flavor_of<A>()>;
  KArgs<A> p  = make_kargs<A>(pa);
  p.s0        = (uint32_t)pa.s;
  p.lazy      = (uint32_t)pass_lazy(pa);
  p.nblocks   = pa.batch << pa.s;
  const uint64_t nl = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
  uint64_t wgs = (p.nblocks + G::BPW - 1) / G::BPW;
  uint64_t cap = 1ull << 20;
  if(G::PERSISTENT) {
    /* persistent prefetching loop: exactly the resident workgroups (LDS- and
     * wave-limited), each striding over the blocks */
    constexpr int by_lds    = G::WG_PER_CU0;
    constexpr int by_waves  = (G::WPS * 4 * 64) / G::WG;
    constexpr int per_cu    = by_lds < by_waves ? by_lds : by_waves;
    cap                     = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * block_oversub<LOGN, G::WG>(pa.oversub, pa.s == 0);
  }
  if(!G::PERSISTENT && G::LDS_TW > 0) {
    /* tables are filled once per workgroup: a few workgroups per resident slot, each looping */
    if(pa.s != 0) return hipErrorInvalidValue;
    constexpr int per_cu = G::WG_PER_CU0 < 8 ? G::WG_PER_CU0 : 8;
    /* Workgroups that loop over the slab in step produce their loads and stores in bursts; how well the memory system takes
     * them depends on the allocation (the "two modes" of 2^8..2^10: 0.63 or 0.72 of the roofline from one hipMalloc block to
     * the next, profiles/r05/small_size_modes.txt).  At 2^8 and 2^9, where a table fill is cheap, sixteen times as many
     * workgroups (one or two iterations each on a 6 GiB slab) lift the slow mode by 5-7 % (0.633 -> 0.678, 0.636 -> 0.667;
     * inverse +4.5 %) and leave the fast one where it was; 2^10 and 2^11 lose what the larger tables cost, 2^6 and 2^7 are mixed:
     * unchanged (profiles/r05/small_size_grid.txt). */
    const int per_slot = pa.oversub > 0 ? pa.oversub : ((LOGN == 8 || LOGN == 9) ? 64 : 4); /* (NTT_OPT_BLOCK_OVERSUB: sweeps) */
    cap                  = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * (uint64_t)per_slot;
    /* 2^10: about six iterations per workgroup on large batches (32768 workgroups on a 6 GiB slab: slow mode 0.633 -> 0.653; the
     * 8192 of smaller batches stay, where more workgroups lost 2 %) */
    if(LOGN == 10 && pa.oversub <= 0 && wgs / 6 > cap) cap = wgs / 6;
  }
  if(pa.max_grid > 0) cap = (uint64_t)pa.max_grid;
  cap = cap / nl > 0 ? cap / nl : 1; /* the limbs of one launch share the resident workgroups */
  /* a persistent workgroup must always see the same block position inside the
   * polynomial (its LDS twiddle table depends on it): the grid, which is its
   * stride, is a multiple of the 2^s blocks per polynomial (nblocks always is) */
  if(G::BPW == 1 && pa.s > 0) {
    if(cap < (1ull << pa.s)) cap = 1ull << pa.s;
    cap &= ~((1ull << pa.s) - 1);
  }
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  p.wgs_per_limb = (uint32_t)wgs;
  const dim3 grid((unsigned)wgs, (unsigned)nl), wg(G::WG); /* (MULTI variants: blockIdx.y is the limb) */
  if(nl > 1) {
    /* several limbs in one launch: the MULTI variants, built for the FP64 policies (the ones RNS bases use) */
    if constexpr(multi_limb_built<A>()) {
      if constexpr(INV) {
        if(pa.lastinv) {
          hipLaunchKernelGGL((fused_kernel<A, LOGN, true, KSH, true, false, true>), grid, wg, 0, pa.stream, p);
        } else if constexpr(LOGN == kFusedLarge || LOGN == kFusedSmallBlock) {
          hipLaunchKernelGGL((fused_kernel<A, LOGN, true, KSH, false, false, true>), grid, wg, 0, pa.stream, p);
        } else {
          return hipErrorInvalidValue;
        }
      } else {
        if constexpr(A::kTracksBounds) { /* (lazy outputs: a kernel variant for the FP64 policies, a run-time flag for the integer ones) */
          if(pa.ends && pa.lazy) {
            hipLaunchKernelGGL((fused_kernel<A, LOGN, false, KSH, false, true, true>), grid, wg, 0, pa.stream, p);
            return hipGetLastError();
          }
        }
        hipLaunchKernelGGL((fused_kernel<A, LOGN, false, KSH, false, false, true>), grid, wg, 0, pa.stream, p);
      }
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
  if constexpr(INV) {
    /* the inverse kernel exists in two variants: ending a whole transform (N^-1 folded into
     * its last group) -- every block size -- and, for the block size used below column
     * passes, not ending it */
    if(pa.lastinv || A::kRadix4) {
      /* (radix-4 formulation: N^-1 is a pass of its own, fused into the LAST pass's store -- the blocks of a larger
       * transform run the same kernel with the multiplier record of 1: ntt_host.hip, limbrec_mid) */
      hipLaunchKernelGGL((fused_kernel<A, LOGN, true, KSH, true>), grid, wg, 0, pa.stream, p);
    } else if constexpr(LOGN == kFusedLarge || LOGN == kFusedSmallBlock) {
      hipLaunchKernelGGL((fused_kernel<A, LOGN, true, KSH, false>), grid, wg, 0, pa.stream, p);
    } else {
      return hipErrorInvalidValue;
    }
  } else {
    if constexpr(A::kTracksBounds) {
      if(pa.ends && pa.lazy) {
        hipLaunchKernelGGL((fused_kernel<A, LOGN, false, KSH, false, true>), grid, wg, 0, pa.stream, p);
        return hipGetLastError();
      }
    }
    hipLaunchKernelGGL((fused_kernel<A, LOGN, false, KSH, false, false>), grid, wg, 0, pa.stream, p);
  }
  return hipGetLastError();
}

/* pa.r = LEAD (1..3): the whole transform of 2^(14+LEAD) points in one launch; pa.batch polynomials */
/* Built for the FP64 policy at N = 2^16 and 2^17 (BASELINE configs 3 and 5).  The integer policy's larger
 * temporaries and the N = 2^15 inverse do not fit the 128-register budget of a 1024-thread workgroup without
 * scratch: those cases stay on the one-launch-per-pass path (ntt_host.hip: two_phase_applies). */
template <class A, int LEAD> constexpr bool two_phase_built() { return A::kTracksBounds && LEAD >= 2; }

template <class A, int LEAD, bool INV, int KSH> hipError_t launch_twophase(const PassArgs &pa)
{
  if constexpr(!two_phase_built<A, LEAD>()) {
    return hipErrorNotSupported;
  } else {
  if(pa.nlimbs > 1) return hipErrorNotSupported; /* (RNS sets take the per-pass launches) */
  KArgs<A> p = make_kargs<A>(pa);
  p.s0       = (uint32_t)LEAD;
  p.lastinv  = (uint32_t)pa.inverse;
  p.nblocks  = pa.batch;
  uint64_t wgs = pa.batch;
  uint64_t cap = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256);
  if(pa.max_grid > 0) cap = (uint64_t)pa.max_grid;
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  p.wgs_per_limb = (uint32_t)wgs;
  hipLaunchKernelGGL((twophase_kernel<A, LEAD, INV, KSH>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, p);
  return hipGetLastError();
  }
}

/* N = 2^15 in one pass (onepass_kernel): one persistent 1024-thread workgroup per CU, pa.batch polynomials per limb */
template <class A> constexpr bool onepass_built() { return A::kCompact && A::kTracksBounds; }
template <class A, bool INV, int KSH> hipError_t launch_onepass(const PassArgs &pa)
{
  if constexpr(!onepass_built<A>()) {
    return hipErrorNotSupported;
  } else {
    if(pa.logn != (uint32_t)kFusedLarge + 1 || pa.lazy) return hipErrorNotSupported;
    const uint64_t nl = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
    if(nl > (uint64_t)kMaxLimbs) return hipErrorNotSupported;
    KArgs<A> p = make_kargs<A>(pa);
    p.s0       = 1;
    p.lastinv  = (uint32_t)pa.inverse;
    p.lazy     = 0;
    p.nblocks  = pa.batch;
    uint64_t wgs = pa.batch;
    uint64_t cap = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256);
    if(pa.max_grid > 0) cap = (uint64_t)pa.max_grid;
    cap = cap / nl > 0 ? cap / nl : 1;
    if(wgs > cap) wgs = cap;
    if(wgs == 0) return hipSuccess;
    p.wgs_per_limb = (uint32_t)wgs;
    const dim3 grid((unsigned)wgs, (unsigned)nl);
    if(nl > 1) hipLaunchKernelGGL((onepass_kernel<A, INV, KSH, true>), grid, dim3(1024), 0, pa.stream, p);
    else hipLaunchKernelGGL((onepass_kernel<A, INV, KSH, false>), grid, dim3(1024), 0, pa.stream, p);
    return hipGetLastError();
  }
}

/* pa.r = LEAD (3..5), pa.batch polynomials of 2^(12 + LEAD) points per limb; pa.team_ctl: TeamCtl with nlimbs * batch counters,
 * zeroed here.  Several limbs (an RNS set, [limb][batch][N]): the MULTI variant, the queues run over all limbs' polynomials. */
/* Zeroes a control block (queue heads, owners, per-polynomial counters) in front of an XCD-local launch -- as a KERNEL, not as an
 * asynchronous memset: captured into a HIP graph, a memset node in front of the kernel node did not always take effect before the
 * kernel's first workgroups read the counters (stale counters of the previous replay: second-pass items that do not wait, or
 * queues that look exhausted -- found by replaying a captured NTT-domain product between other work, round 5:
 * tests/test_gpu_parity.py::test_one_launch_ntt_domain_products_captured_in_a_hip_graph).  A kernel in front of a kernel on the
 * same stream is ordered in a graph exactly as outside one. */
static __global__ void __launch_bounds__(256) team_ctl_clear_kernel(unsigned *w, size_t n)
{
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) w[i] = 0u;
}
static inline hipError_t team_ctl_clear(void *ctl, size_t bytes, hipStream_t stream)
{
  const size_t n = (bytes + 3) / 4;
  size_t       g = (n + 255) / 256;
  if(g > 64) g = 64;
  hipLaunchKernelGGL(team_ctl_clear_kernel, dim3((unsigned)g), dim3(256), 0, stream, static_cast<unsigned *>(ctl), n);
  return hipGetLastError();
}

template <class A, int LEAD, bool INV, int KSH> hipError_t launch_team(const PassArgs &pa)
{
  if constexpr(!(A::kCompact || A::kIntWide)) {
    return hipErrorNotSupported;
  } else {
    const uint64_t nl = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
    if(nl > (uint64_t)kMaxLimbs || !pa.team_ctl || pa.wide || pa.lazy || nl * pa.batch >= (1ull << 31)) return hipErrorNotSupported;
    KTeam<A> kt{};
    kt.k         = make_kargs<A>(pa);
    kt.k.lazy    = 0; /* canonical out (the integer policies read this flag at run time) */
    kt.k.lastinv = (uint32_t)pa.inverse;
    kt.k.nblocks = pa.batch;
    kt.ctl       = static_cast<TeamCtl *>(pa.team_ctl);
    kt.lag       = (uint32_t)(pa.team_lag > 0 ? pa.team_lag : 6);
    kt.nlimbs    = (uint32_t)nl;
    kt.poly_major = nl > 1 && kt.k.poly_stride > kt.k.limb_stride;
    kt.split_rcp  = team_split_rcp(kt.poly_major ? nl : pa.batch);
    const size_t bytes = sizeof(TeamCtl) + (size_t)(nl * pa.batch) * sizeof(unsigned);
    hipError_t   e     = team_ctl_clear(pa.team_ctl, bytes, pa.stream);
    if(e != hipSuccess) return e;
    /* four workgroups per CU: 40,580 bytes of LDS each (32.9 KB exchange buffer + 7.5 KB table), at most 128 VGPRs */
    uint64_t wgs = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * (pa.team_wpc > 0 ? pa.team_wpc : 4);
    if(pa.max_grid > 0) wgs = (uint64_t)pa.max_grid;
    kt.k.wgs_per_limb = (uint32_t)wgs;
    if(nl > 1) hipLaunchKernelGGL((team_kernel<A, LEAD, INV, KSH, true>), dim3((unsigned)wgs), dim3(256), 0, pa.stream, kt);
    else hipLaunchKernelGGL((team_kernel<A, LEAD, INV, KSH, false>), dim3((unsigned)wgs), dim3(256), 0, pa.stream, kt);
    return hipGetLastError();
  }
}

template <class A, int R, bool INV, int KSH> hipError_t launch_column(const PassArgs &pa)
{
  KArgs<A> p = make_kargs<A>(pa);
  p.s0       = (uint32_t)pa.s;
  p.lazy     = (uint32_t)pass_lazy(pa);
  p.nblocks  = pa.batch;
  const uint64_t nl    = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
  const uint64_t total = pa.batch << (pa.logn - R);
  uint64_t       wgs   = (total + 255) / 256;
  uint64_t       cap   = pa.max_grid > 0 ? (uint64_t)pa.max_grid : (1ull << 22);
  cap                  = cap / nl > 0 ? cap / nl : 1;
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  p.wgs_per_limb = (uint32_t)wgs;
  if(nl > 1) {
    if constexpr(multi_limb_built<A>()) {
      hipLaunchKernelGGL((column_kernel<A, R, INV, KSH, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(256), 0, pa.stream, p);
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
  hipLaunchKernelGGL((column_kernel<A, R, INV, KSH>), dim3((unsigned)wgs), dim3(256), 0, pa.stream, p);
  return hipGetLastError();
}

template <class A, int KSH> hipError_t launch_product_impl(const ProdArgs &pa)
{
  if constexpr(!A::kCompact) {
    return hipErrorNotSupported;
  } else {
    if(pa.logn < 8 || pa.logn > 17) return hipErrorNotSupported;
    const uint32_t blog = pa.logn <= 14 ? pa.logn : (pa.block_log ? pa.block_log : 14u);
    if(blog < 12 && pa.logn > 14) return hipErrorInvalidValue;
    const uint32_t s0 = pa.logn - blog; /* leading stages done by column passes around this launch */
    KProd<A> pp{};
    pp.f.a            = pa.b;
    const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(pa.limbs);
    for(int l = 0; l < (pa.nlimbs > 0 ? pa.nlimbs : 1) && l < kMaxLimbs; l++) pp.f.limbs[l] = recs[l];
    pp.f.limb_stride  = pa.limb_stride;
    pp.f.poly_stride  = pa.poly_stride ? pa.poly_stride : (1ull << pa.logn);
    pp.f.wgs_per_limb = 1;
    pp.f.logn         = pa.logn;
    pp.f.s0           = s0;
    pp.f.nblocks      = pa.batch << s0;
    pp.ahat           = pa.ahat;
    pp.out            = pa.out;
    pp.a_lazy         = (uint32_t)pa.a_lazy;
    const uint64_t nl = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
    uint64_t wgs = pp.f.nblocks;
    uint64_t cap = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256);
    if(pa.max_grid > 0) cap = (uint64_t)pa.max_grid;
    cap = cap / nl > 0 ? cap / nl : 1;
    /* a workgroup keeps the tables of ONE block position: its stride is a multiple of the blocks per polynomial */
    if(cap < (1ull << s0)) cap = 1ull << s0;
    cap &= ~((1ull << s0) - 1);
    if(wgs > cap) wgs = cap;
    if(wgs == 0) return hipSuccess;
    /* a^ always arrives as the lazy words ntt_fwd_batch_lazy leaves (the canonical-operand variant is not built) -- or not
     * at all: pa.both, whole polynomials, a's coefficients in pa.ahat */
    if(!pa.a_lazy && !pa.both) return hipErrorNotSupported;
    if(pa.both && s0 != 0 && blog != 12 && blog != 14) return hipErrorInvalidValue;
    if(blog < 12) {
      switch(pa.logn) {
#define NTT_SMALL_PRODUCT(LN)                                                                                       \
  case LN: {                                                                                                        \
    using GS = Geom<LN, false, 3>;                                                                                  \
    constexpr int per_cu = GS::WG_PER_CU0 < 8 ? GS::WG_PER_CU0 : 8;                                                 \
    uint64_t      g      = (pp.f.nblocks + GS::BPW - 1) / GS::BPW;                                                  \
    uint64_t      gcap   = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * 4;           \
    if(pa.max_grid > 0) gcap = (uint64_t)pa.max_grid;                                                               \
    gcap = gcap / nl > 0 ? gcap / nl : 1;                                                                           \
    if(g > gcap) g = gcap;                                                                                          \
    pp.f.wgs_per_limb = (unsigned)g;                                                                                \
    if(pa.both) {                                                                                                   \
      if(nl > 1) hipLaunchKernelGGL((fused_product_small_kernel<A, LN, KSH, true, true>), dim3((unsigned)g, (unsigned)nl), dim3(GS::WG), 0, pa.stream, pp); \
      else hipLaunchKernelGGL((fused_product_small_kernel<A, LN, KSH, false, true>), dim3((unsigned)g), dim3(GS::WG), 0, pa.stream, pp); \
      return hipGetLastError();                                                                                     \
    }                                                                                                               \
    if(nl > 1) hipLaunchKernelGGL((fused_product_small_kernel<A, LN, KSH, true>), dim3((unsigned)g, (unsigned)nl), dim3(GS::WG), 0, pa.stream, pp); \
    else hipLaunchKernelGGL((fused_product_small_kernel<A, LN, KSH>), dim3((unsigned)g), dim3(GS::WG), 0, pa.stream, pp); \
    return hipGetLastError();                                                                                       \
  }
        NTT_SMALL_PRODUCT(8)
        NTT_SMALL_PRODUCT(9)
        NTT_SMALL_PRODUCT(10)
        NTT_SMALL_PRODUCT(11)
#undef NTT_SMALL_PRODUCT
        default: return hipErrorNotSupported;
      }
    }
    if(blog == 12) {
      using G12 = Geom<12, false, 3>;
      constexpr int per_cu = G12::WG_PER_CU0 < G12::WPS ? G12::WG_PER_CU0 : G12::WPS; /* 256-thread workgroups: one wave per SIMD each */
      /* (whole polynomials: several workgroups per resident slot, as for the transforms -- block_oversub; measured 0.369 -> 0.401 of
       * the 24N roofline at 8 per slot, profiles/r05/oversub_sweep.txt) */
      uint64_t      cap12  = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * per_cu * block_oversub<12, G12::WG>(pa.oversub, s0 == 0);
      if(pa.max_grid > 0) cap12 = (uint64_t)pa.max_grid;
      cap12 = cap12 / nl > 0 ? cap12 / nl : 1;
      if(cap12 < (1ull << s0)) cap12 = 1ull << s0;
      cap12 &= ~((1ull << s0) - 1);
      wgs = pp.f.nblocks < cap12 ? pp.f.nblocks : cap12;
      pp.f.wgs_per_limb = (uint32_t)wgs;
      if(pa.both) {
        if(s0 == 0) {
          if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G12::WG), 0, pa.stream, pp);
          else hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, true, false, true>), dim3((unsigned)wgs), dim3(G12::WG), 0, pa.stream, pp);
        } else {
          if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, false, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G12::WG), 0, pa.stream, pp);
          else hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, false, false, true>), dim3((unsigned)wgs), dim3(G12::WG), 0, pa.stream, pp);
        }
        return hipGetLastError();
      }
      if(nl > 1) {
        if(s0 == 0) hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G12::WG), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, false, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G12::WG), 0, pa.stream, pp);
      } else {
        if(s0 == 0) hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, true>), dim3((unsigned)wgs), dim3(G12::WG), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 12, KSH, true, false>), dim3((unsigned)wgs), dim3(G12::WG), 0, pa.stream, pp);
      }
      return hipGetLastError();
    }
    if(blog == 13) {
      using G13 = Geom<13, false, 3>;
      /* one 512-thread workgroup per CU by LDS (64 KB exchange buffer + 30 KB table); a second one does not fit */
      if(s0 != 0) return hipErrorNotSupported; /* (2^13-point blocks of a larger product: measured no faster than 2^14, not built) */
      pp.f.wgs_per_limb = (uint32_t)wgs;
      if(pa.both) {
        if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 13, KSH, true, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G13::WG), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 13, KSH, true, true, false, true>), dim3((unsigned)wgs), dim3(G13::WG), 0, pa.stream, pp);
        return hipGetLastError();
      }
      if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 13, KSH, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G13::WG), 0, pa.stream, pp);
      else hipLaunchKernelGGL((fused_product_kernel<A, 13, KSH, true, true>), dim3((unsigned)wgs), dim3(G13::WG), 0, pa.stream, pp);
      return hipGetLastError();
    }
    pp.f.wgs_per_limb = (uint32_t)wgs;
    if(pa.both) {
      if(s0 == 0) {
        if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(1024), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, true, false, true>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, pp);
      } else {
        if(nl > 1) hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, false, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(1024), 0, pa.stream, pp);
        else hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, false, false, true>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, pp);
      }
      return hipGetLastError();
    }
    if(nl > 1) {
      if(s0 == 0) hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, true, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(1024), 0, pa.stream, pp);
      else hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, false, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(1024), 0, pa.stream, pp);
    } else {
      if(s0 == 0) hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, true>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, pp);
      else hipLaunchKernelGGL((fused_product_kernel<A, 14, KSH, true, false>), dim3((unsigned)wgs), dim3(1024), 0, pa.stream, pp);
    }
    return hipGetLastError();
  }
}

/* a product at N = 2^15..2^17 as one launch (team_product_kernel); pa.team_ctl: TeamProdCtl + 2 * nlimbs * batch counters.
 * Several limbs ([limb][batch][N] slabs, limb_stride apart): the MULTI variants -- ONE launch for a whole RNS product. */
template <class A, int KSH> hipError_t launch_team_product_impl(const ProdArgs &pa)
{
  if constexpr(!A::kCompact) {
    return hipErrorNotSupported;
  } else {
    const uint64_t nl = (uint64_t)(pa.nlimbs > 0 ? pa.nlimbs : 1);
    if(nl > (uint64_t)kMaxLimbs || !pa.team_ctl || (!pa.a_lazy && !pa.four) || pa.logn < kTeamBlock + 3 || pa.logn > kTeamBlock + 5 ||
       nl * pa.batch >= (1ull << 30)) {
      return hipErrorNotSupported;
    }
    KTeamProd<A> kt{};
    kt.k.f.a            = pa.b;
    const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(pa.limbs);
    for(uint64_t l = 0; l < nl; l++) kt.k.f.limbs[l] = recs[l];
    kt.k.f.limb_stride  = nl > 1 ? pa.limb_stride : 0;
    kt.k.f.poly_stride  = pa.poly_stride ? pa.poly_stride : (1ull << pa.logn);
    kt.k.f.logn         = pa.logn;
    kt.k.f.s0           = pa.logn - kTeamBlock;
    kt.k.f.nblocks      = pa.batch;
    kt.k.ahat           = pa.ahat;
    kt.k.out            = pa.out;
    kt.k.a_lazy         = 1;
    kt.ctl              = static_cast<TeamProdCtl *>(pa.team_ctl);
    kt.lag              = (uint32_t)(pa.team_lag > 0 ? pa.team_lag : 8);
    kt.nlimbs           = (uint32_t)nl;
    kt.poly_major       = nl > 1 && kt.k.f.poly_stride > kt.k.f.limb_stride;
    kt.split_rcp        = team_split_rcp(kt.poly_major ? nl : pa.batch);
    const size_t bytes = sizeof(TeamProdCtl) + 2 * (size_t)(nl * pa.batch) * sizeof(unsigned);
    hipError_t   e     = team_ctl_clear(pa.team_ctl, bytes, pa.stream);
    if(e != hipSuccess) return e;
    /* four workgroups per CU (121 VGPRs, 40.6 KB of LDS each) */
    uint64_t wgs = (uint64_t)(pa.num_cus > 0 ? pa.num_cus : 256) * (pa.team_wpc > 0 ? pa.team_wpc : 4);
    if(pa.max_grid > 0) wgs = (uint64_t)pa.max_grid;
    kt.k.f.wgs_per_limb = (uint32_t)wgs;
    const dim3 g((unsigned)wgs), t(256);
#define NTT_TEAM_PROD(LEADV, FOURV)                                                                                \
  do {                                                                                                             \
    if(nl > 1) hipLaunchKernelGGL((team_product_kernel<A, LEADV, KSH, FOURV, true>), g, t, 0, pa.stream, kt);      \
    else hipLaunchKernelGGL((team_product_kernel<A, LEADV, KSH, FOURV, false>), g, t, 0, pa.stream, kt);           \
  } while(0)
    if(pa.four) {
      /* ahat = a itself (coefficients): both forward transforms happen inside the launch */
      switch(pa.logn - kTeamBlock) {
        case 3: NTT_TEAM_PROD(3, true); break;
        case 4: NTT_TEAM_PROD(4, true); break;
        default: NTT_TEAM_PROD(5, true); break;
      }
      return hipGetLastError();
    }
    switch(pa.logn - kTeamBlock) {
      case 3: NTT_TEAM_PROD(3, false); break;
      case 4: NTT_TEAM_PROD(4, false); break;
      default: NTT_TEAM_PROD(5, false); break;
    }
#undef NTT_TEAM_PROD
    return hipGetLastError();
  }
}

template <class A, int LOGN, int KSH, bool LASTINV> hipError_t launch_dot_blocks(const DotArgs &da)
{
  using G = Geom<LOGN, true, flavor_of<A>()>;
  KDot<A> kd{};
  kd.k.a                 = da.out;
  const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(da.limbs);
  const uint64_t    nl   = (uint64_t)(da.nlimbs > 0 ? da.nlimbs : 1);
  for(uint64_t l = 0; l < nl && l < (uint64_t)kMaxLimbs; l++) kd.k.limbs[l] = recs[l];
  kd.k.limb_stride = da.limb_stride;
  kd.k.poly_stride = da.poly_stride ? da.poly_stride : (1ull << da.logn);
  kd.k.logn        = da.logn;
  kd.k.s0          = da.logn - (uint32_t)LOGN;
  kd.k.lastinv     = LASTINV ? 1u : 0u;
  kd.k.lazy        = LASTINV ? 0u : 1u;
  kd.k.nblocks     = da.batch << kd.k.s0;
  kd.npairs        = (uint32_t)da.npairs;
  kd.lazy_in       = (uint32_t)da.lazy_in;
  kd.b_bcast       = (uint32_t)da.b_bcast;
  kd.b_limb_stride = da.b_limb_stride;
  for(int i = 0; i < da.npairs && i < kMaxDot; i++) {
    kd.a[i] = da.a[i];
    kd.b[i] = da.b[i];
  }
  /* the grid of the inverse block kernel (launch_fused): resident workgroups striding over the blocks */
  uint64_t wgs = (kd.k.nblocks + G::BPW - 1) / G::BPW;
  uint64_t cap = 1ull << 20;
  if(G::PERSISTENT) {
    constexpr int by_lds   = G::WG_PER_CU0;
    constexpr int by_waves = (G::WPS * 4 * 64) / G::WG;
    constexpr int per_cu   = by_lds < by_waves ? by_lds : by_waves;
    cap                    = (uint64_t)(da.num_cus > 0 ? da.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * block_oversub<LOGN, G::WG>(da.oversub, kd.k.s0 == 0);
  }
  if(!G::PERSISTENT && G::LDS_TW > 0) {
    if(kd.k.s0 != 0) return hipErrorInvalidValue;
    constexpr int per_cu = G::WG_PER_CU0 < 8 ? G::WG_PER_CU0 : 8;
    cap                  = (uint64_t)(da.num_cus > 0 ? da.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * 4;
  }
  if(da.max_grid > 0) cap = (uint64_t)da.max_grid;
  cap = cap / nl > 0 ? cap / nl : 1;
  if(G::BPW == 1 && kd.k.s0 > 0) { /* a workgroup keeps the tables of ONE block position */
    if(cap < (1ull << kd.k.s0)) cap = 1ull << kd.k.s0;
    cap &= ~((1ull << kd.k.s0) - 1);
  }
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  kd.k.wgs_per_limb = (uint32_t)wgs;
  if(nl > 1) {
    if constexpr(multi_limb_built<A>()) {
      hipLaunchKernelGGL((dot_inv_kernel<A, LOGN, KSH, LASTINV, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G::WG), 0, da.stream, kd);
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
  hipLaunchKernelGGL((dot_inv_kernel<A, LOGN, KSH, LASTINV, false>), dim3((unsigned)wgs), dim3(G::WG), 0, da.stream, kd);
  return hipGetLastError();
}

/* the NTT-domain product at N = 2^15..2^17 as ONE launch (team_dot_kernel); da.team_ctl: TeamCtl + nlimbs * batch counters, zeroed here */
template <class A, int KSH> hipError_t launch_team_dot(const DotArgs &da)
{
  if constexpr(!(A::kCompact || A::kIntWide)) {
    return hipErrorNotSupported;
  } else {
    const uint64_t nl = (uint64_t)(da.nlimbs > 0 ? da.nlimbs : 1);
    if(nl > (uint64_t)kMaxLimbs || !da.team_ctl || da.logn < (uint32_t)kTeamBlock + 3 || da.logn > (uint32_t)kTeamBlock + 5 ||
       nl * da.batch >= (1ull << 31) || da.npairs < 1 || da.npairs > kMaxDot) {
      return hipErrorNotSupported;
    }
    KTeamDot<A> kt{};
    kt.d.k.a                = da.out;
    const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(da.limbs);
    for(uint64_t l = 0; l < nl; l++) kt.d.k.limbs[l] = recs[l];
    kt.d.k.limb_stride = nl > 1 ? da.limb_stride : 0;
    kt.d.k.poly_stride = da.poly_stride ? da.poly_stride : (1ull << da.logn);
    kt.d.k.logn        = da.logn;
    kt.d.k.s0          = da.logn - (uint32_t)kTeamBlock;
    kt.d.k.lastinv     = 1;
    kt.d.k.lazy        = 0;
    kt.d.k.nblocks     = da.batch;
    kt.d.npairs        = (uint32_t)da.npairs;
    kt.d.lazy_in       = (uint32_t)da.lazy_in;
    kt.d.b_bcast       = (uint32_t)da.b_bcast;
    kt.d.b_limb_stride = nl > 1 ? da.b_limb_stride : 0;
    for(int i = 0; i < da.npairs; i++) {
      kt.d.a[i] = da.a[i];
      kt.d.b[i] = da.b[i];
    }
    kt.ctl        = static_cast<TeamCtl *>(da.team_ctl);
    kt.lag        = (uint32_t)(da.team_lag > 0 ? da.team_lag : 8);
    kt.nlimbs     = (uint32_t)nl;
    kt.poly_major = nl > 1 && kt.d.k.poly_stride > kt.d.k.limb_stride;
    kt.split_rcp  = team_split_rcp(kt.poly_major ? nl : da.batch);
    const size_t bytes = sizeof(TeamCtl) + (size_t)(nl * da.batch) * sizeof(unsigned);
    hipError_t   e     = team_ctl_clear(da.team_ctl, bytes, da.stream);
    if(e != hipSuccess) return e;
    uint64_t wgs = (uint64_t)(da.num_cus > 0 ? da.num_cus : 256) * (da.team_wpc > 0 ? da.team_wpc : 4);
    if(da.max_grid > 0) wgs = (uint64_t)da.max_grid;
    kt.d.k.wgs_per_limb = (uint32_t)wgs;
    const dim3 g((unsigned)wgs), t(256);
#define NTT_TEAM_DOT(LEADV)                                                                                  \
  do {                                                                                                       \
    if(nl > 1) hipLaunchKernelGGL((team_dot_kernel<A, LEADV, KSH, true>), g, t, 0, da.stream, kt);           \
    else hipLaunchKernelGGL((team_dot_kernel<A, LEADV, KSH, false>), g, t, 0, da.stream, kt);                \
  } while(0)
    switch(da.logn - kTeamBlock) {
      case 3: NTT_TEAM_DOT(3); break;
      case 4: NTT_TEAM_DOT(4); break;
      default: NTT_TEAM_DOT(5); break;
    }
#undef NTT_TEAM_DOT
    return hipGetLastError();
  }
}

template <class A, int KSH> hipError_t launch_dot_impl(const DotArgs &da)
{
  if(da.npairs < 1 || da.npairs > kMaxDot || da.nlimbs > kMaxLimbs) return hipErrorInvalidValue;
  if(da.team_ctl) return launch_team_dot<A, KSH>(da);
  if(da.logn > (uint32_t)kFusedMax) {
    if(da.block_log == (uint32_t)kFusedSmallBlock) return launch_dot_blocks<A, kFusedSmallBlock, KSH, false>(da);
    if(da.block_log == (uint32_t)kFusedLarge) return launch_dot_blocks<A, kFusedLarge, KSH, false>(da);
    return hipErrorInvalidValue;
  }
  switch(da.logn) {
#define NTT_DOT_CASE(LN) \
  case LN: return launch_dot_blocks<A, LN, KSH, true>(da);
    NTT_DOT_CASE(6) NTT_DOT_CASE(7) NTT_DOT_CASE(8) NTT_DOT_CASE(9) NTT_DOT_CASE(10) NTT_DOT_CASE(11) NTT_DOT_CASE(12) NTT_DOT_CASE(13)
    NTT_DOT_CASE(14)
#undef NTT_DOT_CASE
    default: return hipErrorNotSupported;
  }
}

template <class A, int LOGN, int KSH> hipError_t launch_fwd_mul_blocks(const MulArgs &ma)
{
  using G = Geom<LOGN, false, flavor_of<A>()>;
  KMul<A> km{};
  km.k.a                 = ma.a;
  const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(ma.limbs);
  const uint64_t    nl   = (uint64_t)(ma.nlimbs > 0 ? ma.nlimbs : 1);
  for(uint64_t l = 0; l < nl && l < (uint64_t)kMaxLimbs; l++) km.k.limbs[l] = recs[l];
  km.k.limb_stride = ma.limb_stride;
  km.k.poly_stride = ma.poly_stride ? ma.poly_stride : (1ull << ma.logn);
  km.k.logn        = ma.logn;
  km.k.s0          = ma.logn - (uint32_t)LOGN;
  km.k.nblocks     = ma.batch << km.k.s0;
  km.b             = ma.b;
  km.out           = ma.out;
  km.b_limb_stride = ma.b_limb_stride;
  km.lazy_in       = (uint32_t)ma.lazy_in;
  km.b_bcast       = (uint32_t)ma.b_bcast;
  km.accumulate    = (uint32_t)ma.accumulate;
  /* the grid of the forward block kernel (launch_fused) */
  uint64_t wgs = (km.k.nblocks + G::BPW - 1) / G::BPW;
  uint64_t cap = 1ull << 20;
  if(G::PERSISTENT) {
    constexpr int by_lds   = G::WG_PER_CU0;
    constexpr int by_waves = (G::WPS * 4 * 64) / G::WG;
    constexpr int per_cu   = by_lds < by_waves ? by_lds : by_waves;
    cap                    = (uint64_t)(ma.num_cus > 0 ? ma.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * block_oversub<LOGN, G::WG>(ma.oversub, km.k.s0 == 0);
  }
  if(!G::PERSISTENT && G::LDS_TW > 0) {
    if(km.k.s0 != 0) return hipErrorInvalidValue;
    constexpr int per_cu = G::WG_PER_CU0 < 8 ? G::WG_PER_CU0 : 8;
    cap                  = (uint64_t)(ma.num_cus > 0 ? ma.num_cus : 256) * (per_cu > 0 ? per_cu : 1) * 4;
  }
  if(ma.max_grid > 0) cap = (uint64_t)ma.max_grid;
  cap = cap / nl > 0 ? cap / nl : 1;
  if(G::BPW == 1 && km.k.s0 > 0) { /* a workgroup keeps the tables of ONE block position */
    if(cap < (1ull << km.k.s0)) cap = 1ull << km.k.s0;
    cap &= ~((1ull << km.k.s0) - 1);
  }
  if(G::BPW > 1 && km.k.s0 > 0) return hipErrorInvalidValue; /* (two blocks per workgroup: whole polynomials only) */
  if(wgs > cap) wgs = cap;
  if(wgs == 0) return hipSuccess;
  km.k.wgs_per_limb = (uint32_t)wgs;
  if(nl > 1) {
    if constexpr(multi_limb_built<A>()) {
      hipLaunchKernelGGL((fwd_mul_kernel<A, LOGN, KSH, true>), dim3((unsigned)wgs, (unsigned)nl), dim3(G::WG), 0, ma.stream, km);
      return hipGetLastError();
    } else {
      return hipErrorNotSupported;
    }
  }
  hipLaunchKernelGGL((fwd_mul_kernel<A, LOGN, KSH, false>), dim3((unsigned)wgs), dim3(G::WG), 0, ma.stream, km);
  return hipGetLastError();
}

template <class A, int KSH> hipError_t launch_team_mul(const MulArgs &ma)
{
  if constexpr(!(A::kCompact || A::kIntWide)) {
    return hipErrorNotSupported;
  } else {
    const uint64_t nl = (uint64_t)(ma.nlimbs > 0 ? ma.nlimbs : 1);
    if(nl > (uint64_t)kMaxLimbs || !ma.team_ctl || ma.logn < (uint32_t)kTeamBlock + 3 || ma.logn > (uint32_t)kTeamBlock + 5 ||
       nl * ma.batch >= (1ull << 31)) {
      return hipErrorNotSupported;
    }
    KTeamMul<A> kt{};
    kt.m.k.a               = ma.a;
    const LimbRec<A> *recs = static_cast<const LimbRec<A> *>(ma.limbs);
    for(uint64_t l = 0; l < nl; l++) kt.m.k.limbs[l] = recs[l];
    kt.m.k.limb_stride = nl > 1 ? ma.limb_stride : 0;
    kt.m.k.poly_stride = ma.poly_stride ? ma.poly_stride : (1ull << ma.logn);
    kt.m.k.logn        = ma.logn;
    kt.m.k.s0          = ma.logn - (uint32_t)kTeamBlock;
    kt.m.k.lazy        = 0;
    kt.m.k.nblocks     = ma.batch;
    kt.m.b             = ma.b;
    kt.m.out           = ma.out;
    kt.m.b_limb_stride = nl > 1 ? ma.b_limb_stride : 0;
    kt.m.lazy_in       = (uint32_t)ma.lazy_in;
    kt.m.b_bcast       = (uint32_t)ma.b_bcast;
    kt.m.accumulate    = (uint32_t)ma.accumulate;
    kt.ctl             = static_cast<TeamCtl *>(ma.team_ctl);
    kt.lag             = (uint32_t)(ma.team_lag > 0 ? ma.team_lag : 8);
    kt.nlimbs          = (uint32_t)nl;
    kt.poly_major      = nl > 1 && kt.m.k.poly_stride > kt.m.k.limb_stride;
    kt.split_rcp       = team_split_rcp(kt.poly_major ? nl : ma.batch);
    const size_t bytes = sizeof(TeamCtl) + (size_t)(nl * ma.batch) * sizeof(unsigned);
    hipError_t   e     = team_ctl_clear(ma.team_ctl, bytes, ma.stream);
    if(e != hipSuccess) return e;
    uint64_t wgs = (uint64_t)(ma.num_cus > 0 ? ma.num_cus : 256) * (ma.team_wpc > 0 ? ma.team_wpc : 4);
    if(ma.max_grid > 0) wgs = (uint64_t)ma.max_grid;
    kt.m.k.wgs_per_limb = (uint32_t)wgs;
    const dim3 g((unsigned)wgs), t(256);
#define NTT_TEAM_MUL(LEADV)                                                                                  \
  do {                                                                                                       \
    if(nl > 1) hipLaunchKernelGGL((team_mul_kernel<A, LEADV, KSH, true>), g, t, 0, ma.stream, kt);           \
    else hipLaunchKernelGGL((team_mul_kernel<A, LEADV, KSH, false>), g, t, 0, ma.stream, kt);                \
  } while(0)
    switch(ma.logn - kTeamBlock) {
      case 3: NTT_TEAM_MUL(3); break;
      case 4: NTT_TEAM_MUL(4); break;
      default: NTT_TEAM_MUL(5); break;
    }
#undef NTT_TEAM_MUL
    return hipGetLastError();
  }
}

template <class A, int KSH> hipError_t launch_fwd_mul_impl(const MulArgs &ma)
{
  if(ma.nlimbs > kMaxLimbs) return hipErrorInvalidValue;
  if(ma.team_ctl) return launch_team_mul<A, KSH>(ma);
  if(ma.logn > (uint32_t)kFusedMax) {
    if(ma.block_log == (uint32_t)kFusedSmallBlock) return launch_fwd_mul_blocks<A, kFusedSmallBlock, KSH>(ma);
    if(ma.block_log == (uint32_t)kFusedLarge) return launch_fwd_mul_blocks<A, kFusedLarge, KSH>(ma);
    return hipErrorInvalidValue;
  }
  switch(ma.logn) {
#define NTT_MUL_CASE(LN) \
  case LN: return launch_fwd_mul_blocks<A, LN, KSH>(ma);
    NTT_MUL_CASE(6) NTT_MUL_CASE(7) NTT_MUL_CASE(8) NTT_MUL_CASE(9) NTT_MUL_CASE(10) NTT_MUL_CASE(11) NTT_MUL_CASE(12) NTT_MUL_CASE(13)
    NTT_MUL_CASE(14)
#undef NTT_MUL_CASE
    default: return hipErrorNotSupported;
  }
}

#define NTT_DEFINE_LAUNCH_FWD_MUL(A, KSH) \
  template <> hipError_t launch_fwd_mul<A, KSH>(const MulArgs &ma) { return launch_fwd_mul_impl<A, KSH>(ma); }

#define NTT_DEFINE_LAUNCH_DOT(A, KSH) \
  template <> hipError_t launch_dot<A, KSH>(const DotArgs &da) { return launch_dot_impl<A, KSH>(da); }

#define NTT_DEFINE_LAUNCH_PRODUCT(A, KSH) \
  template <> hipError_t launch_product<A, KSH>(const ProdArgs &pa) { return launch_product_impl<A, KSH>(pa); }
/* (a translation unit of its own per policy: inst_team_*.hip) */
#define NTT_DEFINE_LAUNCH_TEAM_PRODUCT(A, KSH) \
  template <> hipError_t launch_team_product<A, KSH>(const ProdArgs &pa) { return launch_team_product_impl<A, KSH>(pa); }

/* body of launch_pass<A,KSH>; each instantiating .hip file expands this once */
#define NTT_DEFINE_LAUNCH_PASS(A, KSH)                                                   \
  template <> hipError_t launch_pass<A, KSH>(const PassArgs &pa)                         \
  {                                                                                      \
    if(pa.fused == 4) return pa.inverse ? launch_onepass<A, true, KSH>(pa) : launch_onepass<A, false, KSH>(pa); \
    if(pa.fused == 3) {                                                                  \
      switch(pa.r) {                                                                     \
        case 3: return pa.inverse ? launch_team<A, 3, true, KSH>(pa) : launch_team<A, 3, false, KSH>(pa); \
        case 4: return pa.inverse ? launch_team<A, 4, true, KSH>(pa) : launch_team<A, 4, false, KSH>(pa); \
        case 5: return pa.inverse ? launch_team<A, 5, true, KSH>(pa) : launch_team<A, 5, false, KSH>(pa); \
        default: return hipErrorInvalidValue;                                            \
      }                                                                                  \
    }                                                                                    \
    if(pa.fused == 2) {                                                                  \
      switch(pa.r) {                                                                     \
        case 1: return pa.inverse ? launch_twophase<A, 1, true, KSH>(pa) : launch_twophase<A, 1, false, KSH>(pa); \
        case 2: return pa.inverse ? launch_twophase<A, 2, true, KSH>(pa) : launch_twophase<A, 2, false, KSH>(pa); \
        case 3: return pa.inverse ? launch_twophase<A, 3, true, KSH>(pa) : launch_twophase<A, 3, false, KSH>(pa); \
        default: return hipErrorInvalidValue;                                            \
      }                                                                                  \
    }                                                                                    \
    if(pa.fused) {                                                                       \
      switch(pa.r) {                                                                     \
        NTT_FUSED_CASES(A, KSH)                                                          \
        default: return hipErrorInvalidValue;                                            \
      }                                                                                  \
    }                                                                                    \
    switch(pa.r) {                                                                       \
      case 1: return pa.inverse ? launch_column<A, 1, true, KSH>(pa) : launch_column<A, 1, false, KSH>(pa); \
      case 2: return pa.inverse ? launch_column<A, 2, true, KSH>(pa) : launch_column<A, 2, false, KSH>(pa); \
      case 3: return pa.inverse ? launch_column<A, 3, true, KSH>(pa) : launch_column<A, 3, false, KSH>(pa); \
      case 4: return pa.inverse ? launch_column<A, 4, true, KSH>(pa) : launch_column<A, 4, false, KSH>(pa); \
      default: return hipErrorInvalidValue;                                              \
    }                                                                                    \
  }

/* the radix-4 formulation (ArithU64R4): block passes, and column passes of one or two radix-4 levels before (forward) or
 * after (inverse) them (ntt_passplan.h: make_passes_r4) */
#define NTT_DEFINE_LAUNCH_PASS_RADIX4(A, KSH)                                            \
  template <> hipError_t launch_pass<A, KSH>(const PassArgs &pa)                         \
  {                                                                                      \
    if(pa.fused == 1) {                                                                  \
      switch(pa.r) {                                                                     \
        NTT_FUSED_CASES(A, KSH)                                                          \
        default: return hipErrorInvalidValue;                                            \
      }                                                                                  \
    }                                                                                    \
    if(pa.fused || pa.s != 0) return hipErrorInvalidValue;                               \
    switch(pa.r) {                                                                       \
      case 2: return pa.inverse ? launch_column<A, 2, true, KSH>(pa) : launch_column<A, 2, false, KSH>(pa); \
      case 4: return pa.inverse ? launch_column<A, 4, true, KSH>(pa) : launch_column<A, 4, false, KSH>(pa); \
      default: return hipErrorInvalidValue;                                              \
    }                                                                                    \
  }

#define NTT_FUSED_CASE(A, KSH, LN) \
  case LN: return pa.inverse ? launch_fused<A, LN, true, KSH>(pa) : launch_fused<A, LN, false, KSH>(pa);

#define NTT_FUSED_CASES(A, KSH)                                                        \
  NTT_FUSED_CASE(A, KSH, 6) NTT_FUSED_CASE(A, KSH, 7) NTT_FUSED_CASE(A, KSH, 8)        \
  NTT_FUSED_CASE(A, KSH, 9) NTT_FUSED_CASE(A, KSH, 10) NTT_FUSED_CASE(A, KSH, 11)      \
  NTT_FUSED_CASE(A, KSH, 12) NTT_FUSED_CASE(A, KSH, 13) NTT_FUSED_CASE(A, KSH, 14)

} /* namespace ntt */
