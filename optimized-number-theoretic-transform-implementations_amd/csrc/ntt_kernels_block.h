/*
 * ntt_kernels_block.h -- the block kernels: launch geometry (Geom), kernel arguments, coefficient loads / stores / exchanges, fused_kernel (one workgroup
 * transforms whole 2^LOGN blocks), twophase_kernel (both passes of 2^16 / 2^17 inside one workgroup), onepass_kernel (2^15 in one pass).
 * Part of ntt_kernels.h (included from there, in this order: block, team, products, launch); not a header of its own.
 */
#pragma once

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
/* one half (block position blk of a 2^15-point polynomial, s0 = 1) through the forward block stages and out to memory (`out`) */
template <class A, uint32_t MASK, class OUT>
__device__ __forceinline__ void onepass_block_fwd(typename A::val (&x)[kE], uint32_t blk, uint32_t tid, const Params<A> &p, typename A::val *lds_all,
                                                  typename A::ctw *tabl, OUT out)
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
  out(x, tid); /* the half's outputs: whole-line stores (transforms), or the product by b^ where they would be reduced and stored */
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

/* the forward polynomial loop of the one-pass kernels: p.a = the coefficients; out(x, tl, h, off) disposes of half h of the polynomial
 * whose word offset (from every operand's base) is off */
template <class A, int KSH, class OUT>
__device__ __forceinline__ void onepass_forward(Params<A> &p, uint32_t bid, uint32_t gdim, uint32_t tid, typename A::val *lds_all, typename A::ctw *tabl,
                                                OUT out)
{
  constexpr int      LOGN = kFusedLarge;
  constexpr uint64_t HALF = 1ull << LOGN;
  constexpr uint32_t M15  = onepass_fwd_mask<A, KSH>();
  constexpr uint32_t MASK = M15 >> 1;
  constexpr bool     RED0 = (M15 & 1u) != 0;
  p.s0          = 1;
  uint64_t poly = bid;
  if(!below(poly, p.nblocks)) return;
  /* polynomial offsets one polynomial ahead of the loads, as in fused_kernel's loops (a pointer batch reads them from a table) */
  const auto next_poly = [&](uint64_t at) -> uint64_t { return below(at + gdim, p.nblocks) ? at + gdim : at; };
  uint64_t off_cur = poly_offset<true>(poly, p.pstride, p.ptab), off_nxt = poly_offset<true>(next_poly(poly), p.pstride, p.ptab);
  /* ONE copy of the block body: the two halves are the iterations of a loop that is not unrolled, the half that waits parked as bit
   * patterns in `hold` -- two inlined copies let the compiler hoist either copy's lane offsets and LDS addresses out of the
   * polynomial loop, into registers the halves need (79 spilled VGPRs in the first version of this kernel) */
  uint64_t hold[kE], rb[kE];
  const auto bits = [](typename A::val v) -> uint64_t { return __builtin_bit_cast(uint64_t, v); };
  const auto vals = [](uint64_t u) -> typename A::val { return __builtin_bit_cast(typename A::val, u); };
  prefetch_first<LOGN>(hold, tid, p.a + off_cur);
  prefetch_first<LOGN>(rb, tid, p.a + off_cur + HALF);
  for(; below(poly, p.nblocks); poly += gdim) {
    const uint64_t  off  = off_cur;
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
      onepass_block_fwd<A, MASK>(x, h, tl, p, lds_all, tabl, [&](typename A::val(&y)[kE], uint32_t t2) { out(y, t2, h, off, poly); });
      if(h == 0) {
        static_for<0, kE>([&](auto ee) { x[decltype(ee)::value] = vals(hold[decltype(ee)::value]); });
        prefetch_first<LOGN>(hold, tl, nxt, more); /* half A has been stored: the next polynomial's first half */
      }
    }
    prefetch_first<LOGN>(rb, tid, nxt + HALF, more); /* ... and its second half, behind half B's stores */
  }
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
  constexpr uint64_t HALF     = 1ull << LOGN;
  if constexpr(!INV) {
    onepass_forward<A, KSH>(p, bid, gdim, tid, lds_all, tabl, [&](typename A::val(&y)[kE], uint32_t tl, uint32_t h, uint64_t off, uint64_t) {
      store_last_whole_lines<A, LOGN, false>(y, tl, p.a + off + (h ? HALF : 0), p.c, false);
    });
  } else {
    constexpr uint32_t MASK = fused_mask<A, LOGN, true, KSH>() | (A::kWide52 ? kCanonInFlag : 0u); /* the blocks do not end the transform; canonical inputs */
    constexpr bool     LTW  = G::LDS_TW > 0;
    p.s0          = 1;
    uint64_t poly = bid;
    if(!below(poly, p.nblocks)) return;
    const auto next_poly = [&](uint64_t at) -> uint64_t { return below(at + gdim, p.nblocks) ? at + gdim : at; };
    uint64_t off_cur = poly_offset<true>(poly, p.pstride, p.ptab), off_nxt = poly_offset<true>(next_poly(poly), p.pstride, p.ptab);
    /* (the halves as the iterations of a loop that is not unrolled, as in onepass_forward: half B's raw words, then half A's results,
     * wait as bit patterns in `hold`) */
    uint64_t hold[kE], rb[kE];
    const auto bits = [](typename A::val v) -> uint64_t { return __builtin_bit_cast(uint64_t, v); };
    const auto vals = [](uint64_t u) -> typename A::val { return __builtin_bit_cast(typename A::val, u); };
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

} /* namespace ntt */
