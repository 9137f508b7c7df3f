/*
 * emu.cpp -- CPU emulation of the gfx950 kernel logic (TEST INFRASTRUCTURE).
 *
 * Compiles the very same templates the HIP kernels are built from
 * (csrc/ntt_core.h, ntt_arith.h, ntt_passplan.h, ntt_tables.h) with g++ and runs
 * them thread-by-thread, phase-by-phase, so that index maps, LDS exchange
 * layouts, twiddle addressing, the FP64 exactness argument and its reduction
 * schedule can be checked against the oracle without a GPU.  Not part of the
 * product library; built by tests/emu/Makefile, driven by tests/test_emu.py.
 */
#include <array>
#include <cstdio>
#include <cstring>
#include <vector>

#include "ntt_core.h"
#include "ntt_passplan.h"
#include "ntt_tables.h"

using namespace ntt;

template <class A> struct Regs {
  typename A::val x[kE];
};

template <class A, int LOGN, bool INV, int KSH> static void emu_fused(const Params<A> &p)
{
  using P                 = Plan<LOGN>;
  constexpr uint32_t MASK = fused_mask<A, LOGN, INV, KSH>();
  std::vector<typename A::val> lds(P::LDS_ELEMS);
  std::vector<Regs<A>>         regs(P::T);
  for(uint64_t b = 0; b < p.nblocks; b++) {
    const uint32_t blk  = (uint32_t)(b & ((1ull << p.s0) - 1));
    uint64_t *     base = p.a + (b << LOGN);
    if constexpr(!INV) {
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        global_load_first<A, LOGN, false>(regs[t].x, t, base, p.wide, p.c);
        run_group<A, LOGN, 0, false, MASK>(regs[t].x, t, blk, p);
      }
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int G = decltype(gg)::value;
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G + 1>(regs[t].x, t, lds.data());
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
          lds_gather<A, LOGN, G, G + 1>(regs[t].x, t, lds.data());
          run_group<A, LOGN, G + 1, false, MASK>(regs[t].x, t, blk, p);
        }
      });
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) global_store_last<A, LOGN, false>(regs[t].x, t, base, p.c);
    } else {
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
        global_load_last<A, LOGN, true>(regs[t].x, t, base, p.wide, p.c);
        run_group<A, LOGN, P::NG - 1, true, MASK>(regs[t].x, t, blk, p);
      }
      static_for<0, P::NG - 1>([&](auto gg) {
        constexpr int G = P::NG - 1 - decltype(gg)::value; /* writer */
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) lds_scatter<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
        for(uint32_t t = 0; t < (uint32_t)P::T; t++) {
          lds_gather<A, LOGN, G, G - 1>(regs[t].x, t, lds.data());
          run_group<A, LOGN, G - 1, true, MASK>(regs[t].x, t, blk, p);
        }
      });
      for(uint32_t t = 0; t < (uint32_t)P::T; t++) global_store_first<A, LOGN, true>(regs[t].x, t, base, p.c);
    }
  }
}

template <class A, int R, bool INV, int KSH>
static void emu_column(uint64_t *a, uint64_t batch, uint32_t logn, uint32_t S, bool wide, bool lastinv,
                       const typename A::tw *tab, const typename A::consts &c)
{
  constexpr uint32_t MASK = column_mask<A, R, INV, KSH>();
  const uint64_t     cols = (1ull << logn) >> R;
  for(uint64_t pidx = 0; pidx < batch; pidx++) {
    for(uint64_t col = 0; col < cols; col++) {
      column_pass_thread<A, R, INV, MASK>(a + (pidx << logn), (uint32_t)col, logn, S, wide, lastinv, tab, c);
    }
  }
}

template <class A, bool INV, int KSH>
static int emu_run(uint64_t *a, uint64_t batch, int m, const typename A::tw *tab,
                   const typename A::consts &c, bool generic, bool wide, const typename A::ctw *tab8 = nullptr)
{
  const PassList L = make_passes(m, generic);
  for(int k = 0; k < L.n; k++) {
    const Pass &ps      = L.p[INV ? L.n - 1 - k : k];
    const bool  lastinv = INV && ps.s == 0;
    /* only the first pass of a transform sees caller data */
    const bool w = wide && k == 0;
    if(ps.fused) {
      Params<A> p{};
      p.a       = a;
      p.tw      = tab;
      p.tw8     = tab8;
      p.c       = c;
      p.logn    = (uint32_t)m;
      p.s0      = (uint32_t)ps.s;
      p.wide    = w;
      p.lastinv = lastinv;
      p.nblocks = batch << ps.s;
      switch(ps.r) {
#define CASE(LN) \
  case LN: emu_fused<A, LN, INV, KSH>(p); break;
        CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
        default: return -1;
      }
    } else {
      switch(ps.r) {
        case 1: emu_column<A, 1, INV, KSH>(a, batch, m, ps.s, w, lastinv, tab, c); break;
        case 2: emu_column<A, 2, INV, KSH>(a, batch, m, ps.s, w, lastinv, tab, c); break;
        case 3: emu_column<A, 3, INV, KSH>(a, batch, m, ps.s, w, lastinv, tab, c); break;
        case 4: emu_column<A, 4, INV, KSH>(a, batch, m, ps.s, w, lastinv, tab, c); break;
        default: return -1;
      }
    }
  }
  return 0;
}

/* plan introspection for the layout tests: returns LDS row pad, fills info[]:
 * {NG, R0, RL, T, ROW, LDS_ELEMS, wave_local bits, f64 fwd mask ksh0, f64 inv mask ksh0} */
template <int LOGN> static void plan_info(uint64_t *info)
{
  using P     = Plan<LOGN>;
  info[0]     = P::NG;
  info[1]     = P::R0;
  info[2]     = P::RL;
  info[3]     = P::T;
  info[4]     = P::ROW;
  info[5]     = P::LDS_ELEMS;
  uint64_t wl = 0;
  for(int g = 0; g + 1 < P::NG; g++) wl |= (uint64_t)P::WAVE_LOCAL(g, g + 1) << g;
  info[6] = wl;
  info[7] = fused_mask<ArithF64, LOGN, false, 0>();
  info[8] = fused_mask<ArithF64, LOGN, true, 0>();
  info[9] = P::layout_conflict_free();
}

extern "C" {

/* arith: 0 = U64, 1 = F64 (ksh: -1 = class of q, else forced class <= class of q)
 * returns 0 on success, <0 if the request is not representable */
int emu_transform(uint64_t *a, uint64_t batch, int m, uint64_t q, uint64_t root, int arith,
                  int inverse, int generic, int wide, int ksh_force)
{
  const uint64_t N    = 1ull << m;
  const uint64_t rinv = h_powmod(root, q - 2, q);
  const auto     w    = h_power_table(root, N, q);
  const auto     wi   = h_power_table(rinv, N, q);
  const auto &   src  = inverse ? wi : w;
  if(arith == 0) {
    std::vector<TwU64> tab(N);
    for(uint64_t i = 0; i < N; i++) tab[i] = h_tw_u64(src[i], q);
    const auto c = h_consts_u64(q, N, wi);
    return inverse ? emu_run<ArithU64, true, 0>(a, batch, m, tab.data(), c, generic, wide)
                   : emu_run<ArithU64, false, 0>(a, batch, m, tab.data(), c, generic, wide);
  }
  if(!h_f64_eligible(q)) return -2;
  std::vector<TwF64>  tab(N);
  std::vector<double> tab8(N);
  for(uint64_t i = 0; i < N; i++) {
    tab[i]  = h_tw_f64(src[i], q);
    tab8[i] = tab[i].w;
  }
  const auto c   = h_consts_f64(q, N, wi);
  int        ksh = h_f64_ksh(q);
  if(ksh_force >= 0) {
    if(ksh_force > ksh) return -3;
    ksh = ksh_force;
  }
  const int cls = ksh >= 18 ? 18 : (ksh >= 1 ? 1 : 0);
#define RUN(K)                                                                              \
  return inverse ? emu_run<ArithF64, true, K>(a, batch, m, tab.data(), c, generic, wide, tab8.data())   \
                 : emu_run<ArithF64, false, K>(a, batch, m, tab.data(), c, generic, wide, tab8.data());
  if(cls == 18) { RUN(18) }
  if(cls == 1) { RUN(1) }
  RUN(0)
#undef RUN
}

/* pointwise product through both arithmetic policies */
int emu_pointwise(uint64_t *c_out, const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t q, int arith)
{
  std::vector<uint64_t> dummy(2, 1);
  if(arith == 0) {
    const auto c = h_consts_u64(q, 2, dummy);
    for(uint64_t i = 0; i < n; i++) c_out[i] = ArithU64::mulmod_full(a[i], b[i], c);
    return 0;
  }
  if(!h_f64_eligible(q)) return -2;
  const auto c = h_consts_f64(q, 2, dummy);
  for(uint64_t i = 0; i < n; i++) c_out[i] = ArithF64::mulmod_full(a[i], b[i], c);
  return 0;
}

int emu_plan_info(int logn, uint64_t *info)
{
  switch(logn) {
#define CASE(LN) \
  case LN: plan_info<LN>(info); return 0;
    CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13) CASE(14)
#undef CASE
    default: return -1;
  }
}
}
