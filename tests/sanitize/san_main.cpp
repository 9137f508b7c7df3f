/*
 * san_main.cpp -- AddressSanitizer + UndefinedBehaviorSanitizer run of the CPU-side code (TEST INFRASTRUCTURE).
 *
 * Mirrors the reference's sanitizer builds (cmake/compilation-flags.cmake:24-57, tests/pre-commit-script.sh:28-33)
 * for what exists on the CPU here: the oracle (oracle/ntt_oracle.c), the host-side table/pass planning of the
 * product (csrc/ntt_tables.h, ntt_passplan.h) and the kernel templates executed by the emulator (tests/emu/emu.cpp,
 * i.e. csrc/ntt_core.h + ntt_arith.h: index maps, LDS layouts, twiddle addressing).  GPU code cannot run under
 * sanitizers on this pool; this run proves the shared templates free of out-of-bounds accesses, signed overflow
 * and misaligned accesses on the host.  Built and run by tests/test_sanitize.py (-m "not gpu") / `make sanitize`.
 */
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ntt_oracle.h"

extern "C" {
int  emu_transform(uint64_t *a, uint64_t batch, int m, uint64_t q, uint64_t root, int arith, int inverse, int generic,
                   int wide, int ksh_force);
void emu_set_lazy(int on);
int  emu_pointwise(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t q, int arith);
int  emu_pointwise_lazy(uint64_t *c, const uint64_t *a, const uint64_t *b, uint64_t n, uint64_t q, int arith);
void emu_expand_radix4(uint64_t *e, const uint64_t *w, uint64_t N, uint64_t q);
int  emu_plan_info(int logn, uint64_t *info);
int  emu_inv_dot(uint64_t *out, int k, const uint64_t *a, const uint64_t *b, uint64_t batch, int m, uint64_t q, uint64_t root, int arith,
                 int lazy, int bcast);
int  emu_fwd_mul(uint64_t *out, uint64_t *a, const uint64_t *b, uint64_t batch, int m, uint64_t q, uint64_t root, int arith, int lazy,
                 int bcast, int acc);
}

static int g_fail = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if(!(cond)) {                                                          \
      fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);   \
      g_fail++;                                                            \
    }                                                                      \
  } while(0)

struct Case {
  int      m;
  uint64_t q;
};

int main()
{
  /* (m, q): small and odd/even sizes of every kernel family, a 51-bit and a 59-bit modulus, two multi-pass sizes */
  /* q = 0: the largest 59-bit prime for that size (exercises the carry of the 128-bit double product) */
  const Case cases[] = {{4, 0x10001},         {6, 0x10001},         {7, 0x7ffe0001},        {8, 0x1ffc8001}, {9, 0x7fffffffe0001},
                        {12, 0x80000001c0001}, {12, 0x7fffffffe0001}, {15, 0x7fffffffe0001}, {8, 0},          {12, 0},
                        {16, 0x7fffffffe0001}};
  for(const Case &cs : cases) {
    const uint64_t n = 1ull << cs.m, q = cs.q ? cs.q : orc_find_prime(59, n, 0);
    if(!orc_is_prime(q) || (q - 1) % (2 * n)) {
      fprintf(stderr, "skipping unusable case m=%d q=%llx\n", cs.m, (unsigned long long)q);
      continue;
    }
    const uint64_t w = orc_min_root(q, n);
    CHECK(w != 0);
    orc_ctx *cx = orc_ctx_new(n, q, w);
    CHECK(cx != nullptr);
    const uint64_t        batch = 2;
    std::vector<uint64_t> a(batch * n), ref(batch * n), got(batch * n), e(2 * n);
    orc_fill_uniform(a.data(), batch * n, q, 0x5eed, cs.m);
    ref = a;
    orc_fwd_r2_batch(ref.data(), batch, cx);
    emu_expand_radix4(e.data(), cx->w, n, q);
    CHECK(memcmp(e.data(), cx->e, 2 * n * sizeof(uint64_t)) == 0);
    for(int arith : {0, 1, 3}) {
      if(arith == 1 && q > ((1ull << 51) + (1ull << 41))) continue;
      /* radix-4 formulation: single-pass sizes and 2^16 (column pass of one radix-4 level + 2^14 blocks; 2^15 would need
       * the 2^13 block size, which the instrumented build leaves out) */
      if(arith == 3 && (cs.m < 6 || cs.m == 15)) continue;
      for(int generic = 0; generic < 2; generic++) {
        if(generic && (arith == 3 || cs.m > 12)) continue;
        got = a;
        CHECK(emu_transform(got.data(), batch, cs.m, q, w, arith, 0, generic, 0, -1) == 0);
        CHECK(got == ref);
        CHECK(emu_transform(got.data(), batch, cs.m, q, w, arith, 1, generic, 0, -1) == 0);
        CHECK(got == a);
      }
      /* lazy outputs, fed back as wide inputs */
      got = a;
      emu_set_lazy(1);
      CHECK(emu_transform(got.data(), batch, cs.m, q, w, arith, 0, 0, 0, -1) == 0);
      emu_set_lazy(0);
      for(uint64_t i = 0; i < batch * n; i++) CHECK(got[i] < 8 * q && got[i] % q == ref[i]);
      CHECK(emu_transform(got.data(), batch, cs.m, q, w, arith, 1, 0, 1, -1) == 0);
      CHECK(got == a);
    }
    /* the wide integer policy (ArithU64X<3>, checked) for the 59-bit cases' sizes with a 57-bit modulus */
    if(cs.q == 0) {
      const uint64_t q57 = orc_find_prime(57, n, 0), w57 = orc_min_root(q57, n);
      orc_ctx *      c57 = orc_ctx_new(n, q57, w57);
      CHECK(c57 != nullptr);
      std::vector<uint64_t> a57(batch * n), r57;
      orc_fill_uniform(a57.data(), batch * n, q57, 0x57, cs.m);
      r57 = a57;
      orc_fwd_r2_batch(r57.data(), batch, c57);
      for(int generic = 0; generic < 2; generic++) {
        got = a57;
        CHECK(emu_transform(got.data(), batch, cs.m, q57, w57, 6, 0, generic, 0, 3) == 0);
        CHECK(got == r57);
        CHECK(emu_transform(got.data(), batch, cs.m, q57, w57, 6, 1, generic, 0, 3) == 0);
        CHECK(got == a57);
      }
      orc_ctx_free(c57);
    }
    /* pointwise products, strict and lazy operands */
    std::vector<uint64_t> b(n), c(n), c2(n);
    orc_fill_uniform(b.data(), n, q, 0xb, 0);
    orc_pointwise(c.data(), a.data(), b.data(), n, q);
    for(int arith : {0, 1}) {
      if(arith == 1 && q > ((1ull << 51) + (1ull << 41))) continue;
      CHECK(emu_pointwise(c2.data(), a.data(), b.data(), n, q, arith) == 0);
      CHECK(c2 == c);
      std::vector<uint64_t> al(a.begin(), a.begin() + n), bl(b);
      for(uint64_t i = 0; i < n; i++) {
        al[i] += (i % 4) * q;
        bl[i] += ((i / 4) % 4) * q;
      }
      CHECK(emu_pointwise_lazy(c2.data(), al.data(), bl.data(), n, q, arith) == 0);
      CHECK(c2 == c);
    }
    /* products of operands in the NTT domain (dot_inv_kernel, fwd_mul_kernel as the emulator runs them): the operand
     * layouts, the broadcast index and the lazy folds under the sanitizers, against the oracle (block sizes 2^6, 2^8, 2^12) */
    if(cs.m == 6 || cs.m == 8 || cs.m == 12) {
      for(int arith : {0, 1}) {
        if(arith == 1 && q > ((1ull << 51) + (1ull << 41))) continue;
        for(int bcast = 0; bcast < 2; bcast++) {
          const int             k = 3;
          std::vector<uint64_t> da((size_t)k * batch * n), db((size_t)k * (bcast ? n : batch * n)), sum(batch * n, 0), t(batch * n);
          orc_fill_uniform(da.data(), da.size(), q, 0xd07, cs.m);
          orc_fill_uniform(db.data(), db.size(), q, 0xd08, cs.m);
          for(int i = 0; i < k; i++) {
            for(uint64_t pb = 0; pb < batch; pb++) {
              orc_pointwise(t.data() + pb * n, da.data() + (size_t)i * batch * n + pb * n,
                            db.data() + (size_t)i * (bcast ? n : batch * n) + (bcast ? 0 : pb * n), n, q);
            }
            for(uint64_t j = 0; j < batch * n; j++) sum[j] = (sum[j] + t[j]) % q;
          }
          std::vector<uint64_t> want(sum), lz_a(da), lz_b(db);
          orc_inv_r2_batch(want.data(), batch, cx);
          for(size_t j = 0; j < lz_a.size(); j++) lz_a[j] += (j % 4) * q;   /* lazy words: anywhere in [0,4q) */
          for(size_t j = 0; j < lz_b.size(); j++) lz_b[j] += ((j / 3) % 4) * q;
          CHECK(emu_inv_dot(got.data(), k, da.data(), db.data(), batch, cs.m, q, w, arith, 0, bcast) == 0);
          CHECK(got == want);
          CHECK(emu_inv_dot(got.data(), k, lz_a.data(), lz_b.data(), batch, cs.m, q, w, arith, 1, bcast) == 0);
          CHECK(got == want);
          /* c^ = fwd(a) (.) b^ + c^ */
          std::vector<uint64_t> ac(a), acc0(batch * n), exp(batch * n);
          orc_fill_uniform(acc0.data(), acc0.size(), q, 0xacc, cs.m);
          for(uint64_t pb = 0; pb < batch; pb++) orc_pointwise(exp.data() + pb * n, ref.data() + pb * n, db.data() + (bcast ? 0 : pb * n), n, q);
          for(uint64_t j = 0; j < batch * n; j++) exp[j] = (exp[j] + acc0[j]) % q;
          got = acc0;
          CHECK(emu_fwd_mul(got.data(), ac.data(), lz_b.data(), batch, cs.m, q, w, arith, 1, bcast, 1) == 0);
          CHECK(got == exp);
        }
      }
    }
    /* the CPU timing harness of bench.py's cpu_baseline leg (oracle/cpu_bench.inc): a few milliseconds of each unit of
     * work on two threads, and the single-thread method, under the sanitizers */
    if(cs.m >= 10 && cs.m <= 12) {
      for(int op = 0; op < 3; op++) {
        double out[5] = {0, 0, 0, 0, 0};
        CHECK(orc_bench_threads(op, n, q, cx->ninv, cx->e, cx->econ, cx->einv, cx->einv_con, 2, 0.02, 2, out) == 0);
        CHECK(out[0] >= 2 && out[2] == 2 && out[3] >= 1);
        CHECK(orc_bench_single(op, n, q, cx->ninv, cx->e, cx->econ, cx->einv, cx->einv_con, 1, 2, 2) > 0);
      }
    }
    orc_ctx_free(cx);
  }
  uint64_t info[10];
  for(int ln = 6; ln <= 14; ln++) {
    CHECK(emu_plan_info(ln, info) == 0);
    CHECK(info[9] == 1); /* every LDS exchange layout conflict-free */
  }
  if(g_fail) {
    fprintf(stderr, "%d check(s) failed\n", g_fail);
    return 1;
  }
  puts("sanitize: ok");
  return 0;
}
