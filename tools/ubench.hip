/*
 * ubench.hip -- gfx950 micro-benchmarks that decide the NTT arithmetic:
 *   1. issue cost (cycles per wave64 instruction per SIMD) of the integer-multiply and FP64
 *      instructions the two arithmetic policies are made of;
 *   2. register-resident butterfly throughput of ArithU64 and ArithF64 as compiled;
 *   3. plain HBM copy bandwidth (the practical ceiling for the roofline fraction).
 * Build: make ubench   Run (GPU box): build/ubench
 */
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ntt_core.h"
#include "ntt_tables.h"

using namespace ntt;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e = (x);                                                         \
    if(e != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                    \
      exit(1);                                                                  \
    }                                                                           \
  } while(0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

#define OP_KERNEL(NAME, DECL, INIT, ASM, SINK)                                   \
  __global__ void __launch_bounds__(256) k_##NAME(uint64_t *out, uint32_t seed) \
  {                                                                             \
    DECL;                                                                       \
    INIT;                                                                       \
    for(int it = 0; it < ITERS; it++) {                                         \
      _Pragma("unroll") for(int c = 0; c < CHAINS; c++) { ASM; }                \
    }                                                                           \
    uint64_t s = 0;                                                             \
    _Pragma("unroll") for(int c = 0; c < CHAINS; c++) { SINK; }                 \
    if(s == 0x123456789abcdefULL) out[threadIdx.x] = s;                         \
  }

/* 32-bit integer ops */
#define DECL_U32 uint32_t a[CHAINS], b = seed | 1u, d = seed * 3u + 7u
#define INIT_U32 _Pragma("unroll") for(int c = 0; c < CHAINS; c++) a[c] = seed + c * 977u + threadIdx.x
#define SINK_U32 s += a[c]
OP_KERNEL(mul_lo_u32, DECL_U32, INIT_U32, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b)), SINK_U32)
OP_KERNEL(mul_hi_u32, DECL_U32, INIT_U32, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b)), SINK_U32)
OP_KERNEL(mul_u32_u24, DECL_U32, INIT_U32, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(b)), SINK_U32)
OP_KERNEL(mul_hi_u32_u24, DECL_U32, INIT_U32, asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[c]) : "v"(b)), SINK_U32)
OP_KERNEL(mad_u32_u24, DECL_U32, INIT_U32, asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(d)), SINK_U32)
OP_KERNEL(add_u32, DECL_U32, INIT_U32, asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(b)), SINK_U32)
OP_KERNEL(mov_dpp, DECL_U32, INIT_U32, asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[c])), SINK_U32)
OP_KERNEL(cndmask, DECL_U32, INIT_U32, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[c]) : "v"(b)), SINK_U32)

/* 64-bit integer ops */
#define DECL_U64 uint64_t a[CHAINS]; uint32_t b = seed | 1u, d = seed * 3u + 7u; uint64_t e = seed * 0x9e3779b97f4a7c15ULL
#define INIT_U64 _Pragma("unroll") for(int c = 0; c < CHAINS; c++) a[c] = (uint64_t)seed * 0x12345 + c * 977u + threadIdx.x
#define SINK_U64 s += a[c]
OP_KERNEL(mad_u64_u32, DECL_U64, INIT_U64, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[c]) : "v"(b), "v"(d) : "vcc"), SINK_U64)
OP_KERNEL(lshl_add_u64, DECL_U64, INIT_U64, asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[c]) : "v"(e)), SINK_U64)
OP_KERNEL(cmp_lt_u64, DECL_U64, INIT_U64, asm volatile("v_cmp_lt_u64 vcc, %0, %1" : : "v"(a[c]), "v"(e) : "vcc"), SINK_U64)

/* FP64 ops */
#define DECL_F64 double a[CHAINS]; double b = 1.0 + seed * 1e-9, d = 0.5 + seed * 1e-10
#define INIT_F64 _Pragma("unroll") for(int c = 0; c < CHAINS; c++) a[c] = 1.0 + c * 0.001 + threadIdx.x * 1e-6
#define SINK_F64 s += (uint64_t)a[c]
OP_KERNEL(fma_f64, DECL_F64, INIT_F64, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(d)), SINK_F64)
OP_KERNEL(mul_f64, DECL_F64, INIT_F64, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b)), SINK_F64)
OP_KERNEL(add_f64, DECL_F64, INIT_F64, asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(d)), SINK_F64)
OP_KERNEL(rndne_f64, DECL_F64, INIT_F64, asm volatile("v_rndne_f64 %0, %0" : "+v"(a[c])), SINK_F64)
OP_KERNEL(floor_f64, DECL_F64, INIT_F64, asm volatile("v_floor_f64 %0, %0" : "+v"(a[c])), SINK_F64)
OP_KERNEL(min_f64, DECL_F64, INIT_F64, asm volatile("v_min_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b)), SINK_F64)
OP_KERNEL(cmp_gt_f64, DECL_F64, INIT_F64, asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[c]), "v"(b) : "vcc"), SINK_F64)
OP_KERNEL(fma_f32, DECL_U32, INIT_U32, asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(d)), SINK_U32)
OP_KERNEL(pk_fma_f32, DECL_U64, INIT_U64, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[c]) : "v"(e)), SINK_U64)

/* conversion ops: dst and src differ in width, chain through a dummy */
__global__ void __launch_bounds__(256) k_cvt_f64_u32(uint64_t *out, uint32_t seed)
{
  uint32_t a[CHAINS];
  double   r[CHAINS];
  for(int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x;
  for(int it = 0; it < ITERS; it++) {
#pragma unroll
    for(int c = 0; c < CHAINS; c++) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(r[c]) : "v"(a[c]));
  }
  uint64_t s = 0;
  for(int c = 0; c < CHAINS; c++) s += (uint64_t)r[c];
  if(s == 0x123456789abcdefULL) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_cvt_u32_f64(uint64_t *out, uint32_t seed)
{
  double   a[CHAINS];
  uint32_t r[CHAINS];
  for(int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x;
  for(int it = 0; it < ITERS; it++) {
#pragma unroll
    for(int c = 0; c < CHAINS; c++) asm volatile("v_cvt_u32_f64 %0, %1" : "=v"(r[c]) : "v"(a[c]));
  }
  uint64_t s = 0;
  for(int c = 0; c < CHAINS; c++) s += r[c];
  if(s == 0x123456789abcdefULL) out[threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_bpermute(uint64_t *out, uint32_t seed)
{
  uint32_t a[CHAINS];
  for(int c = 0; c < CHAINS; c++) a[c] = seed + c + threadIdx.x;
  const uint32_t addr = ((threadIdx.x ^ 1) & 63) * 4;
  for(int it = 0; it < ITERS; it++) {
#pragma unroll
    for(int c = 0; c < CHAINS; c++) a[c] = __builtin_amdgcn_ds_bpermute(addr, a[c]);
  }
  uint64_t s = 0;
  for(int c = 0; c < CHAINS; c++) s += a[c];
  if(s == 0x123456789abcdefULL) out[threadIdx.x] = s;
}

/* register-resident butterflies: 16 values/thread, radix-16 tile repeated */
template <class A, uint32_t MASK> __global__ void __launch_bounds__(256) k_bfly(uint64_t *out, typename A::consts c, typename A::tw w0, int reps)
{
  typename A::val x[16];
  typename A::tw  w[4];
  for(int i = 0; i < 16; i++) x[i] = A::template load<false, false>((uint64_t)(threadIdx.x * 16 + i + 1), c);
  for(int i = 0; i < 4; i++) {
    w[i] = w0;
  }
  for(int r = 0; r < reps; r++) {
    static_for<0, 4>([&](auto jj) {
      constexpr int  J   = decltype(jj)::value;
      constexpr int  AB  = 3 - J;
      constexpr bool RED = (MASK >> J) & 1u;
      static_for<0, 16>([&](auto ee) {
        constexpr int E0 = decltype(ee)::value;
        if constexpr(((E0 >> AB) & 1) == 0) A::template fwd_bfly<RED>(x[E0], x[E0 | (1 << AB)], w[J], c);
      });
    });
  }
  uint64_t s = 0;
  for(int i = 0; i < 16; i++) s += A::store_fwd(x[i], c);
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) k_copy(uint4 *dst, const uint4 *src, size_t n)
{
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

/* memory-pipe variants of the copy: 8 B/lane, and 16 B/lane stores at 32-B lane stride (two passes) */
__global__ void __launch_bounds__(256) k_copy8(uint2 *dst, const uint2 *src, size_t n)
{
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void __launch_bounds__(256) k_copy8ld_16st_strided(uint4 *dst, const uint2 *src, size_t n16)
{
  /* each thread: 4 x 8-byte loads (rows 2048 elements apart, like the NTT's first group), then two 16-byte
   * stores at 32-byte lane stride (like its last group) */
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16 / 2; i += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = i >> 10, t = i & 1023;
    const uint2 *s   = src + blk * 4096;
    uint2        a = s[t], b = s[t + 1024], c = s[t + 2048], d = s[t + 3072];
    uint4 *      o = dst + blk * 2048 + 2 * t;
    o[0]           = make_uint4(a.x, a.y, b.x, b.y);
    o[1]           = make_uint4(c.x, c.y, d.x, d.y);
  }
}

/* the same row loads, stores as in the NTT's last group (16 B at 32-B lane stride) or contiguous
 * (16 B at 16-B lane stride: every store instruction writes whole 128-B lines), plain or non-temporal */
/* 16-byte stores whose lane pairs cover whole 32-byte sectors: store A writes the first half of every
 * 64-byte chunk, store B the second half (what a lane-pair swap of the last group's quads would give) */
__global__ void __launch_bounds__(256) k_rows_sector(uint4 *dst, const uint2 *src, size_t n16)
{
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16 / 2; i += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = i >> 10, t = i & 1023;
    const uint2 *s   = src + blk * 4096;
    uint2        a = s[t], b = s[t + 1024], c = s[t + 2048], d = s[t + 3072];
    v4u          v0, v1;
    v0.x = a.x; v0.y = a.y; v0.z = b.x; v0.w = b.y;
    v1.x = c.x; v1.y = c.y; v1.z = d.x; v1.w = d.y;
    v4u *o = (v4u *)dst + blk * 2048 + (t >> 1) * 4 + (t & 1);
    o[0]   = v0;
    o[2]   = v1;
  }
}

template <bool CONTIG, bool NT>
__global__ void __launch_bounds__(256) k_rows(uint4 *dst, const uint2 *src, size_t n16)
{
  typedef unsigned int v4u __attribute__((ext_vector_type(4)));
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16 / 2; i += (size_t)gridDim.x * blockDim.x) {
    const size_t blk = i >> 10, t = i & 1023;
    const uint2 *s   = src + blk * 4096;
    uint2        a = s[t], b = s[t + 1024], c = s[t + 2048], d = s[t + 3072];
    v4u          v0, v1;
    v0.x = a.x; v0.y = a.y; v0.z = b.x; v0.w = b.y;
    v1.x = c.x; v1.y = c.y; v1.z = d.x; v1.w = d.y;
    v4u *o0 = (v4u *)dst + blk * 2048 + (CONTIG ? t : 2 * t);
    v4u *o1 = (v4u *)dst + blk * 2048 + (CONTIG ? t + 1024 : 2 * t + 1);
    if(NT) {
      __builtin_nontemporal_store(v0, o0);
      __builtin_nontemporal_store(v1, o1);
    } else {
      *o0 = v0;
      *o1 = v1;
    }
  }
}

__global__ void __launch_bounds__(256) k_copy_nt(uint4 *dst, const uint4 *src, size_t n)
{
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_nontemporal_load((const v4u *)src + i);
    __builtin_nontemporal_store(v, (v4u *)dst + i);
  }
}
__global__ void __launch_bounds__(256) k_copy_ntst(uint4 *dst, const uint4 *src, size_t n)
{
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(((const v4u *)src)[i], (v4u *)dst + i);
  }
}
/* 4 independent 16-byte transfers per thread per iteration */
__global__ void __launch_bounds__(256) k_copy_x4(uint4 *dst, const uint4 *src, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + 3 * stride < n; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
}

static double clock_ghz = 2.4;

template <class K> static void run_op(const char *name, K kernel, uint64_t *d_out, int instr_per_iter = 1)
{
  const int  blocks = 256 * 4; /* 4 blocks of 4 waves per CU -> 4 waves per SIMD */
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_out, 12345u);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  /* per SIMD: 4 waves * ITERS * CHAINS instructions */
  const double instr_per_simd = 4.0 * ITERS * CHAINS * instr_per_iter;
  const double cyc            = ms * 1e-3 * clock_ghz * 1e9 / instr_per_simd;
  printf("%-16s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at %.2f GHz)\n", name, ms, cyc, clock_ghz);
}

int main()
{
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  clock_ghz = prop.clockRate * 1e-6;
  printf("device %s, %d CUs, clock %.2f GHz, wave %d\n", prop.name, prop.multiProcessorCount, clock_ghz, prop.warpSize);
  uint64_t *d_out;
  CK(hipMalloc(&d_out, 1 << 22));
#define RUN(N) run_op(#N, k_##N, d_out)
  RUN(add_u32); RUN(mul_lo_u32); RUN(mul_hi_u32); RUN(mul_u32_u24); RUN(mul_hi_u32_u24); RUN(mad_u32_u24);
  RUN(mad_u64_u32); RUN(lshl_add_u64); RUN(cmp_lt_u64); RUN(cndmask); RUN(mov_dpp);
  RUN(bpermute);
  RUN(fma_f32); RUN(pk_fma_f32); RUN(fma_f64); RUN(mul_f64); RUN(add_f64); RUN(rndne_f64); RUN(floor_f64); RUN(min_f64);
  RUN(cmp_gt_f64); RUN(cvt_f64_u32); RUN(cvt_u32_f64);

  /* butterfly throughput */
  {
    const uint64_t q = 0x7fffffffe0001ULL;
    std::vector<uint64_t> wi(2, 1);
    auto cu = h_consts_u64(q, 1 << 14, wi);
    auto cf = h_consts_f64(q, 1 << 14, wi);
    const int reps = 256;
    int       blocks = 256 * 8;
    for(int wps : {1, 2, 4, 8}) {
      /* occupancy sweep: wps waves per SIMD (256-thread blocks, wps per CU-quarter) */
      blocks = 256 * wps;
      hipEvent_t a0, a1;
      CK(hipEventCreate(&a0));
      CK(hipEventCreate(&a1));
      float msv = 0;
      for(int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a0));
        hipLaunchKernelGGL((k_bfly<ArithF64, 0x5>), dim3(blocks), dim3(256), 0, 0, d_out, cf, h_tw_f64(123456789012345ULL, q), reps);
        CK(hipEventRecord(a1));
        CK(hipEventSynchronize(a1));
        CK(hipEventElapsedTime(&msv, a0, a1));
      }
      const double bf = (double)blocks * 256 * 32.0 * reps;
      printf("bfly F64 (reduce every 2nd) at %d waves/SIMD: %8.2f Gbutterfly/s -> %6.2f M NTT/s-equivalent\n", wps,
             bf / msv * 1e-6, bf / msv * 1e-6 * 1e3 / 114688.0);
    }
    blocks = 256 * 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto report = [&](const char *name, float ms) {
      const double bf = (double)blocks * 256 * 32.0 * reps;
      printf("%-28s %8.3f ms  %8.2f Gbutterfly/s  -> %6.2f M NTT/s at N=2^14 (VALU-only bound)\n", name, ms,
             bf / ms * 1e-6, bf / ms * 1e-6 * 1e3 / 114688.0);
    };
    float ms;
    for(int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_bfly<ArithU64, 0>), dim3(blocks), dim3(256), 0, 0, d_out, cu, h_tw_u64(123456789012345ULL, q), reps);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    report("bfly U64 (Harvey/Shoup)", ms);
    for(int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_bfly<ArithF64, 0xF>), dim3(blocks), dim3(256), 0, 0, d_out, cf, h_tw_f64(123456789012345ULL, q), reps);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    report("bfly F64 reduce every stage", ms);
    for(int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_bfly<ArithF64, 0x5>), dim3(blocks), dim3(256), 0, 0, d_out, cf, h_tw_f64(123456789012345ULL, q), reps);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    report("bfly F64 reduce every 2nd", ms);
    for(int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL((k_bfly<ArithF64, 0x0>), dim3(blocks), dim3(256), 0, 0, d_out, cf, h_tw_f64(123456789012345ULL, q), reps);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
    }
    report("bfly F64 no reduction", ms);
  }
  /* HBM copy */
  {
    const size_t bytes = (size_t)4 << 30;
    uint4 *      src, *dst;
    CK(hipMalloc(&src, bytes));
    CK(hipMalloc(&dst, bytes));
    CK(hipMemset(src, 1, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for(int blocks : {1024, 2048, 8192, 65536}) {
      float best = 1e9;
      for(int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, dst, src, bytes / 16);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("copy 4 GiB, %6d blocks: %.3f ms  %.2f TB/s (read+write)\n", blocks, best, 2.0 * bytes / best * 1e-9);
      best = 1e9;
      for(int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_copy8, dim3(blocks), dim3(256), 0, 0, (uint2 *)dst, (const uint2 *)src, bytes / 8);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("copy8 (8 B/lane) %6d blocks: %.3f ms  %.2f TB/s\n", blocks, best, 2.0 * bytes / best * 1e-9);
#define TRY(KN, LABEL)                                                                         \
      best = 1e9;                                                                              \
      for(int rep = 0; rep < 5; rep++) {                                                       \
        CK(hipEventRecord(e0));                                                                \
        hipLaunchKernelGGL(KN, dim3(blocks), dim3(256), 0, 0, dst, src, bytes / 16);           \
        CK(hipEventRecord(e1));                                                                \
        CK(hipEventSynchronize(e1));                                                           \
        float ms;                                                                              \
        CK(hipEventElapsedTime(&ms, e0, e1));                                                  \
        best = ms < best ? ms : best;                                                          \
      }                                                                                        \
      printf("%-28s %6d blocks: %.3f ms  %.2f TB/s\n", LABEL, blocks, best, 2.0 * bytes / best * 1e-9);
      TRY(k_copy_nt, "copy nt load + nt store")
      TRY(k_copy_ntst, "copy plain load + nt store")
      TRY(k_copy_x4, "copy 4x16B per thread")
      best = 1e9;
      for(int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_copy8ld_16st_strided, dim3(blocks), dim3(256), 0, 0, dst, (const uint2 *)src, bytes / 16);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("copy 4x8B-row loads + 2x16B strided stores %6d blocks: %.3f ms  %.2f TB/s\n", blocks, best, 2.0 * bytes / best * 1e-9);
#define TRYR(C, N, DST, LABEL)                                                                 \
      best = 1e9;                                                                              \
      for(int rep = 0; rep < 5; rep++) {                                                       \
        CK(hipEventRecord(e0));                                                                \
        hipLaunchKernelGGL((k_rows<C, N>), dim3(blocks), dim3(256), 0, 0, DST, (const uint2 *)src, bytes / 16); \
        CK(hipEventRecord(e1));                                                                \
        CK(hipEventSynchronize(e1));                                                           \
        float ms;                                                                              \
        CK(hipEventElapsedTime(&ms, e0, e1));                                                  \
        best = ms < best ? ms : best;                                                          \
      }                                                                                        \
      printf("%-44s %6d blocks: %.3f ms  %.2f TB/s\n", LABEL, blocks, best, 2.0 * bytes / best * 1e-9);
      TRYR(false, false, dst, "rows -> strided stores")
      TRYR(false, true, dst, "rows -> strided stores nt")
      TRYR(true, false, dst, "rows -> contiguous stores")
      TRYR(true, true, dst, "rows -> contiguous stores nt")
      best = 1e9;
      for(int rep = 0; rep < 5; rep++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rows_sector, dim3(blocks), dim3(256), 0, 0, src, (const uint2 *)src, bytes / 16);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
      }
      printf("%-44s %6d blocks: %.3f ms  %.2f TB/s\n", "in place: rows -> 32-byte-sector stores", blocks, best, 2.0 * bytes / best * 1e-9);
      TRYR(false, false, src, "in place: rows -> strided stores")
      TRYR(false, true, src, "in place: rows -> strided stores nt")
      TRYR(true, false, src, "in place: rows -> contiguous stores")
      TRYR(true, true, src, "in place: rows -> contiguous stores nt")
    }
  }
  return 0;
}
