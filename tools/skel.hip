/*
 * skel.hip -- data-movement skeletons of the 2^14-point fused kernel (diagnostic tool, GPU box only).
 *
 * Every variant streams 2^14-coefficient blocks (128 KiB) of a large buffer IN PLACE, like the NTT
 * kernel, with a configurable shape: workgroup size, coefficients per thread, load width, store
 * pattern, persistent loop with register prefetch or one block per workgroup, an optional cross-wave
 * LDS exchange (two s_barriers) and F dependent FP64 FMAs per coefficient standing in for the
 * butterflies (F = 72 issues as many VALU instructions per block as the real kernel, 1158 per wave).
 * It answers two questions the real kernel cannot answer by itself:
 *   1. what the memory system delivers for each access shape (the ceiling of the roofline fraction);
 *   2. what an ideally overlapped kernel (no barriers at all) with the same instruction count reaches,
 *      and at which shader clock (s_memtime / s_memrealtime) the chip runs it.
 * Build: make skel      Run: build/skel [GiB]
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                    \
  do {                                                           \
    hipError_t e_ = (x);                                         \
    if(e_ != hipSuccess) {                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));    \
      exit(1);                                                   \
    }                                                            \
  } while(0)

constexpr int LOGN = 14;
constexpr int NB   = 1 << LOGN; /* coefficients per block */

typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));

struct alignas(16) d2 {
  double a, b;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)(8u << LOGN), 0x00020000);
}
template <int AUX = 2>
__device__ __forceinline__ double ld8(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, AUX /* 2 = nt */);
  return __builtin_bit_cast(double, v);
}
template <int AUX = 0>
__device__ __forceinline__ void st8x(double x, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, x), r, (int)voff, (int)soff, AUX);
}
__device__ __forceinline__ d2 ld16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  const v4u32 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 2);
  return __builtin_bit_cast(d2, v);
}
__device__ __forceinline__ void st8(double x, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, x), r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ void st16(d2 x, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, x), r, (int)voff, (int)soff, 0);
}

/* store patterns (16 coefficients per 1024 threads; generalised to C per T) */
enum { ST_HALF = 0, ST_LINEAR = 1, ST_SPLIT = 2, ST_ROWS8 = 3 };
enum { MODE_ONESHOT = 0, MODE_PREFETCH = 1, MODE_LOOP = 2 };

template <int T, int C, int LW> __device__ __forceinline__ void load_block(double (&x)[C], __amdgpu_buffer_rsrc_t r, uint32_t t)
{
  if constexpr(LW == 8) {
#pragma unroll
    for(int e = 0; e < C; e++) x[e] = ld8(r, t * 8u, (uint32_t)e * T * 8u);
  } else {
#pragma unroll
    for(int h = 0; h < C / 2; h++) {
      const d2 v   = ld16(r, t * 16u, (uint32_t)h * T * 16u);
      x[2 * h]     = v.a;
      x[2 * h + 1] = v.b;
    }
  }
}

template <int T, int C, int ST> __device__ __forceinline__ void store_block(const double (&x)[C], __amdgpu_buffer_rsrc_t r, uint32_t t)
{
  if constexpr(ST == ST_ROWS8) {
#pragma unroll
    for(int e = 0; e < C; e++) st8(x[e], r, t * 8u, (uint32_t)e * T * 8u);
  } else if constexpr(ST == ST_LINEAR) {
#pragma unroll
    for(int h = 0; h < C / 2; h++) st16(d2{x[2 * h], x[2 * h + 1]}, r, t * 16u, (uint32_t)h * T * 16u);
  } else if constexpr(ST == ST_SPLIT) {
    /* what a v_permlane32_swap of the last group's quads gives: lanes 0-31 write the even 16-byte
     * chunks of a 1-KiB run, lanes 32-63 the odd ones */
    const uint32_t lane = t & 63u, wave = t >> 6;
    const uint32_t voff = wave * 1024u + (lane & 31u) * 32u + (lane >> 5) * 16u;
#pragma unroll
    for(int h = 0; h < C / 2; h++) st16(d2{x[2 * h], x[2 * h + 1]}, r, voff, (uint32_t)h * T * 16u);
  } else {
    /* the real kernel's last group: a thread owns runs of 4 consecutive coefficients */
#pragma unroll
    for(int h = 0; h < C / 2; h++)
      st16(d2{x[2 * h], x[2 * h + 1]}, r, t * 32u + (uint32_t)(h & 1) * 16u, (uint32_t)(h >> 1) * T * 32u);
  }
}

template <int C, int F0, int F1> __device__ __forceinline__ void fake_compute(double (&x)[C], double c1, double c2)
{
#pragma unroll
  for(int f = F0; f < F1; f++) {
#pragma unroll
    for(int e = 0; e < C; e++) x[e] = __builtin_fma(x[e], c1, c2);
  }
}

template <int T, int C> __device__ __forceinline__ void lds_exchange(double (&x)[C], double *lds, uint32_t t)
{
  __syncthreads();
#pragma unroll
  for(int e = 0; e < C; e++) lds[e * (T + 1) + t] = x[e];
  __syncthreads();
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = lds[e * (T + 1) + (t ^ (((uint32_t)e << 6) & (T - 1)))];
}

struct Clk {
  unsigned long long c0, c1, r0, r1;
};

/* T threads, C coefficients per thread (T*C = 2^14), LW load width, ST store pattern, MODE, F fmas per
 * coefficient, X cross-wave exchanges per block, LDSB extra LDS bytes (limits workgroups per CU), WPS */
template <int T, int C, int LW, int ST, int MODE, int F, int X, int LDSB, int WPS>
__global__ void __launch_bounds__(T, WPS) k_skel(double *a, uint64_t nblocks, double c1, double c2, Clk *clk)
{
  static_assert(T * C == NB, "block shape");
  constexpr int LDS_ELEMS = (X ? C * (T + 1) : 0) + LDSB / 8 + 1;
  __shared__ double lds[LDS_ELEMS];
  const uint32_t t = threadIdx.x;
  if(LDSB && t == 0) lds[LDS_ELEMS - 1] = c1; /* keep the allocation */
  unsigned long long c0 = 0, r0 = 0;
  if(t == 0) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  constexpr int FA = X ? F / 4 : F; /* work in front of the exchange (the real kernel's first group) */
  if constexpr(MODE == MODE_PREFETCH) {
    uint64_t b = blockIdx.x;
    if(b >= nblocks) return;
    double raw[C];
    load_block<T, C, LW>(raw, rsrc_of(a + (b << LOGN)), t);
    for(; b < nblocks; b += gridDim.x) {
      double x[C];
#pragma unroll
      for(int e = 0; e < C; e++) x[e] = raw[e] + c2;
      const uint64_t nb = b + gridDim.x < nblocks ? b + gridDim.x : b;
      load_block<T, C, LW>(raw, rsrc_of(a + (nb << LOGN)), t);
      fake_compute<C, 0, FA>(x, c1, c2);
      if constexpr(X) {
        lds_exchange<T, C>(x, lds, t);
        fake_compute<C, FA, F>(x, c1, c2);
      }
      store_block<T, C, ST>(x, rsrc_of(a + (b << LOGN)), t);
    }
  } else {
    for(uint64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
      double x[C];
      load_block<T, C, LW>(x, rsrc_of(a + (b << LOGN)), t);
      fake_compute<C, 0, FA>(x, c1, c2);
      if constexpr(X) {
        lds_exchange<T, C>(x, lds, t);
        fake_compute<C, FA, F>(x, c1, c2);
      }
      store_block<T, C, ST>(x, rsrc_of(a + (b << LOGN)), t);
      if(MODE == MODE_ONESHOT) break;
    }
  }
  if(t == 0 && blockIdx.x < 4096) {
    Clk k;
    k.c0             = c0;
    k.r0             = r0;
    k.c1             = __builtin_amdgcn_s_memtime();
    k.r1             = __builtin_amdgcn_s_memrealtime();
    clk[blockIdx.x]  = k;
  }
}

/* Two-pass transforms (N = 2^16, 2^17): ONE workgroup owns a whole polynomial of PB blocks and runs both
 * passes back to back -- pass 1 in the column shape (thread t touches element t of every block: PB
 * coalesced 8-byte accesses 128 KiB apart, several rounds), a workgroup barrier, pass 2 over the PB
 * blocks like the fused kernel.  What pass 1 wrote is read again by the same CU a few microseconds
 * later: this measures how much of that second read the L2 / Infinity Cache absorbs.  LA1/SA1/LA2/SA2:
 * cache-policy bits (0 plain, 2 nt) of pass-1 loads/stores and pass-2 loads/stores. */
template <int PB, int LA1, int SA1, int LA2, int SA2, int F>
__global__ void __launch_bounds__(1024, 4) k_twophase(double *a, uint64_t npoly, double c1, double c2)
{
  constexpr int T = 1024, C = 16;
  const uint32_t t = threadIdx.x;
  for(uint64_t p = blockIdx.x; p < npoly; p += gridDim.x) {
    double *poly = a + p * (uint64_t)PB * NB;
    /* pass 1: C/PB... each round handles rows {r} of all PB blocks: 16 values per thread per round */
    constexpr int RPB = C / PB; /* rows of one block per round */
    for(int round = 0; round < C * PB / C; round++) {
      double x[C];
#pragma unroll
      for(int b = 0; b < PB; b++) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(poly + (uint64_t)b * NB);
#pragma unroll
        for(int k = 0; k < RPB; k++) x[b * RPB + k] = ld8<LA1>(r, t * 8u, (uint32_t)(round * RPB + k) * T * 8u);
      }
      fake_compute<C, 0, F / 4>(x, c1, c2);
#pragma unroll
      for(int b = 0; b < PB; b++) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(poly + (uint64_t)b * NB);
#pragma unroll
        for(int k = 0; k < RPB; k++) st8x<SA1>(x[b * RPB + k], r, t * 8u, (uint32_t)(round * RPB + k) * T * 8u);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    /* pass 2 */
    for(int b = 0; b < PB; b++) {
      const __amdgpu_buffer_rsrc_t r = rsrc_of(poly + (uint64_t)b * NB);
      double x[C];
#pragma unroll
      for(int e = 0; e < C; e++) x[e] = ld8<LA2>(r, t * 8u, (uint32_t)e * T * 8u);
      fake_compute<C, 0, F>(x, c1, c2);
#pragma unroll
      for(int h = 0; h < C / 2; h++) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, d2{x[2 * h], x[2 * h + 1]}), r, (int)(t * 16u),
                                               (int)((uint32_t)h * T * 16u), SA2);
      }
    }
    __syncthreads();
  }
}

/* residency probe: the same CHUNK bytes read-modified-written REP times inside one launch */
__global__ void __launch_bounds__(256) k_rmw_rep(double *a, size_t n2, int rep, double c2)
{
  d2 *p = reinterpret_cast<d2 *>(a);
  for(int r = 0; r < rep; r++) {
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
      d2 v = p[i];
      v.a += c2;
      v.b += c2;
      p[i] = v;
    }
  }
}

/* reference points: grid-stride in-place read-modify-write and a plain copy, 256-thread workgroups */
__global__ void __launch_bounds__(256) k_rmw16(double *a, size_t n2, double c2)
{
  d2 *p = reinterpret_cast<d2 *>(a);
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    d2 v = p[i];
    v.a += c2;
    v.b += c2;
    p[i] = v;
  }
}
__global__ void __launch_bounds__(256) k_copy16(double *dst, const double *src, size_t n2)
{
  const d2 *s = reinterpret_cast<const d2 *>(src);
  d2 *      d = reinterpret_cast<d2 *>(dst);
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ void __launch_bounds__(256) k_fill(double *a, size_t n)
{
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = i * 0x9e3779b97f4a7c15ULL + 0x1234567ULL;
    z ^= z >> 31;
    z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 29;
    a[i] = 1.0 + (double)(z >> 12) * 0x1p-52; /* [1,2), random mantissa */
  }
}

static double *   g_buf;
static size_t     g_n;
static uint64_t   g_nblocks;
static Clk *      g_clk;
static hipEvent_t g_e0, g_e1;
static int        g_reps = 24;

template <class L> static void time_it(const char *label, L launch, double bytes, bool has_clk)
{
  std::vector<float> ms;
  CK(hipMemset(g_clk, 0, 4096 * sizeof(Clk)));
  for(int r = 0; r < g_reps; r++) {
    CK(hipEventRecord(g_e0));
    launch();
    CK(hipEventRecord(g_e1));
    CK(hipEventSynchronize(g_e1));
    float m;
    CK(hipEventElapsedTime(&m, g_e0, g_e1));
    ms.push_back(m);
  }
  CK(hipGetLastError());
  std::vector<float> tail(ms.begin() + g_reps / 2, ms.end());
  std::sort(tail.begin(), tail.end());
  const float med = tail[tail.size() / 2], best = tail[0];
  double      ghz = 0;
  if(has_clk) {
    std::vector<Clk> h(256);
    CK(hipMemcpy(h.data(), g_clk, 256 * sizeof(Clk), hipMemcpyDeviceToHost));
    std::vector<double> f;
    for(const Clk &k : h)
      if(k.r1 > k.r0) f.push_back((double)(k.c1 - k.c0) / (double)(k.r1 - k.r0) * 0.1);
    if(!f.empty()) {
      std::sort(f.begin(), f.end());
      ghz = f[f.size() / 2];
    }
  }
  printf("%-58s med %7.3f ms  best %7.3f  %5.2f TB/s  frac %.3f  %6.2f M blk/s  clk %.2f GHz\n", label, med, best,
         bytes / med * 1e-9, bytes / med * 1e-9 / 8.0, g_nblocks / med * 1e-3, ghz);
  fflush(stdout);
}

template <int T, int C, int LW, int ST, int MODE, int F, int X, int LDSB, int WPS> static void run(const char *label, int wg_per_cu)
{
  const unsigned grid = MODE == MODE_ONESHOT ? (unsigned)g_nblocks : (unsigned)(256 * wg_per_cu);
  char           full[160];
  snprintf(full, sizeof full, "T%-4d C%-2d ld%-2d st%d mode%d F%-2d X%d lds%-3dK wg/cu %d  %s", T, C, LW, ST, MODE, F, X, LDSB >> 10,
           wg_per_cu, label);
  time_it(
    full,
    [&] {
      hipLaunchKernelGGL((k_skel<T, C, LW, ST, MODE, F, X, LDSB, WPS>), dim3(grid), dim3(T), 0, 0, g_buf, g_nblocks, 0.999999, 1e-9,
                         g_clk);
    },
    (double)g_nblocks * NB * 16.0, true);
}

int main(int argc, char **argv)
{
  const double gib = argc > 1 ? atof(argv[1]) : 16.0;
  if(argc > 2) g_reps = atoi(argv[2]);
  g_nblocks = (uint64_t)(gib * 1024.0 * 1024.0 * 1024.0 / (NB * 8.0));
  g_n       = g_nblocks * NB;
  CK(hipMalloc(&g_buf, g_n * 8));
  CK(hipMalloc(&g_clk, 4096 * sizeof(Clk)));
  CK(hipEventCreate(&g_e0));
  CK(hipEventCreate(&g_e1));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, g_n);
  CK(hipDeviceSynchronize());
  printf("# %.1f GiB in place, %llu blocks of 2^14 coefficients, %d launches per row (median of the second half)\n", gib,
         (unsigned long long)g_nblocks, g_reps);

  /* reference points */
  for(int grid : {2048, 8192, 65536}) {
    char l[96];
    snprintf(l, sizeof l, "in-place rmw, 16 B/lane, 256-thread WGs, grid %d", grid);
    time_it(l, [&] { hipLaunchKernelGGL(k_rmw16, dim3(grid), dim3(256), 0, 0, g_buf, g_n / 2, 1e-9); }, g_n * 16.0, false);
  }
  {
    double *half = g_buf + g_n / 2;
    for(int grid : {2048, 65536}) {
      char l[96];
      snprintf(l, sizeof l, "copy half->half, 16 B/lane, grid %d", grid);
      time_it(l, [&] { hipLaunchKernelGGL(k_copy16, dim3(grid), dim3(256), 0, 0, half, g_buf, g_n / 4); }, g_n * 8.0, false);
    }
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, g_n);
    CK(hipDeviceSynchronize());
  }

  /*   T     C  LW  ST         MODE           F   X  LDSB       WPS */
  puts("# the real kernel's shape: 1024 threads x 16, 8-byte row loads, half-line stores, persistent + register prefetch");
  run<1024, 16, 8, ST_HALF, MODE_PREFETCH, 0, 0, 0, 4>("memory only", 1);
  run<1024, 16, 8, ST_HALF, MODE_PREFETCH, 0, 1, 0, 4>("+ exchange", 1);
  run<1024, 16, 8, ST_HALF, MODE_PREFETCH, 24, 0, 0, 4>("", 1);
  run<1024, 16, 8, ST_HALF, MODE_PREFETCH, 48, 0, 0, 4>("", 1);
  run<1024, 16, 8, ST_HALF, MODE_PREFETCH, 60, 0, 0, 4>("", 1);
  run<1024, 16, 8, ST_HALF, MODE_PREFETCH, 72, 0, 0, 4>("VALU count of the real kernel, no barrier", 1);
  run<1024, 16, 8, ST_HALF, MODE_PREFETCH, 72, 1, 0, 4>("the same + exchange", 1);
  run<1024, 16, 8, ST_HALF, MODE_PREFETCH, 96, 0, 0, 4>("", 1);
  puts("# store patterns");
  run<1024, 16, 8, ST_LINEAR, MODE_PREFETCH, 0, 0, 0, 4>("whole-line stores", 1);
  run<1024, 16, 8, ST_SPLIT, MODE_PREFETCH, 0, 0, 0, 4>("split-half whole-line stores", 1);
  run<1024, 16, 8, ST_ROWS8, MODE_PREFETCH, 0, 0, 0, 4>("8-byte row stores (inverse)", 1);
  run<1024, 16, 8, ST_LINEAR, MODE_PREFETCH, 72, 0, 0, 4>("whole-line stores", 1);
  run<1024, 16, 8, ST_SPLIT, MODE_PREFETCH, 72, 0, 0, 4>("split-half whole-line stores", 1);
  run<1024, 16, 8, ST_LINEAR, MODE_PREFETCH, 72, 1, 0, 4>("whole-line stores + exchange", 1);
  puts("# 16-byte loads");
  run<1024, 16, 16, ST_LINEAR, MODE_PREFETCH, 0, 0, 0, 4>("", 1);
  run<1024, 16, 16, ST_HALF, MODE_PREFETCH, 0, 0, 0, 4>("", 1);
  run<1024, 16, 16, ST_LINEAR, MODE_PREFETCH, 72, 0, 0, 4>("", 1);
  puts("# no register prefetch: one block per workgroup (grid = blocks) or a plain persistent loop");
  run<1024, 16, 8, ST_HALF, MODE_ONESHOT, 0, 0, 0, 4>("2 WGs per CU by registers", 1);
  run<1024, 16, 8, ST_HALF, MODE_ONESHOT, 72, 0, 0, 4>("", 1);
  run<1024, 16, 8, ST_HALF, MODE_ONESHOT, 72, 1, 0, 4>("1 WG per CU by LDS", 1);
  run<1024, 16, 8, ST_LINEAR, MODE_ONESHOT, 0, 0, 0, 4>("", 1);
  run<1024, 16, 8, ST_HALF, MODE_LOOP, 0, 0, 0, 4>("", 2);
  run<1024, 16, 8, ST_HALF, MODE_LOOP, 72, 0, 0, 4>("", 2);
  puts("# 512 threads x 32 coefficients, two workgroups per CU");
  run<512, 32, 8, ST_HALF, MODE_LOOP, 0, 0, 0, 4>("", 2);
  run<512, 32, 8, ST_HALF, MODE_LOOP, 72, 0, 0, 4>("", 2);
  run<512, 32, 8, ST_LINEAR, MODE_LOOP, 72, 0, 0, 4>("", 2);
  run<512, 32, 8, ST_HALF, MODE_ONESHOT, 72, 0, 0, 4>("", 2);
  run<512, 32, 8, ST_HALF, MODE_PREFETCH, 72, 0, 0, 2>("one WG per CU, 256 VGPRs", 1);
  run<512, 32, 8, ST_LINEAR, MODE_PREFETCH, 72, 0, 0, 2>("one WG per CU, 256 VGPRs", 1);
  puts("# 256 threads x 64 coefficients, two/four workgroups per CU");
  run<256, 64, 8, ST_HALF, MODE_LOOP, 72, 0, 0, 2>("", 2);
  run<256, 64, 8, ST_LINEAR, MODE_LOOP, 72, 0, 0, 2>("", 2);
  run<256, 64, 8, ST_LINEAR, MODE_LOOP, 0, 0, 0, 2>("", 2);
  puts("# residency: one chunk read-modified-written 8 times in one launch (each thread re-touches its own lines)");
  for(int mib : {32, 64, 96, 128, 160, 192, 256, 384, 1024}) {
    char l[96];
    const size_t n2 = (size_t)mib * 1024 * 1024 / 16;
    snprintf(l, sizeof l, "rmw x8 over %4d MiB, grid 2048", mib);
    time_it(l, [&] { hipLaunchKernelGGL(k_rmw_rep, dim3(2048), dim3(256), 0, 0, g_buf, n2, 8, 1e-9); }, (double)g_nblocks * NB * 16.0 * ((double)mib * 8 / (gib * 1024)), false);
  }
  puts("# two passes per polynomial inside one workgroup (bytes counted ONCE: 16*N per polynomial, like the roofline)");
#define TWO(PB, LA1, SA1, LA2, SA2, F)                                                                                   \
  {                                                                                                                      \
    char l[128];                                                                                                         \
    snprintf(l, sizeof l, "two-phase PB%d ld1 %d st1 %d ld2 %d st2 %d F%d", PB, LA1, SA1, LA2, SA2, F);                  \
    time_it(l, [&] { hipLaunchKernelGGL((k_twophase<PB, LA1, SA1, LA2, SA2, F>), dim3(256), dim3(1024), 0, 0, g_buf, g_nblocks / PB, 0.999999, 1e-9); }, \
            (double)g_nblocks * NB * 16.0, false);                                                                       \
  }
  TWO(4, 2, 0, 0, 0, 0)
  TWO(4, 2, 0, 0, 2, 0)
  TWO(4, 0, 0, 0, 0, 0)
  TWO(4, 2, 0, 2, 0, 0)
  TWO(4, 2, 0, 0, 0, 72)
  TWO(8, 2, 0, 0, 0, 0)
  TWO(8, 2, 0, 0, 2, 0)
  TWO(8, 2, 0, 0, 0, 72)
  TWO(2, 2, 0, 0, 0, 0)
  TWO(2, 2, 0, 0, 0, 72)
  return 0;
}
