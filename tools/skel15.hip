/*
 * skel15.hip -- skeleton of a SINGLE-PASS 2^15-point transform (review r05, item 5; diagnostic tool, GPU box only).
 *
 * The shipped 2^15 transform is two passes (3 column stages + 2^12-point blocks, team_kernel): 24*N bytes cross the fabric, 0.43 of
 * the 16*N roofline forward / 0.38 inverse.  The alternative: ONE 1024-thread workgroup holds a whole polynomial in registers --
 * 32 words per thread, 256 KiB -- so that every coefficient crosses HBM exactly twice (16*N).  The forward data flow fixes the
 * schedule: stage 1 pairs x[i] with x[i + 2^14] (in-thread when a thread owns both), after which the two halves are independent
 * 2^14-point transforms of the shape the 2^14 kernel already runs (16 words per thread, groups of 4+4+4+2 stages, three LDS
 * exchanges through a 128 KiB buffer, one of them across waves).  Registers: 64 VGPRs hold the polynomial, so the NEXT polynomial
 * cannot be prefetched whole (128 VGPRs per lane at 16 waves per CU): its first half is requested when half A has been stored,
 * its second half when half B has been stored -- and stage 1 of the next polynomial needs both, so the second half's latency is
 * exposed once per polynomial (nothing else is resident on the CU to cover it).
 *
 * Variants (template PF): 0 = no prefetch (both halves loaded at the top); 1 = the schedule above; 2 = first half as above, second
 * half requested BEFORE half B's last group + stores (its 16 registers are taken from a third set: 48 data words = 96 VGPRs, which only a
 * memory-only skeleton affords -- the upper bound of any cleverer register plan).
 * F = dependent FP64 FMAs per coefficient and half standing in for the butterflies (72 = the 2^14 kernel's VALU count per
 * block: tools/skel.hip), +4 per coefficient for stage 1.  X = 1: the three exchanges per half (two wave-local, one across waves).
 * Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -o build/skel15 tools/skel15.hip     Run: build/skel15 [GiB] [reps]
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                    \
  do {                                                           \
    hipError_t e_ = (x);                                         \
    if(e_ != hipSuccess) {                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));    \
      exit(1);                                                   \
    }                                                            \
  } while(0)

constexpr int LOGN = 15, NP = 1 << LOGN, T = 1024, H = 16; /* H words per thread and half */

typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
struct alignas(16) d2 {
  double a, b;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, bool live = true)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, live ? (int)(8u << LOGN) : 0, 0x00020000);
}
/* half h (0 / 1) of a polynomial in the first-kind layout: slot e <-> index h * 2^14 + e * 1024 + t (coalesced 8-byte rows) */
__device__ __forceinline__ void load_half(double (&x)[H], __amdgpu_buffer_rsrc_t r, uint32_t t, int h)
{
#pragma unroll
  for(int e = 0; e < H; e++) {
    const v2u32 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)(t * 8u), (int)(((uint32_t)h << 17) + (uint32_t)e * T * 8u), 2 /* nt */);
    x[e]          = __builtin_bit_cast(double, v);
  }
}
/* half h stored as whole lines: the last group's runs of four coefficients after the permlane32 swap of the 2^14 kernel
 * (lanes 0-31 the even 16-byte chunks of a 1-KiB run, lanes 32-63 the odd ones): tools/skel.hip ST_SPLIT */
__device__ __forceinline__ void store_half(const double (&x)[H], __amdgpu_buffer_rsrc_t r, uint32_t t, int h)
{
  const uint32_t lane = t & 63u, wave = t >> 6;
  const uint32_t voff = wave * 1024u + (lane & 31u) * 32u + (lane >> 5) * 16u;
#pragma unroll
  for(int k = 0; k < H / 2; k++) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, d2{x[2 * k], x[2 * k + 1]}), r, (int)voff,
                                           (int)(((uint32_t)h << 17) + (uint32_t)k * T * 16u), 0);
  }
}
template <int F0, int F1> __device__ __forceinline__ void fake(double (&x)[H], double c1, double c2)
{
#pragma unroll
  for(int f = F0; f < F1; f++) {
#pragma unroll
    for(int e = 0; e < H; e++) x[e] = __builtin_fma(x[e], c1, c2);
  }
}
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
/* the 2^14 kernel's exchanges: reader-linear layout slot * (T + 1) + thread; CROSS = across waves (two s_barriers), else
 * inside a wave (lanes permuted, no barrier) */
template <bool CROSS> __device__ __forceinline__ void exchange(double (&x)[H], double *lds, uint32_t t)
{
  if constexpr(CROSS) {
    __syncthreads();
#pragma unroll
    for(int e = 0; e < H; e++) lds[e * (T + 1) + t] = x[e];
    __syncthreads();
#pragma unroll
    for(int e = 0; e < H; e++) x[e] = lds[e * (T + 1) + (t ^ (((uint32_t)e << 6) & (T - 1)))];
  } else {
#pragma unroll
    for(int e = 0; e < H; e++) lds[e * (T + 1) + t] = x[e];
    wave_sync();
#pragma unroll
    for(int e = 0; e < H; e++) x[e] = lds[e * (T + 1) + (t ^ (uint32_t)(e * 3 & 63))];
    wave_sync();
  }
}
/* one half through its fourteen stages: groups of 4 + 4 + 4 + 2 stages = F/14 * {4,4,4,2} FMAs, exchanges between */
template <int F, int X> __device__ __forceinline__ void half_transform(double (&x)[H], double *lds, uint32_t t, double c1, double c2)
{
  constexpr int G1 = F * 4 / 14, G2 = F * 8 / 14, G3 = F * 12 / 14;
  fake<0, G1>(x, c1, c2);
  if constexpr(X) exchange<false>(x, lds, t);
  fake<G1, G2>(x, c1, c2);
  if constexpr(X) exchange<true>(x, lds, t);
  fake<G2, G3>(x, c1, c2);
  if constexpr(X) exchange<false>(x, lds, t);
  fake<G3, F>(x, c1, c2);
}

template <int PF, int F, int X> __global__ void __launch_bounds__(1024, 4) k_skel15(double *a, uint64_t npoly, double c1, double c2)
{
  constexpr int LDS_ELEMS = X ? H * (T + 1) : 1;
  __shared__ double lds[LDS_ELEMS];
  const uint32_t t = threadIdx.x;
  uint64_t       p = blockIdx.x;
  if(p >= npoly) return;
  double ra[H], rb[H];
  if constexpr(PF != 0) {
    load_half(ra, rsrc_of(a + (p << LOGN)), t, 0);
    load_half(rb, rsrc_of(a + (p << LOGN)), t, 1);
  }
  for(; p < npoly; p += gridDim.x) {
    const __amdgpu_buffer_rsrc_t r    = rsrc_of(a + (p << LOGN));
    const bool                   more = p + gridDim.x < npoly;
    const __amdgpu_buffer_rsrc_t rn   = rsrc_of(a + ((more ? p + gridDim.x : p) << LOGN), more);
    double xa[H], xb[H];
    if constexpr(PF == 0) {
      load_half(xa, r, t, 0);
      load_half(xb, r, t, 1);
    } else {
#pragma unroll
      for(int e = 0; e < H; e++) xa[e] = ra[e], xb[e] = rb[e];
    }
    /* stage 1: x[i] +- w x[i + 2^14], in-thread (4 FMAs per coefficient of each half) */
    if constexpr(F > 0) {
#pragma unroll
      for(int e = 0; e < H; e++) {
        const double m = __builtin_fma(xb[e], c1, c2);
        const double k = __builtin_fma(m, c1, xb[e]);
        const double d = __builtin_fma(k, c2, m);
        xb[e]          = __builtin_fma(xa[e], c1, -d);
        xa[e]          = __builtin_fma(xa[e], c1, d);
      }
    }
    if constexpr(PF == 2) load_half(rb, rn, t, 1); /* a third register set: the memory-only upper bound */
    half_transform<F, X>(xa, lds, t, c1, c2);
    store_half(xa, r, t, 0);
    if constexpr(PF != 0) load_half(ra, rn, t, 0); /* half A's registers are free: the next polynomial's first half */
    half_transform<F, X>(xb, lds, t, c1, c2);
    store_half(xb, r, t, 1);
    if constexpr(PF == 1) load_half(rb, rn, t, 1); /* half B's registers are free: the second half -- waited for at the top */
  }
}

/* reference: the two-pass shape at the same size is not modelled here (tools/skel4.hip, skel5.hip); this one: a plain in-place rmw */
__global__ void __launch_bounds__(256) k_rmw16(double *a, size_t n2, double c2)
{
  d2 *q = reinterpret_cast<d2 *>(a);
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) {
    d2 v = q[i];
    v.a += c2;
    v.b += c2;
    q[i] = v;
  }
}
__global__ void __launch_bounds__(256) k_fill(double *a, size_t n)
{
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long z = i * 0x9e3779b97f4a7c15ULL + 0x1234567ULL;
    z ^= z >> 31;
    z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 29;
    a[i] = 1.0 + (double)(z >> 12) * 0x1p-52;
  }
}

static double *   g_buf;
static uint64_t   g_npoly;
static hipEvent_t g_e0, g_e1;
static int        g_reps = 16;

template <class L> static void time_it(const char *label, L launch)
{
  std::vector<float> ms;
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
  const float  med   = tail[tail.size() / 2], best = tail[0];
  const double bytes = (double)g_npoly * NP * 16.0;
  printf("%-86s med %7.3f ms  best %7.3f  %5.2f TB/s  frac of 8 TB/s at 16N %.3f (best %.3f)  %6.3f M poly/s\n", label, med, best, bytes / med * 1e-9,
         bytes / med * 1e-9 / 8.0, bytes / best * 1e-9 / 8.0, g_npoly / med * 1e-3);
  fflush(stdout);
}
template <int PF, int F, int X> static void run(const char *label)
{
  char full[200];
  snprintf(full, sizeof full, "single-pass 2^15: prefetch %d  F%-2d  exchanges %d  %s", PF, F, X, label);
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(&k_skel15<PF, F, X>)));
  char full2[260];
  snprintf(full2, sizeof full2, "%s [%d VGPRs, %zu B scratch]", full, fa.numRegs, (size_t)fa.localSizeBytes);
  time_it(full2, [&] { hipLaunchKernelGGL((k_skel15<PF, F, X>), dim3(256), dim3(1024), 0, 0, g_buf, g_npoly, 0.999999, 1e-9); });
}

int main(int argc, char **argv)
{
  const double gib = argc > 1 ? atof(argv[1]) : 16.0;
  if(argc > 2) g_reps = atoi(argv[2]);
  g_npoly        = (uint64_t)(gib * 1024.0 * 1024.0 * 1024.0 / (NP * 8.0));
  const size_t n = g_npoly * NP;
  CK(hipMalloc(&g_buf, n * 8));
  CK(hipEventCreate(&g_e0));
  CK(hipEventCreate(&g_e1));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, n);
  CK(hipDeviceSynchronize());
  printf("# %.1f GiB in place, %llu polynomials of 2^15 coefficients, %d launches per row (median / best of the second half)\n", gib,
         (unsigned long long)g_npoly, g_reps);
  printf("# shipped two-pass transform at 2^15 (profiles/r05/sweep_sizes.txt): forward 0.43, inverse 0.38; the bar for building the kernel: > 0.47 with the FMA count\n");
  time_it("in-place rmw, 16 B/lane, 256-thread WGs, grid 8192", [&] { hipLaunchKernelGGL(k_rmw16, dim3(8192), dim3(256), 0, 0, g_buf, n / 2, 1e-9); });
  puts("# memory only");
  run<0, 0, 0>("no prefetch");
  run<1, 0, 0>("half prefetch (the schedule a kernel could run)");
  run<2, 0, 0>("second half early (third register set: upper bound)");
  puts("# + the LDS exchanges (three per half, one across waves)");
  run<1, 0, 1>("");
  run<2, 0, 1>("");
  puts("# + the VALU count of the butterflies");
  run<0, 72, 1>("no prefetch");
  run<1, 24, 1>("");
  run<1, 48, 1>("");
  run<1, 72, 1>("the 2^14 kernel's VALU count per coefficient and half + stage 1");
  run<2, 72, 1>("(third register set)");
  run<1, 72, 0>("no exchanges");
  run<1, 96, 1>("");
  return 0;
}
