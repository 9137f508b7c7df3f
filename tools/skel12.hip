/*
 * skel12.hip -- data-movement skeletons of the 2^12-point block kernel next to the 2^14-point one (diagnostic tool, GPU box only).
 *
 * tools/skel.hip answered "what bounds the 2^14 kernel" (profiles/r02/skeleton.txt).  The 2^12 block kernel -- BASELINE config 2,
 * and the row item of every transform of 2^15 points and more -- issues 12 % fewer VALU instructions per coefficient than the
 * 2^14 kernel and still runs 3 % below it (profiles/r04/NOTES.txt).  This tool rebuilds BOTH kernels' shapes from the same
 * template, ingredient by ingredient, on the same box in the same run:
 *   shape     LOGN = 12: 256 threads x 16 coefficients, 32 KiB blocks, four persistent workgroups per CU
 *             LOGN = 14: 1024 threads x 16, 128 KiB blocks, one persistent workgroup per CU
 *   loads     8-byte row loads (row e of a thread = index e * T + t), the next block prefetched into registers
 *   stores    half-line (a lane owns runs of four coefficients: the kernels' last group) or whole-line
 *   VALU      F dependent FP64 FMAs per coefficient, split over the stage groups like the kernel's butterflies
 *   exchange  XB cross-wave LDS exchanges (two s_barriers each), XW wave-local ones (no barrier), between the groups
 * Build: make skel12      Run: build/skel12 [GiB] [launches per row]
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

typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
struct alignas(16) d2 {
  double a, b;
};
constexpr int C = 16; /* coefficients per thread */

template <int LOGN> __device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, bool live = true)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, live ? (int)(8u << LOGN) : 0, 0x00020000);
}
__device__ __forceinline__ double ld8(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 2 /* nt */));
}
__device__ __forceinline__ void st16(d2 x, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, x), r, (int)voff, (int)soff, 0);
}

enum { ST_HALF = 0, ST_SPLIT = 2 };

template <int T> __device__ __forceinline__ void load_block(double (&x)[C], __amdgpu_buffer_rsrc_t r, uint32_t t)
{
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = ld8(r, t * 8u, (uint32_t)e * T * 8u);
}
template <int T, int ST> __device__ __forceinline__ void store_block(const double (&x)[C], __amdgpu_buffer_rsrc_t r, uint32_t t)
{
  if constexpr(ST == ST_SPLIT) {
    /* what v_permlane32_swap of the last group's quads gives: every store instruction of a wave covers one contiguous KiB */
    const uint32_t lane = t & 63u, wave = t >> 6;
    const uint32_t voff = wave * 1024u + (lane & 31u) * 32u + (lane >> 5) * 16u;
#pragma unroll
    for(int h = 0; h < C / 2; h++) st16(d2{x[2 * h], x[2 * h + 1]}, r, voff, (uint32_t)h * T * 16u);
  } else {
    /* the kernels' last group: a thread owns runs of 4 consecutive coefficients */
#pragma unroll
    for(int h = 0; h < C / 2; h++) st16(d2{x[2 * h], x[2 * h + 1]}, r, t * 32u + (uint32_t)(h & 1) * 16u, (uint32_t)(h >> 1) * T * 32u);
  }
}
template <int F> __device__ __forceinline__ void fake_compute(double (&x)[C], double c1, double c2)
{
#pragma unroll
  for(int f = 0; f < F; f++) {
#pragma unroll
    for(int e = 0; e < C; e++) x[e] = __builtin_fma(x[e], c1, c2);
  }
}
/* cross-wave exchange: two workgroup barriers, scattered ds_write_b64, linear ds_read_b64 (row pitch T + 1 as the kernels) */
template <int T> __device__ __forceinline__ void exchange_cross(double (&x)[C], double *lds, uint32_t t)
{
  __syncthreads();
#pragma unroll
  for(int e = 0; e < C; e++) lds[e * (T + 1) + (t ^ (((uint32_t)e << 6) & (T - 1)))] = x[e];
  __syncthreads();
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = lds[e * (T + 1) + t];
}
/* wave-local exchange: the lanes of one wave permute among themselves, no barrier */
template <int T> __device__ __forceinline__ void exchange_wave(double (&x)[C], double *lds, uint32_t t)
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for(int e = 0; e < C; e++) lds[e * (T + 1) + (t ^ ((uint32_t)e * 4u & 63u))] = x[e];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = lds[e * (T + 1) + t];
}

struct Clk {
  unsigned long long c0, c1, r0, r1;
};

/* F FMAs per coefficient split like the butterflies of the stage groups: 2^12 = 2 + 4 + 4 + 2 stages, 2^14 = 4 + 4 + 4 + 2 */
template <int LOGN, int F, int G> constexpr int fshare()
{
  constexpr int st[4] = {LOGN == 12 ? 2 : 4, 4, 4, 2};
  int           before = 0;
  for(int g = 0; g < G; g++) before += st[g];
  return (F * (before + st[G])) / LOGN - (F * before) / LOGN;
}

template <int LOGN, int ST, int F, int XB, int XW, int LDSB, int WPS>
__global__ void __launch_bounds__((1 << (LOGN - 4)), WPS) k_skel(double *a, uint64_t nblocks, double c1, double c2, Clk *clk)
{
  constexpr int T         = 1 << (LOGN - 4);
  constexpr int LDS_ELEMS = ((XB || XW) ? C * (T + 1) : 0) + LDSB / 8 + 1;
  __shared__ double lds[LDS_ELEMS];
  const uint32_t t = threadIdx.x;
  if(LDSB && t == 0) lds[LDS_ELEMS - 1] = c1; /* keep the allocation */
  unsigned long long c0 = 0, r0 = 0;
  if(t == 0) {
    c0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  uint64_t b = blockIdx.x;
  if(b >= nblocks) return;
  double raw[C];
  load_block<T>(raw, rsrc_of<LOGN>(a + (b << LOGN)), t);
  for(; b < nblocks; b += gridDim.x) {
    double x[C];
#pragma unroll
    for(int e = 0; e < C; e++) x[e] = raw[e] + c2;
    const bool     more = b + gridDim.x < nblocks;
    const uint64_t nb   = more ? b + gridDim.x : b;
    load_block<T>(raw, rsrc_of<LOGN>(a + (nb << LOGN), more), t);
    fake_compute<fshare<LOGN, F, 0>()>(x, c1, c2);
    if constexpr(XB) exchange_cross<T>(x, lds, t);
    fake_compute<fshare<LOGN, F, 1>()>(x, c1, c2);
    if constexpr(XW >= 1) exchange_wave<T>(x, lds, t);
    fake_compute<fshare<LOGN, F, 2>()>(x, c1, c2);
    if constexpr(XW >= 2) exchange_wave<T>(x, lds, t);
    fake_compute<fshare<LOGN, F, 3>()>(x, c1, c2);
    store_block<T, ST>(x, rsrc_of<LOGN>(a + (b << LOGN)), t);
  }
  if(t == 0 && blockIdx.x < 4096) {
    Clk k;
    k.c0            = c0;
    k.r0            = r0;
    k.c1            = __builtin_amdgcn_s_memtime();
    k.r1            = __builtin_amdgcn_s_memrealtime();
    clk[blockIdx.x] = k;
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
static size_t     g_n;
static Clk *      g_clk;
static hipEvent_t g_e0, g_e1;
static int        g_reps = 16;

template <class L> static void time_it(const char *label, L launch, double bytes)
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
  const float      med = tail[tail.size() / 2], best = tail[0];
  std::vector<Clk> h(256);
  CK(hipMemcpy(h.data(), g_clk, 256 * sizeof(Clk), hipMemcpyDeviceToHost));
  std::vector<double> f;
  for(const Clk &k : h)
    if(k.r1 > k.r0) f.push_back((double)(k.c1 - k.c0) / (double)(k.r1 - k.r0) * 0.1);
  double ghz = 0;
  if(!f.empty()) {
    std::sort(f.begin(), f.end());
    ghz = f[f.size() / 2];
  }
  printf("%-66s med %7.3f ms  best %7.3f  %5.2f TB/s  frac %.3f  clk %.2f GHz\n", label, med, best, bytes / med * 1e-9, bytes / med * 1e-9 / 8.0, ghz);
  fflush(stdout);
}

template <int LOGN, int ST, int F, int XB, int XW, int LDSB, int WPS> static void run(const char *label, int wg_per_cu)
{
  constexpr int  T       = 1 << (LOGN - 4);
  const uint64_t nblocks = g_n >> LOGN;
  char           full[200];
  snprintf(full, sizeof full, "2^%d T%-4d st%d F%-2d XB%d XW%d lds+%-2dK wg/cu %d  %s", LOGN, T, ST, F, XB, XW, LDSB >> 10, wg_per_cu, label);
  time_it(
    full, [&] { hipLaunchKernelGGL((k_skel<LOGN, ST, F, XB, XW, LDSB, WPS>), dim3(256 * wg_per_cu), dim3(T), 0, 0, g_buf, nblocks, 0.999999, 1e-9, g_clk); },
    (double)g_n * 16.0);
}

int main(int argc, char **argv)
{
  const double gib = argc > 1 ? atof(argv[1]) : 8.0;
  if(argc > 2) g_reps = atoi(argv[2]);
  g_n = ((size_t)(gib * 1024.0 * 1024.0 * 1024.0 / 8.0) >> 14) << 14;
  CK(hipMalloc(&g_buf, g_n * 8));
  CK(hipMalloc(&g_clk, 4096 * sizeof(Clk)));
  CK(hipEventCreate(&g_e0));
  CK(hipEventCreate(&g_e1));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, g_n);
  CK(hipDeviceSynchronize());
  printf("# %.1f GiB in place, %d launches per row (median of the second half)\n", gib, g_reps);
  constexpr int PIN = 36 << 10; /* rows without an exchange buffer: pin four 256-thread workgroups per CU by LDS like the kernel */
  /* F: 2^14 kernel 1171 VALU per wave and block = 73 per coefficient; 2^12 kernel: its own count (PMC), about 66 */
  puts("# --- 2^12 shape: 256 threads x 16, four workgroups per CU ---");
  run<12, ST_HALF, 0, 0, 0, PIN, 4>("memory only", 4);
  run<12, ST_SPLIT, 0, 0, 0, PIN, 4>("memory only, whole-line stores", 4);
  run<12, ST_HALF, 0, 1, 0, 0, 4>("+ cross-wave exchange", 4);
  run<12, ST_HALF, 0, 1, 2, 0, 4>("+ all three exchanges", 4);
  run<12, ST_HALF, 48, 0, 0, PIN, 4>("", 4);
  run<12, ST_HALF, 60, 0, 0, PIN, 4>("", 4);
  run<12, ST_HALF, 66, 0, 0, PIN, 4>("VALU count of the 2^12 kernel, no barrier", 4);
  run<12, ST_HALF, 72, 0, 0, PIN, 4>("", 4);
  run<12, ST_HALF, 66, 1, 0, 0, 4>("+ cross-wave exchange", 4);
  run<12, ST_HALF, 66, 1, 2, 0, 4>("+ all three exchanges (the kernel's LDS traffic)", 4);
  run<12, ST_SPLIT, 66, 1, 2, 0, 4>("the same, whole-line stores", 4);
  run<12, ST_HALF, 66, 0, 2, 0, 4>("wave-local exchanges only", 4);
  run<12, ST_HALF, 66, 1, 2, 0, 4>("three workgroups per CU", 3);
  run<12, ST_HALF, 66, 1, 2, 0, 4>("two workgroups per CU", 2);
  puts("# --- 2^14 shape: 1024 threads x 16, one workgroup per CU ---");
  run<14, ST_HALF, 0, 0, 0, 0, 4>("memory only", 1);
  run<14, ST_SPLIT, 0, 0, 0, 0, 4>("memory only, whole-line stores", 1);
  run<14, ST_HALF, 0, 1, 2, 0, 4>("+ all three exchanges", 1);
  run<14, ST_HALF, 66, 0, 0, 0, 4>("", 1);
  run<14, ST_HALF, 72, 0, 0, 0, 4>("VALU count of the 2^14 kernel, no barrier", 1);
  run<14, ST_HALF, 66, 1, 2, 0, 4>("2^12's VALU count in the 2^14 shape", 1);
  run<14, ST_HALF, 72, 1, 0, 0, 4>("+ cross-wave exchange", 1);
  run<14, ST_HALF, 72, 1, 2, 0, 4>("+ all three exchanges (the kernel's LDS traffic)", 1);
  run<14, ST_SPLIT, 72, 1, 2, 0, 4>("the same, whole-line stores (the shipped shape)", 1);
  return 0;
}
