/*
 * skel5.hip -- second skeleton of the XCD-local two-pass transform (diagnostic tool, GPU box only): the structure a real
 * kernel would have, with FMAs standing in for the butterflies.  What tools/skel4.hip established (profiles/r03):
 * plain column stores + sc1 ("write-through, drop the line") final stores keep the intermediate in the XCD's L2 (FETCH_SIZE
 * = 1.0x the data instead of 2.0x), 0.59 of the HBM peak memory-only at 2^16 -- but workgroup barriers inside the items
 * (the cross-wave exchange of a 256-row column tile) halve that.  Here:
 *   - split 6 + (m - 6) stages: column items are 64 rows x 64 columns (stride N/64), a WAVE owns 16 columns, so the
 *     exchange between the two stage groups of the column pass stays inside the wave: no workgroup barrier;
 *     row items are 4096 consecutive elements = 4 rows of 1024 (2^16: one wave per row, wave-local exchanges) or 2 rows of
 *     2048 (2^17: one workgroup barrier pair per item);
 *   - both directions of every access are whole 128-byte segments per wave instruction (4 rows x 128 B);
 *   - twiddles: column pass 63 wave-uniform values (nothing to model); row pass from a 32 KiB LDS table that a workgroup
 *     keeps for its row class (LDS footprint modelled, no global reads) or, TW = 1, per-lane global reads;
 *   - queues per XCD (HW_REG_XCC_ID): one in-order queue of column items, one queue of row items per class (class =
 *     position of the 4096-element tile inside the polynomial; a workgroup keeps one class).  A workgroup prefers its
 *     pending row item as soon as that polynomial's column items are all done, else takes the next column item: the lag
 *     is whatever the machine needs, not a parameter.  Column items never wait: no deadlock whatever the residency.
 * Pass 1 adds 1, pass 2 doubles; every element is checked.
 * Build: make skel5      Run: build/skel5 [GiB] [reps] [filter]
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                 \
  do {                                                        \
    hipError_t e_ = (x);                                      \
    if(e_ != hipSuccess) {                                    \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
      exit(1);                                                \
    }                                                         \
  } while(0)

typedef unsigned int v2u32 __attribute__((ext_vector_type(2)));
typedef unsigned int v4u32 __attribute__((ext_vector_type(4)));
struct alignas(16) d2 {
  double a, b;
};

constexpr int T    = 256;
constexpr int C    = 16;
constexpr int TILE = T * C;
constexpr int MAXC = 32; /* row classes per polynomial (2^17: 32) */

struct Ctl {
  unsigned col_next[8][32];        /* per XCD: next column item                       */
  unsigned slot[8][32];            /* per XCD: workgroups that reported (class = slot mod classes) */
  unsigned spins[8][32];
  unsigned row_next[8][MAXC][32];  /* per XCD and class: next row item (polynomial)   */
  unsigned done[1];                /* [polys] column items finished                   */
};

__device__ __forceinline__ unsigned xcc_id()
{
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 7u;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc_of(const void *p, uint32_t bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
template <int AUX> __device__ __forceinline__ double ld8(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, AUX));
}
template <int AUX> __device__ __forceinline__ void st8(double x, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u32, x), r, (int)voff, (int)soff, AUX);
}
template <int F> __device__ __forceinline__ void fake_compute(double (&x)[C], double c1, double c2)
{
#pragma unroll
  for(int f = 0; f < F; f++) {
#pragma unroll
    for(int e = 0; e < C; e++) x[e] = __builtin_fma(x[e], c1, c2);
  }
}
__device__ __forceinline__ void wave_sync()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* column item: 64 rows (h) x 64 columns, wave w owns columns 16 w .. 16 w + 15; element (h, c) at h * rowlen + c0 + c.
 * in : slot e <-> h = 4 e + (lane >> 4)      out: slot e <-> h = 16 (lane >> 4) + e      (lane & 15 = column) */
template <int LA, int SA, int F>
__device__ __forceinline__ void col_item(double *poly, uint32_t nbytes, uint32_t rowlen, uint32_t tile, uint32_t t, double *wl,
                                         double c1, double c2)
{
  const __amdgpu_buffer_rsrc_t r = rsrc_of(poly, nbytes);
  const uint32_t lane = t & 63u, w = t >> 6, hl = lane >> 4, c = lane & 15u;
  const uint32_t c0 = 64u * tile + 16u * w;
  double         x[C];
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = ld8<LA>(r, (hl * rowlen + c) * 8u, ((uint32_t)e * 4u * rowlen + c0) * 8u);
#pragma unroll
  for(int e = 0; e < C; e++) x[e] += 1.0;
  fake_compute<(F * 2) / 3>(x, c1, c2); /* four of the six stages */
#pragma unroll
  for(int e = 0; e < C; e++) wl[((uint32_t)e * 4u + hl) * 17u + c] = x[e];
  wave_sync();
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = wl[(hl * 16u + (uint32_t)e) * 17u + c];
  wave_sync();
  fake_compute<F - (F * 2) / 3>(x, c1, c2);
#pragma unroll
  for(int e = 0; e < C; e++) st8<SA>(x[e], r, (hl * 16u * rowlen + c) * 8u, ((uint32_t)e * rowlen + c0) * 8u);
}

/* row item, 2^16: wave w owns row 4 tile + w (1024 consecutive elements); two wave-local exchanges (stage groups 4,4,2);
 * 2^17 (WIDE): the workgroup's two halves own rows of 2048: first exchange across the two waves of a half (workgroup barrier) */
template <int LA, int SA, int F, int TW, bool WIDE>
__device__ __forceinline__ void row_item(double *poly, uint32_t tile, uint32_t t, double *lds, double c1, double c2, const double *tw,
                                         double c0)
{
  constexpr uint32_t RT = WIDE ? 128u : 64u; /* threads per row */
  constexpr uint32_t RL = RT * 16u;          /* row length       */
  const uint32_t sub = t / RT, tt = t % RT;
  double *       rl  = lds + sub * (16u * (RT + 1u)); /* this row's exchange buffer: 16 x (RT + 1) */
  const __amdgpu_buffer_rsrc_t r = rsrc_of(poly + (size_t)tile * TILE + (size_t)sub * RL, RL * 8u);
  double x[C];
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = ld8<LA>(r, tt * 8u, (uint32_t)e * RT * 8u);
  if constexpr(TW) {
    const double *w = tw + (size_t)tile * TILE + (size_t)sub * RL + tt;
#pragma unroll
    for(int e = 0; e < C; e++) x[e] = __builtin_fma(w[(size_t)e * RT], c0, x[e]);
  }
#pragma unroll
  for(int e = 0; e < C; e++) x[e] *= 2.0;
  constexpr int FA = (F * 4) / (WIDE ? 11 : 10);
  fake_compute<FA>(x, c1, c2);
  /* exchange 1: index e * RT + tt  ->  index (tt / 16) * 256 + e * 16 + (tt % 16) */
  if constexpr(WIDE) __syncthreads();
#pragma unroll
  for(int e = 0; e < C; e++) rl[(uint32_t)e * (RT + 1u) + tt] = x[e];
  if constexpr(WIDE) __syncthreads();
  else wave_sync();
#pragma unroll
  for(int e = 0; e < C; e++) {
    const uint32_t i = (tt >> 4) * 256u + (uint32_t)e * 16u + (tt & 15u);
    x[e]             = rl[(i / RT) * (RT + 1u) + (i % RT)];
  }
  if constexpr(WIDE) __syncthreads(); /* both waves of the row have read before either overwrites the buffer */
  else wave_sync();
  fake_compute<FA>(x, c1, c2);
  /* exchange 2 (wave-local: each wave keeps its 1024 consecutive elements): -> runs of four consecutive elements per lane */
#pragma unroll
  for(int e = 0; e < C; e++) {
    const uint32_t i           = (tt >> 4) * 256u + (uint32_t)e * 16u + (tt & 15u);
    rl[(i / RT) * (RT + 1u) + (i % RT)] = x[e];
  }
  wave_sync();
#pragma unroll
  for(int e = 0; e < C; e++) {
    const uint32_t i = (tt >> 6) * 1024u + (((uint32_t)e >> 2) * 64u + (tt & 63u)) * 4u + ((uint32_t)e & 3u);
    x[e]             = rl[(i / RT) * (RT + 1u) + (i % RT)];
  }
  wave_sync();
  fake_compute<F - 2 * FA>(x, c1, c2);
  /* whole-line stores as in store_last_whole_lines (ntt_kernels.h): one v_permlane32_swap per dword exchanges slot bit 1
   * with lane bit 5; lanes 0-31 then hold the even 16-byte chunks of a 1-KiB run, lanes 32-63 the odd ones */
  {
    const uint32_t lane = tt & 63u, wv = tt >> 6;
#pragma unroll
    for(int e = 0; e < C; e++) {
      if((e & 2) == 0) {
        const v2u32 a = __builtin_bit_cast(v2u32, x[e]), b = __builtin_bit_cast(v2u32, x[e | 2]);
        const auto  lo = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
        const auto  hi = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
        x[e]           = __builtin_bit_cast(double, v2u32{lo[0], hi[0]});
        x[e | 2]       = __builtin_bit_cast(double, v2u32{lo[1], hi[1]});
      }
    }
#pragma unroll
    for(int h = 0; h < C / 2; h++) {
      const uint32_t e0 = 2u * (uint32_t)h, q = e0 >> 2, b1 = (e0 >> 1) & 1u;
      const uint32_t ql = lane < 32u ? (b1 ? lane + 32u : lane) : (b1 ? lane : lane - 32u);
      const uint32_t i  = wv * 1024u + (q * 64u + ql) * 4u + (lane < 32u ? 0u : 2u);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, d2{x[e0], x[e0 + 1]}), r, (int)(i * 8u), 0, SA);
    }
  }
}

/* mode 0: fused; 1: column items only (own launch); 2: row items only (own launch) */
template <int LA1, int SA1, int LA2, int SA2, int F, int TW, bool WIDE>
__global__ void __launch_bounds__(T, 2) k_five(double *a, uint32_t logn, uint32_t batch, int mode, Ctl *ctl, double c1, double c2,
                                               const double *tw, double c0, int prefer_rows)
{
  constexpr uint32_t RT = WIDE ? 128u : 64u;
  __shared__ double   lds[(T / RT) * 16 * (RT + 1) > 4 * 64 * 17 ? (T / RT) * 16 * (RT + 1) : 4 * 64 * 17];
  __shared__ double   table[TW ? 1 : 4096]; /* the class's twiddle table: 32 KiB of LDS the real kernel would hold */
  __shared__ unsigned s_k;
  const uint32_t      t      = threadIdx.x;
  const uint32_t      xcc    = xcc_id();
  const uint32_t      N      = 1u << logn;
  const uint32_t      rowlen = N >> 6;
  const uint32_t      NT     = N / TILE; /* items per pass and polynomial = classes */
  const uint32_t      J      = batch > xcc ? (batch - xcc + 7u) / 8u : 0u;
  if(!TW && t == 0) table[1] = c1;
  if(t == 0) s_k = atomicAdd(&ctl->slot[xcc][0], 1u);
  __syncthreads();
  const uint32_t cls = s_k % NT;
  __syncthreads();
  double *const wl = lds + (t >> 6) * (64 * 17);
  unsigned      spins = 0;
  int64_t       pend  = -1; /* row item in hand: index into this XCD's polynomial list */
  bool          rows_left = mode != 1, cols_left = mode != 2;
  for(;;) {
    /* 1. a row item of my class in hand? */
    if(rows_left && pend < 0) {
      if(t == 0) s_k = atomicAdd(&ctl->row_next[xcc][cls][0], 1u);
      __syncthreads();
      pend = s_k;
      __syncthreads();
      if(pend >= (int64_t)J) {
        rows_left = false;
        pend      = -1;
      }
    }
    bool ready = false;
    if(pend >= 0) {
      const uint32_t p = xcc + 8u * (uint32_t)pend;
      if(mode == 2) {
        ready = true;
      } else {
        if(t == 0) s_k = __hip_atomic_load(&ctl->done[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= NT;
        __syncthreads();
        ready = s_k != 0;
        __syncthreads();
      }
      if(ready && (prefer_rows || !cols_left)) {
        row_item<LA2, SA2, F, TW, WIDE>(a + ((size_t)p << logn), cls, t, lds, c1, c2, tw, c0);
        pend = -1;
        continue;
      }
    }
    /* 2. next column item */
    if(cols_left) {
      if(t == 0) s_k = atomicAdd(&ctl->col_next[xcc][0], 1u);
      __syncthreads();
      const uint32_t k = s_k;
      __syncthreads();
      if(k < J * NT) {
        const uint32_t p = xcc + 8u * (k / NT);
        col_item<LA1, SA1, F>(a + ((size_t)p << logn), N * 8u, rowlen, k % NT, t, wl, c1, c2);
        if(mode == 0) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          if(t == 0) __hip_atomic_fetch_add(&ctl->done[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if(!prefer_rows && pend >= 0 && ready) {
          /* (policy 0: strictly alternate column item, row item) */
          row_item<LA2, SA2, F, TW, WIDE>(a + ((size_t)(xcc + 8u * (uint32_t)pend) << logn), cls, t, lds, c1, c2, tw, c0);
          pend = -1;
        }
        continue;
      }
      cols_left = false;
      continue;
    }
    /* 3. nothing but a row item that is not ready yet: wait for it */
    if(pend >= 0) {
      __builtin_amdgcn_s_sleep(8);
      spins++;
      continue;
    }
    break;
  }
  if(t == 0 && spins) atomicAdd(&ctl->spins[xcc][0], spins);
}

__global__ void __launch_bounds__(256) k_fill(double *a, size_t n)
{
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    a[i] = (double)((i * 2654435761ull) & 0xfffffull);
}
__global__ void __launch_bounds__(256) k_check(const double *a, size_t n, unsigned long long *bad)
{
  unsigned long long b = 0;
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double x0 = (double)((i * 2654435761ull) & 0xfffffull);
    b += a[i] != 2.0 * (x0 + 1.0);
  }
  if(b) atomicAdd(bad, b);
}

static double *            g_buf;
static double *            g_tw;
static size_t              g_n;
static Ctl *               g_ctl;
static size_t              g_ctl_bytes;
static unsigned long long *g_bad;
static hipEvent_t          g_e0, g_e1;
static int                 g_reps = 12;
static const char *        g_filter = nullptr;

template <int LA1, int SA1, int LA2, int SA2, int F, int TW, bool WIDE> static void run(int wpc, int mode, int prefer_rows, const char *note)
{
  const int logn = WIDE ? 17 : 16;
  char      label[200];
  snprintf(label, sizeof label, "m%d mode%d wpc%d pol%d la1 %2d sa1 %2d la2 %2d sa2 %2d F%-2d TW%d %s", logn, mode, wpc, prefer_rows, LA1, SA1,
           LA2, SA2, F, TW, note);
  if(g_filter && !strstr(label, g_filter)) return;
  const uint32_t batch = (uint32_t)(g_n >> logn);
  auto           go    = [&](int md) {
    CK(hipMemsetAsync(g_ctl, 0, g_ctl_bytes));
    hipLaunchKernelGGL((k_five<LA1, SA1, LA2, SA2, F, TW, WIDE>), dim3(256 * wpc), dim3(T), 0, 0, g_buf, (uint32_t)logn, batch, md, g_ctl,
                       1.0, 0.0, g_tw, 0.0, prefer_rows);
  };
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, g_n);
  CK(hipMemset(g_bad, 0, 8));
  if(mode == 0) go(0);
  else {
    go(1);
    go(2);
  }
  hipLaunchKernelGGL(k_check, dim3(8192), dim3(256), 0, 0, g_buf, g_n, g_bad);
  unsigned long long bad = 0;
  CK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
  std::vector<unsigned> cen(sizeof(Ctl) / 4);
  CK(hipMemcpy(cen.data(), g_ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
  const Ctl *census = reinterpret_cast<const Ctl *>(cen.data());
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, g_n);
  std::vector<float> ms;
  for(int rpt = 0; rpt < g_reps; rpt++) {
    float m = 0;
    CK(hipMemsetAsync(g_ctl, 0, g_ctl_bytes));
    CK(hipEventRecord(g_e0));
    if(mode == 0) {
      hipLaunchKernelGGL((k_five<LA1, SA1, LA2, SA2, F, TW, WIDE>), dim3(256 * wpc), dim3(T), 0, 0, g_buf, (uint32_t)logn, batch, 0, g_ctl,
                         1.0, 0.0, g_tw, 0.0, prefer_rows);
    } else {
      hipLaunchKernelGGL((k_five<LA1, SA1, LA2, SA2, F, TW, WIDE>), dim3(256 * wpc), dim3(T), 0, 0, g_buf, (uint32_t)logn, batch, 1, g_ctl,
                         1.0, 0.0, g_tw, 0.0, prefer_rows);
      CK(hipMemsetAsync(g_ctl, 0, g_ctl_bytes));
      hipLaunchKernelGGL((k_five<LA1, SA1, LA2, SA2, F, TW, WIDE>), dim3(256 * wpc), dim3(T), 0, 0, g_buf, (uint32_t)logn, batch, 2, g_ctl,
                         1.0, 0.0, g_tw, 0.0, prefer_rows);
    }
    CK(hipEventRecord(g_e1));
    CK(hipEventSynchronize(g_e1));
    CK(hipEventElapsedTime(&m, g_e0, g_e1));
    ms.push_back(m);
    if((rpt & 7) == 7) hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, g_n);
  }
  CK(hipGetLastError());
  std::vector<float> tail(ms.begin() + g_reps / 2, ms.end());
  std::sort(tail.begin(), tail.end());
  const float  med = tail[tail.size() / 2], best = tail[0];
  const double bytes = (double)g_n * 16.0;
  unsigned     wmin = ~0u, wmax = 0, sp = 0;
  for(int x = 0; x < 8; x++) {
    wmin = std::min(wmin, census->slot[x][0]);
    wmax = std::max(wmax, census->slot[x][0]);
    sp += census->spins[x][0];
  }
  printf("%-64s med %7.3f ms best %7.3f  %5.2f TB/s frac %.3f  bad %llu  wg/xcd %u..%u spins %u\n", label, med, best, bytes / med * 1e-9,
         bytes / med * 1e-9 / 8.0, bad, wmin, wmax, sp);
  fflush(stdout);
}

int main(int argc, char **argv)
{
  const double gib = argc > 1 ? atof(argv[1]) : 16.0;
  if(argc > 2) g_reps = atoi(argv[2]);
  if(argc > 3) g_filter = argv[3];
  g_n = (size_t)(gib * 1024.0 * 1024.0 * 1024.0 / 8.0);
  g_n &= ~((size_t)(1u << 17) * 8 - 1);
  CK(hipMalloc(&g_buf, g_n * 8));
  CK(hipMalloc(&g_tw, (size_t)8 << 17));
  CK(hipMemset(g_tw, 0, (size_t)8 << 17));
  g_ctl_bytes = sizeof(Ctl) + (g_n >> 16) * sizeof(unsigned);
  CK(hipMalloc(&g_ctl, g_ctl_bytes));
  CK(hipMalloc(&g_bad, 8));
  CK(hipEventCreate(&g_e0));
  CK(hipEventCreate(&g_e1));
  printf("# %.1f GiB in place; algorithmic bytes = 16 per element per transform; %d launches per row (median of the second half)\n", gib,
         g_reps);
  constexpr int NTL = 2, SC1 = 16, S01 = 17;
#define BOTH(LA1, SA1, LA2, SA2, F, TW, wpc, mode, pol, note)          \
  run<LA1, SA1, LA2, SA2, F, TW, false>(wpc, mode, pol, note);          \
  run<LA1, SA1, LA2, SA2, F, TW, true>(wpc, mode, pol, note);
  puts("# two launches (one per pass) with the same items: the baseline of this tiling");
  BOTH(NTL, 0, NTL, 0, 0, 0, 2, 1, 1, "two launches")
  BOTH(NTL, 0, NTL, 0, 80, 0, 2, 1, 1, "two launches")
  puts("# fused, XCD-local: memory + exchanges only (F0), then with the FP64 work of a transform (F = FMAs per element and pass)");
  for(int wpc : {2}) {
    for(int pol : {1, 0}) {
      BOTH(S01, 0, NTL, SC1, 0, 0, wpc, 0, pol, "")
      BOTH(S01, 0, NTL, SC1, 40, 0, wpc, 0, pol, "")
      BOTH(S01, 0, NTL, SC1, 56, 0, wpc, 0, pol, "")
      BOTH(S01, 0, NTL, SC1, 40, 1, wpc, 0, pol, "table reads from L2")
      BOTH(NTL, 0, NTL, SC1, 40, 0, wpc, 0, pol, "")
      BOTH(NTL, 0, NTL, 0, 40, 0, wpc, 0, pol, "plain final stores")
    }
  }
  return 0;
}
