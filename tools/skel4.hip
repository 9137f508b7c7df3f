/*
 * skel4.hip -- memory skeleton of an XCD-local "four-step" transform for N = 2^16 / 2^17 (diagnostic tool, GPU box only).
 *
 * Question: can the intermediate of a two-pass transform stay inside ONE XCD's 4 MiB L2, so that
 * only 16*N bytes per transform cross the fabric instead of 32*N?  Shape tried here:
 *   - the polynomial is cut into 4096-element items for each pass: "column" items (16 adjacent columns x 256 rows, stride
 *     N/256: the leading 8 stages) and "row" items (4096 consecutive elements: the remaining stages);
 *   - 256-thread workgroups, 16 elements per thread, several workgroups per CU;
 *   - a workgroup reads its XCD from HW_REG_XCC_ID and pulls items from THAT XCD's queue (one atomic counter per XCD), so
 *     all items of a polynomial run on one XCD whatever the dispatcher does -- placement is read, never assumed;
 *   - queue order: col(poly j), row(poly j - LAG), col(j + 1), ...; a row item waits on a per-polynomial counter that the
 *     column items bump after their stores have completed (s_waitcnt vmcnt(0) + barrier + agent-scope atomic).  Column
 *     items never wait, and every item a row item waits for was handed out earlier to a running workgroup: no deadlock
 *     whatever the residency;
 *   - column stores stay dirty in the L2 (plain stores); row loads bypass the L1 (nt / sc1).
 * Pass 1 adds 1, pass 2 doubles: the result 2(x+1) is checked element by element, so an ordering or visibility error of
 * the hand-off shows up as a mismatch count.
 * Build: make skel4      Run: build/skel4 [GiB] [reps] [filter]
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
constexpr int TILE = T * C; /* elements per item */

struct Ctl {
  unsigned next[8][32];  /* per-XCD item counter, one 128-byte line each */
  unsigned nwg[8][32];   /* census: workgroups that reported this XCD     */
  unsigned spins[8][32]; /* poll iterations spent waiting, per XCD        */
  unsigned done[1];      /* [polys] column items finished (flexible)      */
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
template <int AUX> __device__ __forceinline__ void st16(d2 x, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, x), r, (int)voff, (int)soff, AUX);
}

template <int F> __device__ __forceinline__ void fake_compute(double (&x)[C], double c1, double c2)
{
#pragma unroll
  for(int f = 0; f < F; f++) {
#pragma unroll
    for(int e = 0; e < C; e++) x[e] = __builtin_fma(x[e], c1, c2);
  }
}

/* column item: element (h, c) of the tile, h = 0..255, c = 0..15, lives at h * rowlen + 16 * tile + c.
 * in : slot e <-> h = 16 e + (t >> 4)   (first stage group: top four bits of h in the thread)
 * out: slot e <-> h = 16 (t >> 4) + e   (second stage group: low four bits of h in the thread) */
template <int LA, int SA, int F, int X>
__device__ __forceinline__ void col_item(double *poly, uint32_t nbytes, uint32_t rowlen, uint32_t tile, uint32_t t, double *lds,
                                         double c1, double c2, double add)
{
  const __amdgpu_buffer_rsrc_t r = rsrc_of(poly, nbytes);
  const uint32_t hl = t >> 4, c = t & 15u;
  double         x[C];
  const uint32_t vin = (hl * rowlen + c) * 8u;
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = ld8<LA>(r, vin, ((uint32_t)e * 16u * rowlen + 16u * tile) * 8u);
#pragma unroll
  for(int e = 0; e < C; e++) x[e] += add;
  fake_compute<F / 2>(x, c1, c2);
  if constexpr(X) {
    __syncthreads();
#pragma unroll
    for(int e = 0; e < C; e++) lds[((uint32_t)e * 16u + hl) * 17u + c] = x[e];
    __syncthreads();
#pragma unroll
    for(int e = 0; e < C; e++) x[e] = lds[(hl * 16u + (uint32_t)e) * 17u + c];
    fake_compute<F - F / 2>(x, c1, c2);
    const uint32_t vout = (hl * 16u * rowlen + c) * 8u;
#pragma unroll
    for(int e = 0; e < C; e++) st8<SA>(x[e], r, vout, ((uint32_t)e * rowlen + 16u * tile) * 8u);
  } else {
    fake_compute<F - F / 2>(x, c1, c2);
#pragma unroll
    for(int e = 0; e < C; e++) st8<SA>(x[e], r, vin, ((uint32_t)e * 16u * rowlen + 16u * tile) * 8u);
  }
}

/* column item of the library's existing first pass (column_pass_thread, four stages held in registers, no exchange):
 * 256 consecutive columns x 16 rows at stride N/16; every access is a contiguous 2 KiB row segment */
template <int LA, int SA, int F>
__device__ __forceinline__ void col16_item(double *poly, uint32_t nbytes, uint32_t n, uint32_t tile, uint32_t t, double c1, double c2)
{
  const __amdgpu_buffer_rsrc_t r = rsrc_of(poly, nbytes);
  const uint32_t span = n >> 4;
  double         x[C];
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = ld8<LA>(r, t * 8u, ((uint32_t)e * span + 256u * tile) * 8u);
#pragma unroll
  for(int e = 0; e < C; e++) x[e] += 1.0;
  fake_compute<F>(x, c1, c2);
#pragma unroll
  for(int e = 0; e < C; e++) st8<SA>(x[e], r, t * 8u, ((uint32_t)e * span + 256u * tile) * 8u);
}

__device__ __forceinline__ void wave_sync4()
{
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

/* row item shaped like the library's 2^12-point block (fused_kernel<., 12>: stage groups 2,4,4,2): one cross-wave exchange
 * (two barriers), two wave-local ones, a 7.5 KiB twiddle table copied into LDS per item (coalesced reads of a table the
 * whole batch shares), 12 per-lane table reads for the last group, whole-line final stores */
template <int LA, int SA, int F>
__device__ __forceinline__ void row12_item(double *poly, uint32_t tile, uint32_t t, double *lds, double *tab, double c1, double c2,
                                           const double *tw, double c0)
{
  const __amdgpu_buffer_rsrc_t r = rsrc_of(poly + (size_t)tile * TILE, TILE * 8u);
  double x[C];
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = ld8<LA>(r, t * 8u, (uint32_t)e * T * 8u);
  /* this block position's table slice: 960 words */
  const double *w = tw + (size_t)tile * TILE;
  double tl[4];
#pragma unroll
  for(int k = 0; k < 4; k++) tl[k] = (t + 256u * k) < 960u ? w[t + 256u * k] : 0.0;
  double pl[12];
#pragma unroll
  for(int k = 0; k < 12; k++) pl[k] = w[1024u + (uint32_t)k * 256u + t];
#pragma unroll
  for(int e = 0; e < C; e++) x[e] *= 2.0;
  fake_compute<F / 6>(x, c1, c2); /* 2 of 12 stages */
  __syncthreads();
#pragma unroll
  for(int e = 0; e < C; e++) lds[e * (T + 1) + t] = x[e];
#pragma unroll
  for(int k = 0; k < 4; k++)
    if((t + 256u * k) < 960u) tab[t + 256u * k] = tl[k];
  __syncthreads();
#pragma unroll
  for(int e = 0; e < C; e++) {
    const uint32_t i = (t >> 4) * 256u + (uint32_t)e * 16u + (t & 15u);
    x[e]             = lds[(i >> 8) * (T + 1) + (i & 255u)];
  }
  fake_compute<F / 3>(x, c1, c2); /* 4 stages */
  wave_sync4();
  /* (wave-local exchanges: every wave owns 1024 consecutive elements from here on) */
#pragma unroll
  for(int e = 0; e < C; e++) {
    const uint32_t i                    = (t >> 4) * 256u + (uint32_t)e * 16u + (t & 15u);
    lds[(i >> 8) * (T + 1) + (i & 255u)] = x[e];
  }
  wave_sync4();
#pragma unroll
  for(int e = 0; e < C; e++) {
    const uint32_t i = (t >> 6) * 1024u + (uint32_t)e * 64u + (t & 63u);
    x[e]             = lds[(i >> 8) * (T + 1) + (i & 255u)] + tab[(e * 60 + (t & 63u)) % 960u] * c0;
  }
  fake_compute<F / 3>(x, c1, c2); /* 4 stages */
  wave_sync4();
#pragma unroll
  for(int e = 0; e < C; e++) {
    const uint32_t i                    = (t >> 6) * 1024u + (uint32_t)e * 64u + (t & 63u);
    lds[(i >> 8) * (T + 1) + (i & 255u)] = x[e];
  }
  wave_sync4();
#pragma unroll
  for(int e = 0; e < C; e++) {
    const uint32_t i = (t >> 6) * 1024u + (((uint32_t)e >> 2) * 64u + (t & 63u)) * 4u + ((uint32_t)e & 3u);
    x[e]             = lds[(i >> 8) * (T + 1) + (i & 255u)] + pl[e % 12] * c0;
  }
  fake_compute<F - F / 6 - 2 * (F / 3)>(x, c1, c2); /* 2 stages */
  {
    const uint32_t lane = t & 63u, wv = t >> 6;
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

/* row item: 4096 consecutive elements; 8-byte coalesced loads, 16-byte stores in runs of four per lane */
template <int LA, int SA, int F, int X, int TW>
__device__ __forceinline__ void row_item(double *poly, uint32_t nbytes, uint32_t tile, uint32_t t, double *lds, double c1, double c2,
                                         double mul, const double *tw, double c0)
{
  const __amdgpu_buffer_rsrc_t r = rsrc_of(poly + (size_t)tile * TILE, TILE * 8u);
  (void)nbytes;
  double x[C];
#pragma unroll
  for(int e = 0; e < C; e++) x[e] = ld8<LA>(r, t * 8u, (uint32_t)e * T * 8u);
  if constexpr(TW) {
    /* as many per-lane 8-byte table reads as a transform's per-lane twiddles (cacheable: the table is shared by all polynomials) */
    const double *w = tw + (size_t)tile * TILE + t;
#pragma unroll
    for(int e = 0; e < C; e++) x[e] = __builtin_fma(w[(size_t)e * T], c0, x[e]);
  }
#pragma unroll
  for(int e = 0; e < C; e++) x[e] *= mul;
  fake_compute<F / 2>(x, c1, c2);
  if constexpr(X) {
    __syncthreads();
#pragma unroll
    for(int e = 0; e < C; e++) lds[e * (T + 1) + t] = x[e];
    __syncthreads();
    /* like the real kernels' last group: thread t then owns runs of four consecutive elements, 16-byte stores */
#pragma unroll
    for(int e = 0; e < C; e++) {
      const uint32_t i = (((uint32_t)e >> 2) * T + t) * 4u + ((uint32_t)e & 3u);
      x[e]             = lds[(i >> 8) * (T + 1) + (i & 255u)];
    }
    fake_compute<F - F / 2>(x, c1, c2);
#pragma unroll
    for(int h = 0; h < C / 2; h++) st16<SA>(d2{x[2 * h], x[2 * h + 1]}, r, t * 32u + (uint32_t)(h & 1) * 16u, (uint32_t)(h >> 1) * T * 32u);
  } else {
    fake_compute<F - F / 2>(x, c1, c2);
#pragma unroll
    for(int e = 0; e < C; e++) st8<SA>(x[e], r, t * 8u, (uint32_t)e * T * 8u);
  }
}

/* mode 0: fused (both passes, XCD-local hand-off); 1: column items only; 2: row items only (no waiting) */
template <int LA1, int SA1, int LA2, int SA2, int F, int X, int TW, int WPS>
__global__ void __launch_bounds__(T, WPS) k_four(double *a, uint32_t logn, uint32_t batch, int lag, int mode, Ctl *ctl, double c1,
                                                 double c2, const double *tw, double c0, int blocked)
{
  __shared__ double   lds[X ? 256 * 17 : 1]; /* column layout 256 x 17 (the row layout needs 16 x 257) */
  __shared__ double   tab[X == 2 ? 960 : 1];
  __shared__ unsigned s_k;
  const uint32_t      t      = threadIdx.x;
  const uint32_t      xcc    = xcc_id();
  const uint32_t      N      = 1u << logn;
  const uint32_t      rowlen = N >> 8;
  const uint32_t      NT     = N / TILE;                       /* items per pass and polynomial */
  /* polynomials of this XCD: xcc, xcc + 8, ... (interleaved) or a contiguous eighth of the batch (blocked) */
  const uint32_t      per    = (batch + 7u) / 8u;
  const uint32_t      J      = blocked ? (xcc * per < batch ? (batch - xcc * per < per ? batch - xcc * per : per) : 0u)
                                       : (batch > xcc ? (batch - xcc + 7u) / 8u : 0u);
  if(t == 0) atomicAdd(&ctl->nwg[xcc][0], 1u);
  const uint32_t per_step = mode == 0 ? 2u * NT : NT;
  const uint32_t steps    = mode == 0 ? J + (uint32_t)lag : J;
  unsigned       spins    = 0;
  for(;;) {
    if(t == 0) s_k = atomicAdd(&ctl->next[xcc][0], 1u);
    __syncthreads();
    const uint32_t k = s_k;
    __syncthreads();
    const uint32_t step = k / per_step, r = k % per_step;
    if(step >= steps) break;
    const bool     is_row = mode == 2 || (mode == 0 && r >= NT);
    const uint32_t tile   = r >= NT ? r - NT : r;
    const int64_t  j      = (mode == 0 && is_row) ? (int64_t)step - lag : (int64_t)step;
    if(j < 0 || j >= (int64_t)J) continue;
    const uint32_t p    = blocked ? xcc * ((batch + 7u) / 8u) + (uint32_t)j : xcc + 8u * (uint32_t)j;
    double *       poly = a + ((size_t)p << logn);
    if(!is_row) {
      if constexpr(X == 2) col16_item<LA1, SA1, F / 4>(poly, N * 8u, N, tile, t, c1, c2); /* 4 of 16 stages */
      else col_item<LA1, SA1, F, X>(poly, N * 8u, rowlen, tile, t, lds, c1, c2, 1.0);
      if(mode == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if(t == 0) __hip_atomic_fetch_add(&ctl->done[p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {
      if(mode == 0) {
        if(t == 0) {
          while(__hip_atomic_load(&ctl->done[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NT) {
            __builtin_amdgcn_s_sleep(8);
            spins++;
          }
        }
        __syncthreads();
      }
      if constexpr(X == 2) row12_item<LA2, SA2, F - F / 4>(poly, tile, t, lds, tab, c1, c2, tw, c0);
      else row_item<LA2, SA2, F, X, TW>(poly, N * 8u, tile, t, lds, c1, c2, 2.0, tw, c0);
    }
  }
  if(t == 0 && spins) atomicAdd(&ctl->spins[xcc][0], spins);
}

/* X2 items with the NEXT item's 16 data loads in flight while the current item is computed and stored (register double
 * buffer, as in the library's persistent single-pass kernels).  A second-pass item is prefetched only if its polynomial is
 * ready when asked; otherwise the workgroup waits for it after finishing the current item. */
template <int LA1, int SA1, int LA2, int SA2, int F, int WPS>
__global__ void __launch_bounds__(T, WPS) k_four_pipe(double *a, uint32_t logn, uint32_t batch, int lag, Ctl *ctl, double c1, double c2,
                                                      const double *tw, double c0)
{
  __shared__ double   lds[256 * 17];
  __shared__ double   tab[960];
  __shared__ unsigned s_k, s_rdy;
  const uint32_t      t   = threadIdx.x;
  const uint32_t      xcc = xcc_id();
  const uint32_t      N   = 1u << logn;
  const uint32_t      NT  = N / TILE;
  const uint32_t      J   = batch > xcc ? (batch - xcc + 7u) / 8u : 0u;
  const uint32_t      steps = J + (uint32_t)lag;
  const uint32_t      span = N >> 4;
  if(t == 0) atomicAdd(&ctl->nwg[xcc][0], 1u);
  struct Item {
    bool     valid, row;
    uint32_t tile, p;
  };
  auto decode = [&](uint32_t k, bool &end) {
    Item it{false, false, 0, 0};
    const uint32_t step = k / (2u * NT), r = k % (2u * NT);
    end = step >= steps;
    if(end) return it;
    it.row           = r >= NT;
    it.tile          = it.row ? r - NT : r;
    const int64_t j  = it.row ? (int64_t)step - lag : (int64_t)step;
    it.valid         = j >= 0 && j < (int64_t)J;
    it.p             = it.valid ? xcc + 8u * (uint32_t)j : 0u;
    return it;
  };
  auto issue = [&](const Item &it, double (&raw)[C]) {
    double *poly = a + ((size_t)it.p << logn);
    if(it.row) {
      const __amdgpu_buffer_rsrc_t r = rsrc_of(poly + (size_t)it.tile * TILE, TILE * 8u);
#pragma unroll
      for(int e = 0; e < C; e++) raw[e] = ld8<LA2>(r, t * 8u, (uint32_t)e * T * 8u);
    } else {
      const __amdgpu_buffer_rsrc_t r = rsrc_of(poly, N * 8u);
#pragma unroll
      for(int e = 0; e < C; e++) raw[e] = ld8<LA1>(r, t * 8u, ((uint32_t)e * span + 256u * it.tile) * 8u);
    }
  };
  unsigned spins = 0;
  double   raw[C];
  bool     end = false, cur_loaded = false;
  if(t == 0) s_k = atomicAdd(&ctl->next[xcc][0], 1u);
  __syncthreads();
  Item cur = decode(s_k, end);
  __syncthreads();
  while(!end) {
    /* next index */
    if(t == 0) s_k = atomicAdd(&ctl->next[xcc][0], 1u);
    /* make sure the current item is loaded (blocking poll if it is a second-pass item that was not ready before) */
    if(cur.valid && !cur_loaded) {
      if(cur.row) {
        if(t == 0) {
          while(__hip_atomic_load(&ctl->done[cur.p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NT) {
            __builtin_amdgcn_s_sleep(8);
            spins++;
          }
        }
        __syncthreads();
      }
      issue(cur, raw);
      cur_loaded = true;
    }
    __syncthreads();
    bool       nend = false;
    const Item nxt  = decode(s_k, nend);
    if(t == 0) s_rdy = (nxt.valid && nxt.row) ? (__hip_atomic_load(&ctl->done[nxt.p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= NT) : 1u;
    __syncthreads();
    const bool nready = nxt.valid && s_rdy != 0;
    __syncthreads();
    if(cur.valid) {
      double *poly = a + ((size_t)cur.p << logn);
      double  x[C];
#pragma unroll
      for(int e = 0; e < C; e++) x[e] = raw[e];
      if(nready) issue(nxt, raw); /* the next item's loads are in flight from here on */
      if(!cur.row) {
        const __amdgpu_buffer_rsrc_t r = rsrc_of(poly, N * 8u);
#pragma unroll
        for(int e = 0; e < C; e++) x[e] += 1.0;
        fake_compute<F / 4>(x, c1, c2);
#pragma unroll
        for(int e = 0; e < C; e++) st8<SA1>(x[e], r, t * 8u, ((uint32_t)e * span + 256u * cur.tile) * 8u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if(t == 0) __hip_atomic_fetch_add(&ctl->done[cur.p], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
      } else {
        /* the row item's work on x (as row12_item, minus its loads) */
        const __amdgpu_buffer_rsrc_t r = rsrc_of(poly + (size_t)cur.tile * TILE, TILE * 8u);
        const double *w = tw + (size_t)cur.tile * TILE;
        double tl[4];
#pragma unroll
        for(int k2 = 0; k2 < 4; k2++) tl[k2] = (t + 256u * k2) < 960u ? w[t + 256u * k2] : 0.0;
        double pl[12];
#pragma unroll
        for(int k2 = 0; k2 < 12; k2++) pl[k2] = w[1024u + (uint32_t)k2 * 256u + t];
#pragma unroll
        for(int e = 0; e < C; e++) x[e] *= 2.0;
        constexpr int FR = F - F / 4;
        fake_compute<FR / 6>(x, c1, c2);
        __syncthreads();
#pragma unroll
        for(int e = 0; e < C; e++) lds[e * (T + 1) + t] = x[e];
#pragma unroll
        for(int k2 = 0; k2 < 4; k2++)
          if((t + 256u * k2) < 960u) tab[t + 256u * k2] = tl[k2];
        __syncthreads();
#pragma unroll
        for(int e = 0; e < C; e++) {
          const uint32_t i = (t >> 4) * 256u + (uint32_t)e * 16u + (t & 15u);
          x[e]             = lds[(i >> 8) * (T + 1) + (i & 255u)];
        }
        fake_compute<FR / 3>(x, c1, c2);
        wave_sync4();
#pragma unroll
        for(int e = 0; e < C; e++) {
          const uint32_t i                    = (t >> 4) * 256u + (uint32_t)e * 16u + (t & 15u);
          lds[(i >> 8) * (T + 1) + (i & 255u)] = x[e];
        }
        wave_sync4();
#pragma unroll
        for(int e = 0; e < C; e++) {
          const uint32_t i = (t >> 6) * 1024u + (uint32_t)e * 64u + (t & 63u);
          x[e]             = lds[(i >> 8) * (T + 1) + (i & 255u)] + tab[(e * 60 + (t & 63u)) % 960u] * c0;
        }
        fake_compute<FR / 3>(x, c1, c2);
        wave_sync4();
#pragma unroll
        for(int e = 0; e < C; e++) {
          const uint32_t i                    = (t >> 6) * 1024u + (uint32_t)e * 64u + (t & 63u);
          lds[(i >> 8) * (T + 1) + (i & 255u)] = x[e];
        }
        wave_sync4();
#pragma unroll
        for(int e = 0; e < C; e++) {
          const uint32_t i = (t >> 6) * 1024u + (((uint32_t)e >> 2) * 64u + (t & 63u)) * 4u + ((uint32_t)e & 3u);
          x[e]             = lds[(i >> 8) * (T + 1) + (i & 255u)] + pl[e % 12] * c0;
        }
        fake_compute<FR - FR / 6 - 2 * (FR / 3)>(x, c1, c2);
        const uint32_t lane = t & 63u, wv = t >> 6;
#pragma unroll
        for(int e = 0; e < C; e++) {
          if((e & 2) == 0) {
            const v2u32 aa = __builtin_bit_cast(v2u32, x[e]), bb = __builtin_bit_cast(v2u32, x[e | 2]);
            const auto  lo = __builtin_amdgcn_permlane32_swap(aa.x, bb.x, false, false);
            const auto  hi = __builtin_amdgcn_permlane32_swap(aa.y, bb.y, false, false);
            x[e]           = __builtin_bit_cast(double, v2u32{lo[0], hi[0]});
            x[e | 2]       = __builtin_bit_cast(double, v2u32{lo[1], hi[1]});
          }
        }
#pragma unroll
        for(int h = 0; h < C / 2; h++) {
          const uint32_t e0 = 2u * (uint32_t)h, qd = e0 >> 2, b1 = (e0 >> 1) & 1u;
          const uint32_t ql = lane < 32u ? (b1 ? lane + 32u : lane) : (b1 ? lane : lane - 32u);
          const uint32_t i  = wv * 1024u + (qd * 64u + ql) * 4u + (lane < 32u ? 0u : 2u);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u32, d2{x[e0], x[e0 + 1]}), r, (int)(i * 8u), 0, SA2);
        }
        __syncthreads();
      }
    }
    cur        = nxt;
    cur_loaded = nready;
    end        = nend;
  }
  if(t == 0 && spins) atomicAdd(&ctl->spins[xcc][0], spins);
}

__global__ void __launch_bounds__(256) k_fill(double *a, size_t n)
{
  for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    a[i] = (double)((i * 2654435761ull) & 0xfffffull);
  }
}
/* counts elements that are not 2 (x0 + 1) (fused / both passes) */
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
static int                 g_reps = 16;
static const char *        g_filter = nullptr;

struct Cfg {
  int logn, wpc, lag, mode, blocked = 0;
};

template <int LA1, int SA1, int LA2, int SA2, int F, int X, int TW, int WPS> static void run(Cfg cf, const char *note)
{
  char label[200];
  snprintf(label, sizeof label, "m%d mode%d wpc%d lag%-2d blk%d la1 %2d sa1 %2d la2 %2d sa2 %2d F%-2d X%d TW%d %s", cf.logn, cf.mode, cf.wpc, cf.lag, cf.blocked,
           LA1, SA1, LA2, SA2, F, X, TW, note);
  if(g_filter && !strstr(label, g_filter)) return;
  const uint32_t batch = (uint32_t)(g_n >> cf.logn);
  auto           launch = [&](int mode) {
    CK(hipMemsetAsync(g_ctl, 0, g_ctl_bytes));
    hipLaunchKernelGGL((k_four<LA1, SA1, LA2, SA2, F, X, TW, WPS>), dim3(256 * cf.wpc), dim3(T), 0, 0, g_buf, (uint32_t)cf.logn, batch,
                       cf.lag, mode, g_ctl, 1.0, 0.0, g_tw, 0.0, cf.blocked);
  };
  /* correctness of the hand-off first, on freshly filled data */
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, g_n);
  CK(hipMemset(g_bad, 0, 8));
  if(cf.mode == 0) {
    launch(0);
  } else {
    launch(1);
    launch(2);
  }
  hipLaunchKernelGGL(k_check, dim3(8192), dim3(256), 0, 0, g_buf, g_n, g_bad);
  unsigned long long bad = 0;
  CK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
  Ctl census;
  CK(hipMemcpy(&census, g_ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, g_n); /* keep values small: every timed launch doubles them */
  std::vector<float> ms;
  for(int rpt = 0; rpt < g_reps; rpt++) {
    float m = 0;
    if(cf.mode == 0) {
      CK(hipMemsetAsync(g_ctl, 0, g_ctl_bytes));
      CK(hipEventRecord(g_e0));
      hipLaunchKernelGGL((k_four<LA1, SA1, LA2, SA2, F, X, TW, WPS>), dim3(256 * cf.wpc), dim3(T), 0, 0, g_buf, (uint32_t)cf.logn, batch,
                         cf.lag, 0, g_ctl, 1.0, 0.0, g_tw, 0.0, cf.blocked);
      CK(hipEventRecord(g_e1));
    } else {
      /* two launches, the memset between them inside the timed region (a few microseconds) */
      CK(hipMemsetAsync(g_ctl, 0, g_ctl_bytes));
      CK(hipEventRecord(g_e0));
      hipLaunchKernelGGL((k_four<LA1, SA1, LA2, SA2, F, X, TW, WPS>), dim3(256 * cf.wpc), dim3(T), 0, 0, g_buf, (uint32_t)cf.logn, batch,
                         cf.lag, 1, g_ctl, 1.0, 0.0, g_tw, 0.0, cf.blocked);
      CK(hipMemsetAsync(g_ctl, 0, g_ctl_bytes));
      hipLaunchKernelGGL((k_four<LA1, SA1, LA2, SA2, F, X, TW, WPS>), dim3(256 * cf.wpc), dim3(T), 0, 0, g_buf, (uint32_t)cf.logn, batch,
                         cf.lag, 2, g_ctl, 1.0, 0.0, g_tw, 0.0, cf.blocked);
      CK(hipEventRecord(g_e1));
    }
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
    wmin = std::min(wmin, census.nwg[x][0]);
    wmax = std::max(wmax, census.nwg[x][0]);
    sp += census.spins[x][0];
  }
  printf("%-78s med %7.3f ms best %7.3f  %5.2f TB/s frac %.3f  bad %llu  wg/xcd %u..%u spins %u\n", label, med, best, bytes / med * 1e-9,
         bytes / med * 1e-9 / 8.0, bad, wmin, wmax, sp);
  fflush(stdout);
}

template <int LA1, int SA1, int LA2, int SA2, int F, int WPS> static void run_pipe(Cfg cf, const char *note)
{
  char label[200];
  snprintf(label, sizeof label, "m%d mode%d wpc%d lag%-2d blk%d la1 %2d sa1 %2d la2 %2d sa2 %2d F%-2d X%d TW%d %s", cf.logn, 0, cf.wpc, cf.lag, 0,
           LA1, SA1, LA2, SA2, F, 3, 0, note);
  if(g_filter && !strstr(label, g_filter)) return;
  const uint32_t batch = (uint32_t)(g_n >> cf.logn);
  auto           go    = [&] {
    CK(hipMemsetAsync(g_ctl, 0, g_ctl_bytes));
    hipLaunchKernelGGL((k_four_pipe<LA1, SA1, LA2, SA2, F, WPS>), dim3(256 * cf.wpc), dim3(T), 0, 0, g_buf, (uint32_t)cf.logn, batch, cf.lag,
                       g_ctl, 1.0, 0.0, g_tw, 0.0);
  };
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, g_n);
  CK(hipMemset(g_bad, 0, 8));
  go();
  hipLaunchKernelGGL(k_check, dim3(8192), dim3(256), 0, 0, g_buf, g_n, g_bad);
  unsigned long long bad = 0;
  CK(hipMemcpy(&bad, g_bad, 8, hipMemcpyDeviceToHost));
  Ctl census;
  CK(hipMemcpy(&census, g_ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, g_buf, g_n);
  std::vector<float> ms;
  for(int rpt = 0; rpt < g_reps; rpt++) {
    float m = 0;
    CK(hipMemsetAsync(g_ctl, 0, g_ctl_bytes));
    CK(hipEventRecord(g_e0));
    hipLaunchKernelGGL((k_four_pipe<LA1, SA1, LA2, SA2, F, WPS>), dim3(256 * cf.wpc), dim3(T), 0, 0, g_buf, (uint32_t)cf.logn, batch, cf.lag,
                       g_ctl, 1.0, 0.0, g_tw, 0.0);
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
  unsigned     sp = 0;
  for(int x = 0; x < 8; x++) sp += census.spins[x][0];
  printf("%-78s med %7.3f ms best %7.3f  %5.2f TB/s frac %.3f  bad %llu  wg/xcd %u..%u spins %u\n", label, med, best, bytes / med * 1e-9,
         bytes / med * 1e-9 / 8.0, bad, census.nwg[0][0], census.nwg[7][0], sp);
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
  constexpr int NTL = 2, SC1 = 16;
  if(getenv("SKEL4_SWEEP5")) {
    /* fifth sweep: the library-shaped items with the next item's data loads in flight during the current item (X3) */
    constexpr int S01 = SC1 | 1;
    for(int wpc : {1, 2, 3})
      for(int lag : {4, 6, 8, 10, 12, 16}) {
        run_pipe<S01, 0, NTL, SC1, 0, 4>(Cfg{16, wpc, lag, 0}, "pipelined");
        run_pipe<S01, 0, NTL, SC1, 72, 4>(Cfg{16, wpc, lag, 0}, "pipelined");
        run_pipe<S01, 0, NTL, SC1, 96, 4>(Cfg{16, wpc, lag, 0}, "pipelined");
      }
    return 0;
  }
  if(getenv("SKEL4_SWEEP4")) {
    /* fourth sweep: items shaped like the library's existing passes at 2^16 -- column item = column_pass_thread with four
     * stages (no exchange, 2 KiB row segments), row item = a 2^12-point block (one barrier pair, LDS table per item,
     * per-lane reads for the last group, whole-line stores).  F = FMAs per element over BOTH passes (split 1:3). */
    constexpr int S01 = SC1 | 1;
    puts("# two launches of the same items (the library's per-pass structure, unchunked)");
    run<NTL, 0, NTL, 0, 0, 2, 0, 4>(Cfg{16, 4, 0, 1}, "two launches");
    run<NTL, 0, NTL, 0, 72, 2, 0, 4>(Cfg{16, 4, 0, 1}, "two launches");
    run<NTL, 0, NTL, 0, 72, 2, 0, 4>(Cfg{16, 2, 0, 1}, "two launches");
    puts("# fused, XCD-local");
    for(int wpc : {2, 3, 4})
      for(int lag : {3, 4, 5, 6, 7, 8, 10}) {
        run<S01, 0, NTL, SC1, 0, 2, 0, 4>(Cfg{16, wpc, lag, 0}, "");
        run<S01, 0, NTL, SC1, 72, 2, 0, 4>(Cfg{16, wpc, lag, 0}, "");
      }
    for(int lag : {4, 5, 6, 7})
      for(int wpc : {2, 3}) {
        run<S01, 0, NTL, 0, 72, 2, 0, 4>(Cfg{16, wpc, lag, 0}, "plain final stores");
        run<NTL, 0, NTL, SC1, 72, 2, 0, 4>(Cfg{16, wpc, lag, 0}, "nt input loads");
        run<S01, 0, NTL, SC1, 96, 2, 0, 4>(Cfg{16, wpc, lag, 0}, "");
      }
    return 0;
  }
  if(getenv("SKEL4_SWEEP3")) {
    /* third sweep: lag in single steps, polynomial-to-XCD assignment, which cache-policy bit matters, then the
     * LDS exchanges, table reads and FP64 work of a real transform on top of the best shapes */
    constexpr int S01 = SC1 | 1;
    for(int m : {16, 17}) {
      printf("# N = 2^%d: lag and assignment (blk1 = each XCD owns a contiguous eighth of the batch)\n", m);
      for(int blocked : {0, 1})
        for(int lag : {2, 3, 4, 5, 6, 7, 8, 10, 12}) {
          Cfg cf{m, 2, lag, 0};
          cf.blocked = blocked;
          run<S01, 0, NTL, SC1, 0, 0, 0, 4>(cf, "");
        }
      puts("# which bits: input loads sc1 only / sc0 only / sc0 sc1 nt; row loads plain / sc0 sc1; final stores sc0 sc1");
      for(int blocked : {0, 1})
        for(int lag : {3, 5, 6}) {
          Cfg cf{m, 2, lag, 0};
          cf.blocked = blocked;
          run<SC1, 0, NTL, SC1, 0, 0, 0, 4>(cf, "");
          run<1, 0, NTL, SC1, 0, 0, 0, 4>(cf, "");
          run<S01 | NTL, 0, NTL, SC1, 0, 0, 0, 4>(cf, "");
          run<S01, 0, 0, SC1, 0, 0, 0, 4>(cf, "");
          run<S01, 0, S01, SC1, 0, 0, 0, 4>(cf, "");
          run<S01, 0, NTL, S01, 0, 0, 0, 4>(cf, "");
          run<S01, NTL, NTL, SC1, 0, 0, 0, 4>(cf, "");
        }
      puts("# + LDS exchange in both passes (X1), + per-lane table reads in the row pass (TW1), + F FP64 FMAs per element and pass");
      for(int blocked : {0, 1})
        for(int wpc : {2, 3})
          for(int lag : {3, 5, 6, 7}) {
            Cfg cf{m, wpc, lag, 0};
            cf.blocked = blocked;
            run<S01, 0, NTL, SC1, 0, 1, 0, 4>(cf, "");
            run<S01, 0, NTL, SC1, 0, 1, 1, 4>(cf, "");
            run<S01, 0, NTL, SC1, 40, 1, 1, 4>(cf, "");
          }
    }
    return 0;
  }
  const bool sweep2 = getenv("SKEL4_SWEEP2") != nullptr;
  if(sweep2) {
    /* second sweep: fewer resident workgroups (smaller in-flight footprint), longer lags, cache-policy hints on the
     * streaming sides (input loads, final stores) so that they do not displace the intermediate from the L2 */
    for(int m : {16, 17}) {
      printf("# N = 2^%d: fused; hints on the streaming sides\n", m);
      for(int wpc : {1, 2, 3})
        for(int lag : {2, 3, 4, 6, 8}) {
          run<NTL, 0, NTL, 0, 0, 0, 0, 4>(Cfg{m, wpc, lag, 0}, "");
          run<NTL, 0, NTL, NTL, 0, 0, 0, 4>(Cfg{m, wpc, lag, 0}, "");
          run<NTL, 0, NTL, SC1, 0, 0, 0, 4>(Cfg{m, wpc, lag, 0}, "");
          run<NTL, 0, NTL, SC1 | NTL, 0, 0, 0, 4>(Cfg{m, wpc, lag, 0}, "");
          run<SC1 | 1, 0, NTL, SC1, 0, 0, 0, 4>(Cfg{m, wpc, lag, 0}, "");
          run<NTL | 1, 0, NTL, NTL, 0, 0, 0, 4>(Cfg{m, wpc, lag, 0}, "");
        }
    }
    return 0;
  }
  for(int m : {16, 17}) {
    printf("# N = 2^%d: separate launches per pass (mode 1 then 2) -- the two-launch baseline of this tiling\n", m);
    run<NTL, 0, NTL, 0, 0, 0, 0, 4>(Cfg{m, 4, 0, 1}, "two launches");
    run<NTL, 0, NTL, 0, 0, 1, 0, 4>(Cfg{m, 4, 0, 1}, "two launches");
    printf("# N = 2^%d: fused, XCD-local hand-off; memory only\n", m);
    for(int wpc : {2, 3, 4})
      for(int lag : {1, 2, 3, 4, 6}) run<NTL, 0, NTL, 0, 0, 0, 0, 4>(Cfg{m, wpc, lag, 0}, "");
    puts("# final stores write-through (sc1: the line is dropped from the L2) / nt; row loads sc1");
    for(int lag : {2, 3, 4}) {
      run<NTL, 0, NTL, SC1, 0, 0, 0, 4>(Cfg{m, 4, lag, 0}, "");
      run<NTL, 0, NTL, NTL, 0, 0, 0, 4>(Cfg{m, 4, lag, 0}, "");
      run<NTL, 0, SC1, 0, 0, 0, 0, 4>(Cfg{m, 4, lag, 0}, "");
      run<0, 0, SC1, 0, 0, 0, 0, 4>(Cfg{m, 4, lag, 0}, "");
    }
    puts("# + LDS exchange in both passes, + table reads in the row pass, + FP64 work (F FMAs per element and pass)");
    for(int lag : {2, 3, 4}) {
      run<NTL, 0, NTL, 0, 0, 1, 0, 4>(Cfg{m, 4, lag, 0}, "");
      run<NTL, 0, NTL, 0, 0, 1, 1, 4>(Cfg{m, 4, lag, 0}, "");
      run<NTL, 0, NTL, 0, 40, 1, 1, 4>(Cfg{m, 4, lag, 0}, "");
      run<NTL, 0, NTL, 0, 40, 1, 1, 4>(Cfg{m, 3, lag, 0}, "");
    }
  }
  return 0;
}
