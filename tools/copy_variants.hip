// tools/copy_variants.hip -- out-of-place copies of a 8 GiB slab in several shapes: which one is the fair "copy ceiling" of this part?
// (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy; bench.py's ntt_copy_probe is variant 0)
//   hipcc --offload-arch=gfx950 -O3 -o build/copy_variants tools/copy_variants.hip && ./build/copy_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int v4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_stride(v4 *d, const v4 *s, uint64_t n)
{
  for(uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) d[i] = s[i];
}
__global__ void __launch_bounds__(256) k_one(v4 *d, const v4 *s, uint64_t n)
{
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if(i < n) d[i] = s[i];
}
template <int U, bool NT> __global__ void __launch_bounds__(256) k_unroll(v4 *d, const v4 *s, uint64_t n)
{
  // a workgroup copies U * 256 consecutive 16-byte words per iteration: U loads in flight, then U stores
  for(uint64_t base = (uint64_t)blockIdx.x * (256 * U); base < n; base += (uint64_t)gridDim.x * (256 * U)) {
    v4 r[U];
#pragma unroll
    for(int u = 0; u < U; u++) {
      const uint64_t i = base + (uint64_t)u * 256 + threadIdx.x;
      if(i < n) r[u] = NT ? __builtin_nontemporal_load(s + i) : s[i];
    }
#pragma unroll
    for(int u = 0; u < U; u++) {
      const uint64_t i = base + (uint64_t)u * 256 + threadIdx.x;
      if(i < n) { if(NT) __builtin_nontemporal_store(r[u], d + i); else d[i] = r[u]; }
    }
  }
}
template <class F> double run(const char *name, F launch, uint64_t bytes)
{
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for(int i = 0; i < 3; i++) launch();
  hipEventRecord(e0);
  const int reps = 10;
  for(int i = 0; i < reps; i++) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double gbs = 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
  printf("%-44s %8.1f GB/s (read + written)  %.3f of 8 TB/s\n", name, gbs, gbs / 8000.0);
  return gbs;
}
int main()
{
  const uint64_t bytes = 8ull << 30, n = bytes / 16;
  v4 *s, *d;
  if(hipMalloc(&s, bytes) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
  for(int rep = 0; rep < 2; rep++) {
    run("grid-stride, 65536 workgroups (bench.py's probe)", [&] { hipLaunchKernelGGL(k_stride, dim3(65536), dim3(256), 0, 0, d, s, n); }, bytes);
    run("grid-stride, 2048 workgroups", [&] { hipLaunchKernelGGL(k_stride, dim3(2048), dim3(256), 0, 0, d, s, n); }, bytes);
    run("one word per thread, no loop", [&] { hipLaunchKernelGGL(k_one, dim3((unsigned)(n / 256)), dim3(256), 0, 0, d, s, n); }, bytes);
    run("4 loads then 4 stores, 2048 workgroups", [&] { hipLaunchKernelGGL((k_unroll<4, false>), dim3(2048), dim3(256), 0, 0, d, s, n); }, bytes);
    run("8 loads then 8 stores, 2048 workgroups", [&] { hipLaunchKernelGGL((k_unroll<8, false>), dim3(2048), dim3(256), 0, 0, d, s, n); }, bytes);
    run("8 loads then 8 stores, 8192 workgroups", [&] { hipLaunchKernelGGL((k_unroll<8, false>), dim3(8192), dim3(256), 0, 0, d, s, n); }, bytes);
    run("8 x nt loads then 8 x nt stores, 2048 workgroups", [&] { hipLaunchKernelGGL((k_unroll<8, true>), dim3(2048), dim3(256), 0, 0, d, s, n); }, bytes);
    run("4 x nt loads then 4 x nt stores, 8192 workgroups", [&] { hipLaunchKernelGGL((k_unroll<4, true>), dim3(8192), dim3(256), 0, 0, d, s, n); }, bytes);
  }
  return 0;
}
