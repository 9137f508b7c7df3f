// tools/hwid_probe.hip -- where do the workgroups of a 2^12-shaped launch (256 threads, 40 KB LDS: four per CU) land, and what does
// HW_REG_HW_ID's TG_ID field say about the slot a workgroup occupies on its CU?  (the start stagger of NTT_STAGGER builds keys on it)
//   hipcc --offload-arch=gfx950 -O2 -o build/hwid_probe tools/hwid_probe.hip && ./build/hwid_probe [workgroups]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void __launch_bounds__(256) probe(uint32_t *out, int spin)
{
  __shared__ uint64_t lds[5 * 1024];
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  lds[threadIdx.x] = hw;
  __syncthreads();
  uint64_t acc = 0;
  for(int i = 0; i < spin; i++) acc += lds[(threadIdx.x + i) & 1023] + clock64();   // stay resident long enough for the chip to fill
  if(threadIdx.x == 0) {
    out[2 * blockIdx.x]     = hw;
    out[2 * blockIdx.x + 1] = (xcc & 7u) | ((uint32_t)(acc & 1) << 31);
  }
}
int main(int argc, char **argv)
{
  const int wgs = argc > 1 ? atoi(argv[1]) : 1024;
  uint32_t *d;
  hipMalloc(&d, wgs * 8);
  hipLaunchKernelGGL(probe, dim3(wgs), dim3(256), 0, 0, d, 20000);
  std::vector<uint32_t> h(2 * wgs);
  hipMemcpy(h.data(), d, wgs * 8, hipMemcpyDeviceToHost);
  std::map<uint32_t, std::vector<int>> per_cu;   // (xcc, se, sh, cu) -> TG_IDs
  int tg_hist[16] = {0}, wave_hist[16] = {0};
  for(int i = 0; i < wgs; i++) {
    const uint32_t hw = h[2 * i], xcc = h[2 * i + 1] & 7u;
    const uint32_t wave = hw & 15u, cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u, tg = (hw >> 16) & 15u;
    per_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back((int)tg);
    tg_hist[tg]++, wave_hist[wave]++;
    if(i < 24) printf("wg %4d: xcc %u se %u sh %u cu %2u  wave %2u simd %u  tg %2u   (hw_id %08x)\n", i, xcc, se, sh, cu, wave, (hw >> 4) & 3u, tg, hw);
  }
  printf("%zu distinct (xcc, se, sh, cu) for %d workgroups\n", per_cu.size(), wgs);
  printf("TG_ID histogram:"); for(int i = 0; i < 16; i++) printf(" %d", tg_hist[i]); printf("\n");
  printf("WAVE_ID (wave 0) histogram:"); for(int i = 0; i < 16; i++) printf(" %d", wave_hist[i]); printf("\n");
  int shown = 0, distinct4 = 0;
  for(auto &kv : per_cu) {
    bool seen[4] = {false, false, false, false};
    for(int t : kv.second) seen[t & 3] = true;
    if(kv.second.size() == 4 && seen[0] && seen[1] && seen[2] && seen[3]) distinct4++;
    if(shown++ < 8) { printf("cu key %05x:", kv.first); for(int t : kv.second) printf(" tg %d", t); printf("\n"); }
  }
  printf("%d CUs hold four workgroups whose TG_ID & 3 are all different\n", distinct4);
  return 0;
}
