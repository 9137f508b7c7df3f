// tools/memset_graph_repro.hip -- library-independent reproducer for the round-5 finding "a memset node captured into a HIP graph did
// not reliably precede the next kernel node's first reads" (profiles/r05/graph_replay_control_block_clear.txt; review r05, item 7).
//
//   hipcc -O2 --offload-arch=gfx950 -o build/memset_graph_repro tools/memset_graph_repro.hip && build/memset_graph_repro
//
// No library code.  A control block of counters is zeroed by a captured hipMemsetAsync and then used by a kernel the way the XCD-local
// kernels use theirs: every workgroup (a) reads ITS OWN completion word at the very start with three kinds of load -- plain, sc0 sc1
// (what the library's first-pass inputs use), agent-scope atomic -- (b) takes a ticket from its XCD's queue head with an agent-scope
// atomic add, (c) spins for a few microseconds (work), (d) bumps its completion word.  The block starts every replay as the previous
// replay left it (heads = workgroups per XCD, completion words = 1), so any word that is not 0 at (a), any ticket outside
// [0, workgroups on that XCD), or a final head / completion word that is not exactly what ONE launch adds is a memset that had not
// (fully) taken effect when the kernel ran -- or that ran in the middle of it.
// Variants: what clears the block (memset node / a clearing kernel / memset behind another kernel node), what happens between two
// replays (nothing / 2 GiB of fills on the same stream / on another stream / a direct kernel on the same stream), graph or plain stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                   \
  do {                                                                                          \
    hipError_t e_ = (x);                                                                        \
    if(e_ != hipSuccess) {                                                                      \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));         \
      exit(2);                                                                                  \
    }                                                                                           \
  } while(0)

constexpr int kGrid = 2048, kHeads = 8 * 32, kWords = kHeads + kGrid;
// second part (torch-like replay): the library's own block size at the failing call -- sizeof(TeamCtl) + 96 counters = 2052 + 384
// = 2436 bytes (not a multiple of 16: a fill of that size is split into an aligned body and a tail by the runtime)
constexpr int kGridSmall = 96, kWordsSmall = 513 + kGridSmall;

struct Obs {
  unsigned first[3]; // own completion word at kernel start: plain, sc0 sc1, atomic
  unsigned ticket, xcd;
};

__device__ __forceinline__ unsigned xcc_id()
{
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
  return v & 7u;
}

__global__ void __launch_bounds__(256) user_kernel(unsigned *ctl, Obs *obs, unsigned spin_clocks, int heads = kHeads)
{
  const unsigned wg = blockIdx.x;
  if(threadIdx.x == 0) {
    unsigned *mine = ctl + heads + wg;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(mine, 0, 4, 0x00020000);
    Obs            o;
    o.first[0] = *(volatile unsigned *)mine;
    o.first[1] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(r, 0, 0, 17); // sc0 sc1
    o.first[2] = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    o.xcd      = xcc_id();
    o.ticket   = __hip_atomic_fetch_add(ctl + o.xcd * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_readcyclecounter();
    while(__builtin_readcyclecounter() - t0 < spin_clocks) __builtin_amdgcn_s_sleep(8);
    __hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    obs[wg] = o;
  }
}

__global__ void clear_kernel(unsigned *w, size_t n)
{
  for(size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) w[i] = 0;
}
__global__ void touch_kernel(unsigned *w) { if(threadIdx.x == 0 && blockIdx.x == 0) w[0] += 1; }

enum Clear { MEMSET, KERNEL, MEMSET_BEHIND_KERNEL };
enum Between { NONE, FLUSH_SAME, FLUSH_OTHER, DIRECT_SAME };
static const char *clear_name[]   = {"memset node at the root", "clearing kernel", "memset node behind a kernel node"};
static const char *between_name[] = {"nothing", "2 GiB of fills on the capture stream", "2 GiB of fills on another stream, synchronised",
                                     "a direct launch of the same kernel on the capture stream (own block)"};

int main(int argc, char **argv)
{
  const int replays = argc > 1 ? atoi(argv[1]) : 12;
  int       dev     = 0;
  CK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, dev));
  int rt = 0;
  CK(hipRuntimeGetVersion(&rt));
  printf("# device %s, %d CUs, HIP runtime %d\n", prop.gcnArchName, prop.multiProcessorCount, rt);
  unsigned *ctl, *ctl2, *scratch, *dummy;
  Obs *     obs, *obs2;
  const size_t flush_bytes = 2ull << 30;
  CK(hipMalloc(&ctl, kWords * 4));
  CK(hipMalloc(&ctl2, kWords * 4));
  CK(hipMalloc(&obs, kGrid * sizeof(Obs)));
  CK(hipMalloc(&obs2, kGrid * sizeof(Obs)));
  CK(hipMalloc(&scratch, flush_bytes));
  CK(hipMalloc(&dummy, 256));
  hipStream_t st, other;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking));
  std::vector<unsigned> h(kWords);
  std::vector<Obs>      ho(kGrid);
  int                   total_bad = 0;
  for(int use_graph = 1; use_graph >= 0; use_graph--) {
    for(int clr = MEMSET; clr <= MEMSET_BEHIND_KERNEL; clr++) {
      for(int btw = NONE; btw <= DIRECT_SAME; btw++) {
        for(unsigned spin : {2000u, 40000u}) {
          CK(hipMemset(ctl, 0, kWords * 4));
          CK(hipMemset(ctl2, 0, kWords * 4));
          CK(hipDeviceSynchronize());
          auto enqueue = [&]() {
            if(clr == MEMSET_BEHIND_KERNEL) touch_kernel<<<1, 64, 0, st>>>(dummy);
            if(clr == KERNEL) clear_kernel<<<4, 256, 0, st>>>(ctl, (size_t)kWords);
            else CK(hipMemsetAsync(ctl, 0, kWords * 4, st));
            user_kernel<<<kGrid, 256, 0, st>>>(ctl, obs, spin);
          };
          hipGraphExec_t exec = nullptr;
          hipGraph_t     graph = nullptr;
          if(use_graph) {
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            enqueue();
            CK(hipStreamEndCapture(st, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
          }
          int bad_replays = 0, stale[3] = {0, 0, 0}, bad_ticket = 0, bad_final = 0, first_bad = -1;
          for(int r = 0; r < replays; r++) {
            if(use_graph) CK(hipGraphLaunch(exec, st));
            else enqueue();
            CK(hipStreamSynchronize(st));
            CK(hipMemcpy(h.data(), ctl, kWords * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(ho.data(), obs, kGrid * sizeof(Obs), hipMemcpyDeviceToHost));
            unsigned per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for(int w = 0; w < kGrid; w++) per_xcd[ho[w].xcd & 7]++;
            int bad = 0;
            for(int w = 0; w < kGrid; w++) {
              for(int k = 0; k < 3; k++)
                if(ho[w].first[k] != 0) stale[k]++, bad = 1;
              if(ho[w].ticket >= per_xcd[ho[w].xcd & 7]) bad_ticket++, bad = 1;
              if(h[kHeads + w] != 1) bad_final++, bad = 1;
            }
            for(int x = 0; x < 8; x++)
              if(h[x * 32] != per_xcd[x]) bad_final++, bad = 1;
            if(bad && first_bad < 0) first_bad = r;
            bad_replays += bad;
            // between two replays
            if(btw == FLUSH_SAME) CK(hipMemsetAsync(scratch, r & 0xff, flush_bytes, st));
            if(btw == FLUSH_OTHER) {
              CK(hipMemsetAsync(scratch, r & 0xff, flush_bytes, other));
              CK(hipStreamSynchronize(other));
            }
            if(btw == DIRECT_SAME) {
              clear_kernel<<<4, 256, 0, st>>>(ctl2, (size_t)kWords);
              user_kernel<<<kGrid, 256, 0, st>>>(ctl2, obs2, spin);
            }
          }
          CK(hipStreamSynchronize(st));
          printf("%-6s clear=%-34s between=%-70s spin=%-6u : %2d of %d replays wrong (first %d)  stale-at-start plain/sc0sc1/atomic %d/%d/%d  bad tickets %d  bad final words %d\n",
                 use_graph ? "graph" : "stream", clear_name[clr], between_name[btw], spin, bad_replays, replays, first_bad, stale[0], stale[1],
                 stale[2], bad_ticket, bad_final);
          total_bad += bad_replays;
          if(exec) CK(hipGraphExecDestroy(exec));
          if(graph) CK(hipGraphDestroy(graph));
        }
      }
    }
  }
  printf("# total wrong replays (part 1): %d\n", total_bad);

  // ---- part 2: the replay pattern of the failing library test (tools/dbg_graph3.py drives the library through torch.cuda.CUDAGraph):
  // captured on a non-blocking side stream in GLOBAL capture mode, REPLAYED ON THE NULL STREAM, inputs copied host -> device on the null
  // stream before every replay, a 2 GiB fill on the null stream + device synchronisation between replays; block of 2436 bytes.
  {
    unsigned *c3, *inbuf;
    Obs *     o3;
    CK(hipMalloc(&c3, kWordsSmall * 4));
    CK(hipMalloc(&o3, kGridSmall * sizeof(Obs)));
    CK(hipMalloc(&inbuf, 64 << 20));
    std::vector<unsigned> hin((64 << 20) / 4, 7u), h3(kWordsSmall);
    std::vector<Obs>      ho3(kGridSmall);
    for(int replay_on_null = 0; replay_on_null <= 1; replay_on_null++) {
      for(int clr = MEMSET; clr <= KERNEL; clr++) {
        for(int flush = 0; flush <= 1; flush++) {
          for(int nodes_before = 0; nodes_before <= 1; nodes_before++) {
            CK(hipMemset(c3, 0, kWordsSmall * 4));
            CK(hipDeviceSynchronize());
            auto enqueue = [&]() {
              if(nodes_before) touch_kernel<<<1, 64, 0, st>>>(dummy);
              if(clr == KERNEL) clear_kernel<<<4, 256, 0, st>>>(c3, (size_t)kWordsSmall);
              else CK(hipMemsetAsync(c3, 0, kWordsSmall * 4, st));
              user_kernel<<<kGridSmall, 256, 0, st>>>(c3, o3, 40000u, 513);
              // a second "operation" behind it, as in the library test (its own clear in front)
              if(clr == KERNEL) clear_kernel<<<4, 256, 0, st>>>(ctl2, (size_t)kWords);
              else CK(hipMemsetAsync(ctl2, 0, kWords * 4, st));
              user_kernel<<<kGrid, 256, 0, st>>>(ctl2, obs2, 2000u);
            };
            enqueue(); // the direct call in front of the capture (the library's rule: the blocks exist before capturing)
            CK(hipStreamSynchronize(st));
            hipGraph_t     graph;
            hipGraphExec_t exec;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
            enqueue();
            CK(hipStreamEndCapture(st, &graph));
            CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            int bad_replays = 0, stale = 0, bad_ticket = 0, bad_final = 0, first_bad = -1;
            for(int r = 0; r < replays; r++) {
              CK(hipMemcpy(inbuf, hin.data(), 64 << 20, hipMemcpyHostToDevice)); // "ta.copy_(...)": null stream
              CK(hipGraphLaunch(exec, replay_on_null ? (hipStream_t) nullptr : st));
              CK(hipDeviceSynchronize());
              CK(hipMemcpy(h3.data(), c3, kWordsSmall * 4, hipMemcpyDeviceToHost));
              CK(hipMemcpy(ho3.data(), o3, kGridSmall * sizeof(Obs), hipMemcpyDeviceToHost));
              unsigned per_xcd[8] = {0, 0, 0, 0, 0, 0, 0, 0};
              for(int w = 0; w < kGridSmall; w++) per_xcd[ho3[w].xcd & 7]++;
              int bad = 0;
              for(int w = 0; w < kGridSmall; w++) {
                for(int k = 0; k < 3; k++)
                  if(ho3[w].first[k] != 0) stale++, bad = 1;
                if(ho3[w].ticket >= per_xcd[ho3[w].xcd & 7]) bad_ticket++, bad = 1;
                if(h3[513 + w] != 1) bad_final++, bad = 1;
              }
              for(int x = 0; x < 8; x++)
                if(h3[x * 32] != per_xcd[x]) bad_final++, bad = 1;
              if(bad && first_bad < 0) first_bad = r;
              bad_replays += bad;
              if(flush) {
                CK(hipMemsetAsync(scratch, r & 0xff, flush_bytes, nullptr));
                CK(hipDeviceSynchronize());
              }
            }
            printf("torch-like: replay on %-12s clear=%-16s flush between=%d kernel node in front=%d : %2d of %d replays wrong (first %d)  stale-at-start %d  bad tickets %d  bad final words %d\n",
                   replay_on_null ? "NULL stream" : "capture strm", clr == KERNEL ? "clearing kernel" : "memset node", flush, nodes_before, bad_replays,
                   replays, first_bad, stale, bad_ticket, bad_final);
            total_bad += bad_replays;
            CK(hipGraphExecDestroy(exec));
            CK(hipGraphDestroy(graph));
          }
        }
      }
    }
  }
  printf("# total wrong replays: %d\n", total_bad);
  return 0;
}
