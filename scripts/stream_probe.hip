// stream_probe.hip — how fast can the gallery access pattern of knn_scores_kernel be streamed through LDS-DMA, as a
// function of the ring geometry (bytes in flight per CU)?  No MFMA, no query tile: LDS-DMA + counted vmcnt + raw
// barriers + one ds_read_b128 per lane and slot.  Used to choose the ring of the round-2 score kernel.
//
// Pattern (as the real kernel): gallery [N][16896 B]; workgroup w owns rows [N*w/G, N*(w+1)/G); K-step ks reads bytes
// [128 ks, 128 ks + 128) of every owned row; a K-step is cut into slots of SR rows; slot = SR/8 groups of 8 rows, one
// 1-KiB wave-instruction per group (lane l: row l>>3, 16-B chunk l&7).  Ring of R slots, R-1 in flight.
//
// build: hipcc -O3 --offload-arch=gfx950 scripts/stream_probe.hip -o scripts/stream_probe.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ROW_BYTES = 16896;     // 8448 bf16
constexpr int NK = ROW_BYTES / 128;

template <int AUX>
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, AUX);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// NW waves, slot of SR rows (SR/8 groups, (SR/8)/NW per wave: must divide), ring of R slots.
template <int NW, int SR, int R, int AUX, int WGPC>
__global__ __launch_bounds__(NW * 64, (NW * WGPC + 3) / 4) void probe(const char* __restrict__ G, int N, float* __restrict__ sink) {
  constexpr int SG = SR / 8, LPW = SG / NW, SLOT = SR * 128;
  static_assert(SG % NW == 0, "uniform loads per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long long r0 = (long long)N * blockIdx.x / gridDim.x, r1 = (long long)N * (blockIdx.x + 1) / gridDim.x;
  const int rows = (int)(r1 - r0);
  if (rows <= 0) return;
  const int spk = (rows + SR - 1) / SR;            // slots per K-step
  const int total = spk * NK;
  const int sw = lane & 7;
  auto issue = [&](int s) {
    const int ks = s / spk, part = s - ks * spk;
    char* base = smem + (s % R) * SLOT;
#pragma unroll
    for (int i = 0; i < LPW; ++i) {
      const int g = wave + NW * i;
      const int tr = part * SR + g * 8 + (lane >> 3);
      const long long r = r0 + min(tr, rows - 1);
      glds16<AUX>(G + r * ROW_BYTES + ks * 128 + ((sw ^ ((tr >> 1) & 7)) << 4), base + g * 1024);
    }
  };
  float acc = 0.f;
  for (int p = 0; p < R - 1 && p < total; ++p) issue(p);
  for (int s = 0; s < total; ++s) {
    if (s + R - 2 < total) wait_vm<(R - 2) * LPW>(); else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (s + R - 1 < total) issue(s + R - 1);
    const float4 v = *reinterpret_cast<const float4*>(smem + (s % R) * SLOT + ((threadIdx.x * 16) % SLOT));
    acc += v.x + v.w;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 123.456f) sink[blockIdx.x] = acc;
}

template <int NW, int SR, int R, int AUX, int WGPC>
static void run(const char* name, const char* G, int N, float* sink, int cus) {
  const size_t lds = (size_t)R * SR * 128;
  if (lds * WGPC > 160 * 1024) { printf("%-34s skipped (LDS %zu x %d)\n", name, lds, WGPC); return; }
  auto k = probe<NW, SR, R, AUX, WGPC>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, NW * 64, lds));
  const int grid = cus * WGPC;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> ms;
  for (int it = 0; it < 7; ++it) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(grid), dim3(NW * 64), lds, 0, G, N, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1));
    if (it > 0) ms.push_back(t);
  }
  CK(hipGetLastError());
  std::sort(ms.begin(), ms.end());
  const double bytes = (double)N * ROW_BYTES;
  printf("%-34s occ/CU %d  ring %3zu KB/WG  in flight <= %3zu KB/CU  median %7.1f us  min %7.1f us  -> %6.0f GB/s (median)\n",
         name, occ, lds / 1024, (size_t)(R - 1) * SR * 128 * WGPC / 1024, ms[ms.size() / 2] * 1e3, ms[0] * 1e3,
         bytes / (ms[ms.size() / 2] * 1e-3) / 1e9);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 98304;
  int dev = 0, cus = 0;
  CK(hipGetDevice(&dev));
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  char* G; float* sink;
  CK(hipMalloc(&G, (size_t)N * ROW_BYTES));
  CK(hipMalloc(&sink, 4096 * sizeof(float)));
  CK(hipMemset(G, 1, (size_t)N * ROW_BYTES));
  printf("N = %d rows x %d B = %.3f GB, %d CUs\n", N, ROW_BYTES, (double)N * ROW_BYTES / 1e9, cus);
#define RUN(NW, SR, R, AUX, W) run<NW, SR, R, AUX, W>("NW" #NW " SR" #SR " R" #R " aux" #AUX " wgpc" #W, G, N, sink, cus)
  // the round-1 geometry: 2 workgroups per CU, one tile per K-step, 2-slot ring (1 in flight per workgroup)
  RUN(4, 192, 2, 0, 2); RUN(4, 192, 2, 2, 2);
  RUN(4, 192, 3, 0, 2); RUN(4, 192, 3, 2, 2);
  // finer slots, 2 workgroups per CU
  RUN(4, 96, 2, 0, 2); RUN(4, 96, 3, 0, 2); RUN(4, 96, 4, 0, 2); RUN(4, 96, 6, 0, 2); RUN(4, 96, 6, 2, 2);
  RUN(4, 64, 4, 0, 2); RUN(4, 64, 8, 0, 2); RUN(4, 64, 8, 2, 2);
  // one 8-wave workgroup per CU (384 rows each at N = 98304)
  RUN(8, 384, 2, 0, 1); RUN(8, 384, 3, 0, 1);
  RUN(8, 128, 3, 0, 1); RUN(8, 128, 4, 0, 1); RUN(8, 128, 6, 0, 1); RUN(8, 128, 8, 0, 1); RUN(8, 128, 9, 0, 1);
  RUN(8, 128, 6, 2, 1); RUN(8, 128, 9, 2, 1);
  RUN(8, 64, 8, 0, 1); RUN(8, 64, 16, 0, 1); RUN(8, 64, 16, 2, 1);
  // 16 waves
  RUN(16, 128, 9, 0, 1); RUN(16, 384, 3, 0, 1);
  return 0;
}
