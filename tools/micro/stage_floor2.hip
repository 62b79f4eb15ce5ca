// Probe 2: does the observation write of a launch (6 KB per wave, write-only) push the env state (3 KB per wave, written back with
// plain stores) out of the L2 the next launch reads it from?  Same shape as stage_floor.hip; OBS selects how the 6 KB are stored.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stage_floor2.hip -o tools/micro/stage_floor2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
typedef unsigned v4 __attribute__((ext_vector_type(4)));
template <int OBS>
__global__ __launch_bounds__(1024) void probe(uint4* state, const uint4* shared, uint4* obs, unsigned long long* out, unsigned long long* out2) {
  extern __shared__ uint4 lds[];
  constexpr int items = 3, OI = 6;
  unsigned long long t0, t1, t2;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 16 + wave;
  uint4* mine = state + (size_t)gw * items * 64;
  uint4* dst = lds + 1024 + wave * items * 64;
  __builtin_amdgcn_global_load_lds((const AS1 void*)(shared + threadIdx.x), (AS3 void*)(lds + wave * 64), 16, 0, 0);
  for (int j = 0; j < items; ++j)
    __builtin_amdgcn_global_load_lds((const AS1 void*)(mine + j * 64 + lane), (AS3 void*)(dst + j * 64), 16, 0, 0);
  __syncthreads();
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  uint4 v = dst[lane];
  v.x += 1;
  v4 w = {v.x, v.y, v.z, v.w};
  uint4* ob = obs + (size_t)gw * OI * 64;
  for (int j = 0; j < OI; ++j) {
    const unsigned long long pa = (unsigned long long)(ob + j * 64 + lane);
    if (OBS == 1) ob[j * 64 + lane] = v;
    if (OBS == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(pa), "v"(w) : "memory");
    if (OBS == 3) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(pa), "v"(w) : "memory");
    if (OBS == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(pa), "v"(w) : "memory");
    if (OBS == 5) asm volatile("global_store_dwordx4 %0, %1, off nt sc1" :: "v"(pa), "v"(w) : "memory");
    if (OBS == 6) asm volatile("global_store_dwordx4 %0, %1, off nt sc0 sc1" :: "v"(pa), "v"(w) : "memory");
  }
  for (int j = 0; j < items; ++j) mine[j * 64 + lane] = v;
  asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2) :: "memory");
  if (lane == 0) { out[gw] = (t1 - t0) + (v.y & 1u); out2[gw] = t2 - t1; }
}
template <int OBS>
static void run(const char* name, uint4* st, uint4* sh, uint4* ob, unsigned long long* o, unsigned long long* o2) {
  const int waves = 4096;
  std::vector<unsigned long long> h(waves), h2(waves);
  const size_t lds = (1024 + 16 * 3 * 64) * 16;
  hipFuncSetAttribute((const void*)probe<OBS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL((probe<OBS>), dim3(256), dim3(1024), lds, 0, st, sh, ob, o, o2);
    hipEventRecord(e0, 0);
    for (int k = 0; k < 50; ++k) hipLaunchKernelGGL((probe<OBS>), dim3(256), dim3(1024), lds, 0, st, sh, ob, o, o2);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), o, waves * 8, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), o2, waves * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end()); std::sort(h2.begin(), h2.end());
    printf("%-34s stage p50 %llu p90 %llu p99 %llu | stores drained p50 %llu p99 %llu cycles | launch period %.2f us\n", name, h[waves / 2], h[waves * 9 / 10], h[waves * 99 / 100],
           h2[waves / 2], h2[waves * 99 / 100], ms * 1000.f / 50.f);
  }
}
int main() {
  uint4 *st, *sh, *ob; unsigned long long *o, *o2;
  hipMalloc(&st, (size_t)4096 * 3 * 1024); hipMemset(st, 0, (size_t)4096 * 3 * 1024);
  hipMalloc(&ob, (size_t)4096 * 6 * 1024);
  hipMalloc(&sh, 16384); hipMemset(sh, 0, 16384);
  hipMalloc(&o, 4096 * 8); hipMalloc(&o2, 4096 * 8);
  run<0>("no observation", st, sh, ob, o, o2);
  run<1>("observation: plain stores", st, sh, ob, o, o2);
  run<2>("observation: sc1", st, sh, ob, o, o2);
  run<3>("observation: nt", st, sh, ob, o, o2);
  run<4>("observation: sc0 sc1", st, sh, ob, o, o2);
  run<5>("observation: nt sc1", st, sh, ob, o, o2);
  run<6>("observation: nt sc0 sc1", st, sh, ob, o, o2);
  return 0;
}
