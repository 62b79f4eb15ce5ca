// Probe: the floor of the tick kernel's staging phase.  4096 waves (256 workgroups x 16), each reads `bytes` of per-wave state
// plus a shared 16 KB block into LDS (global_load_lds), waits, passes the workgroup barrier: cycles from wave entry to behind
// the barrier.  Variants: the state was WRITTEN by the previous launch (as the env state is) or is never written (clean).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stage_floor.hip -o tools/micro/stage_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
#define AS1 __attribute__((address_space(1)))
#define AS3 __attribute__((address_space(3)))
template <bool WRITE, bool SC1>
__global__ __launch_bounds__(1024) void probe(uint4* state, const uint4* shared, unsigned long long* out, int items) {
  extern __shared__ uint4 lds[];
  unsigned long long t0, t1;
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
  if (WRITE) {
    v.x += 1;
    for (int j = 0; j < items; ++j) {
      if (SC1) { typedef unsigned v4 __attribute__((ext_vector_type(4))); v4 w = {v.x, v.y, v.z, v.w};
                 asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"((unsigned long long)(mine + j * 64 + lane)), "v"(w) : "memory"); }
      else mine[j * 64 + lane] = v;
    }
  }
  if (lane == 0) out[gw] = (t1 - t0) + (v.y & 1u);
}
template <bool WRITE, bool SC1>
static void run(const char* name, uint4* st, uint4* sh, unsigned long long* o, int items) {
  const int waves = 4096;
  std::vector<unsigned long long> h(waves);
  const size_t lds = (1024 + 16 * items * 64) * 16;
  hipFuncSetAttribute((const void*)probe<WRITE, SC1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 3; ++rep) {
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL((probe<WRITE, SC1>), dim3(256), dim3(1024), lds, 0, st, sh, o, items);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), o, waves * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-44s items %d: wave entry -> behind the barrier: p10 %llu  p50 %llu  p90 %llu  p99 %llu cycles\n", name, items, h[waves / 10], h[waves / 2], h[waves * 9 / 10], h[waves * 99 / 100]);
  }
}
int main() {
  uint4 *st, *sh; unsigned long long* o;
  const size_t bytes = (size_t)4096 * 8 * 1024;
  hipMalloc(&st, bytes); hipMemset(st, 0, bytes);
  hipMalloc(&sh, 16384); hipMemset(sh, 0, 16384);
  hipMalloc(&o, 4096 * 8);
  for (int items : {1, 3, 6}) {
    run<false, false>("clean state (never written)", st, sh, o, items);
    run<true, false>("state written by the previous launch", st, sh, o, items);
    run<true, true>("... written through (sc1)", st, sh, o, items);
  }
  return 0;
}
