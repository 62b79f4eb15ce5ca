// Probe: does kernarg preload remove the first scalar-cache round trip on this box?
// Two builds of the same kernel (with / without -mllvm -amdgpu-kernarg-preload-count=8): cycles from wave entry until a
// global load whose address comes from the first kernel argument has returned.   hipcc --offload-arch=gfx950 -O3 ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
__global__ void probe(const unsigned* data, unsigned long long* out, int stride, int pad0, long long pad1, long long pad2) {
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
  unsigned v = data[(blockIdx.x * blockDim.x + threadIdx.x) * stride];
  asm volatile("s_waitcnt vmcnt(0)" :: "v"(v) : "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  if ((threadIdx.x & 63) == 0) out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = (t1 - t0) + (v & 1u) + pad0 + pad1 + pad2;
}
int main() {
  const int blocks = 256, threads = 1024, waves = blocks * threads / 64;
  unsigned* d; unsigned long long* o;
  hipMalloc(&d, (size_t)blocks * threads * 16 * 4); hipMemset(d, 0, (size_t)blocks * threads * 16 * 4);
  hipMalloc(&o, waves * 8);
  std::vector<unsigned long long> h(waves);
  for (int rep = 0; rep < 6; ++rep) {
    for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), 0, 0, d, o, 16, 0, 0LL, 0LL);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), o, waves * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("wave entry -> first dependent global load back: p10 %llu  p50 %llu  p90 %llu cycles\n", h[waves / 10], h[waves / 2], h[waves * 9 / 10]);
  }
  return 0;
}
