// cg_wave.hpp -- Wave-level helpers: wave-scope sync, ballots, small reductions, SWAR on byte planes.
// Part of cygym_hip.hip (included inside its anonymous namespace, in order); not a standalone header.
#ifndef CG_WAVE_HPP
#define CG_WAVE_HPP

// ---------------- wave-level helpers ----------------
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int below(uint64_t m) {  // set bits of m below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int wave_or(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
  return v;
}
// sum of a small per-lane count (< 2^bits) with `bits` ballots (SALU popcounts; no LDS permutes)
__device__ __forceinline__ int wave_sum_bits(int v, int bits) {
  int t = 0;
  for (int b = 0; b < bits; ++b) t += __popcll(ballot((v >> b) & 1)) << b;
  return t;
}
__device__ __forceinline__ int wave_sum(int v) {   // general (permute-based); rare paths only
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int n = __shfl_up(v, o);
    if (lane >= o) v += n;
  }
  return v;
}
__device__ __forceinline__ int nth_bit(uint64_t m, int r) {  // position of the r-th set bit (uniform)
  for (int i = 0; i < r; ++i) m &= m - 1;
  return __builtin_ctzll(m);
}
__device__ __forceinline__ int nth_bit32(uint32_t m, int r) {
  for (int i = 0; i < r; ++i) m &= m - 1;
  return __builtin_ctz(m);
}

// cg_cdf_lookup (cygym_spec.h) over a table that may live in any address space (kernarg or the constant-space copy)
template <class Tab>
__device__ __forceinline__ int cdf_lookup(uint32_t u, const Tab& thr, int n) {
  int k = 0;
  for (int j = 0; j < n; ++j) k += ((uint64_t)u >= thr[j]) ? 1 : 0;
  return k;
}

// ---- SWAR on 4 device bytes per 32-bit word ----
#define ONES 0x01010101u
__device__ __forceinline__ uint32_t nz01(uint32_t b) {   // 0x01 in every byte of b that is non-zero
  return ((b | ((b & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & ONES;
}

#endif  // CG_WAVE_HPP
