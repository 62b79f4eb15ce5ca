// cg_wave.hpp -- Wave-level helpers: wave-scope sync, ballots, small reductions, SWAR on byte planes.
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_WAVE_HPP
#define CG_WAVE_HPP

// ---------------- wave-level helpers ----------------
// The value is complete here: a load feeding it cannot be sunk towards a later conditional use.  Staged code (the
// loads of a group first, then the work on them) relies on it: without the pin the compiler moves a load whose only
// use is conditional into that branch, next to its s_waitcnt, and a group of independent LDS reads turns into a chain.
#define PIN(x) asm volatile("" :: "v"(x))
__device__ __forceinline__ void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// The env's 16 + 3 scalars as lanes of ONE vector register (lane i = ienv word i, lanes 16..21 = the three fenv doubles as word
// pairs: the layout of the prologue's gathered load): read with v_readlane where used, updated with v_writelane, written back by one
// store.  As 22 SGPRs for the whole kernel they were read out behind the barrier and immediately spilled into another register's
// lanes (two instructions per scalar before the first use, a restore per use).
extern "C" __device__ int cg_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");   // v_writelane_b32 (this toolchain's clang has no builtin for it)
struct LaneI {
  uint32_t& hv; int i;
  __device__ __forceinline__ operator int32_t() const { return __builtin_amdgcn_readlane((int)hv, i); }
  __device__ __forceinline__ void set(int32_t v) const { hv = (uint32_t)cg_writelane(__builtin_amdgcn_readfirstlane(v), i, (int)hv); }
  __device__ __forceinline__ const LaneI& operator=(int32_t v) const { set(v); return *this; }
  __device__ __forceinline__ const LaneI& operator+=(int32_t v) const { set((int32_t)*this + v); return *this; }
  __device__ __forceinline__ const LaneI& operator|=(int32_t v) const { set((int32_t)*this | v); return *this; }
  __device__ __forceinline__ const LaneI& operator&=(int32_t v) const { set((int32_t)*this & v); return *this; }
};
struct EnvI {
  uint32_t& hv;
  __device__ __forceinline__ LaneI operator[](int i) const { return LaneI{hv, i}; }
};
struct LaneD {
  uint32_t& hv; int i;   // low word in lane i, high word in lane i + 1
  __device__ __forceinline__ operator double() const { return __hiloint2double(__builtin_amdgcn_readlane((int)hv, i + 1), __builtin_amdgcn_readlane((int)hv, i)); }
  __device__ __forceinline__ void set(double v) const {
    hv = (uint32_t)cg_writelane(__builtin_amdgcn_readfirstlane(__double2loint(v)), i, (int)hv);
    hv = (uint32_t)cg_writelane(__builtin_amdgcn_readfirstlane(__double2hiint(v)), i + 1, (int)hv);
  }
  __device__ __forceinline__ const LaneD& operator=(double v) const { set(v); return *this; }
  __device__ __forceinline__ const LaneD& operator+=(double v) const { set((double)*this + v); return *this; }
};
struct EnvF {
  uint32_t& hv;
  __device__ __forceinline__ LaneD operator[](int i) const { return LaneD{hv, 16 + 2 * i}; }
};
template <bool LS> __device__ __forceinline__ auto scalars_i(uint32_t& hv, int32_t* a) { if constexpr (LS) return EnvI{hv}; else return a; }
template <bool LS> __device__ __forceinline__ auto scalars_f(uint32_t& hv, double* a) { if constexpr (LS) return EnvF{hv}; else return a; }
__device__ __forceinline__ uint64_t ballot(bool p) { return __ballot(p); }
__device__ __forceinline__ int below(uint64_t m) {  // set bits of m below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
// Wave-wide inclusive scans on the VALU's DPP path (no LDS permutes): prefix within each row of 16 lanes with
// row_shr 1/2/4/8, then lane 15 of rows 0/2 into rows 1/3 (row_bcast15) and lane 31 into rows 2/3 (row_bcast31).
// Every lane must be active (all call sites are in wave-uniform control flow).
#define CG_DPP(v, ctrl, rmask) __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rmask), 0xf, false)
__device__ __forceinline__ int dpp_scan_add(int v) {
  v += CG_DPP(v, 0x111, 0xf); v += CG_DPP(v, 0x112, 0xf); v += CG_DPP(v, 0x114, 0xf); v += CG_DPP(v, 0x118, 0xf);
  v += CG_DPP(v, 0x142, 0xa); v += CG_DPP(v, 0x143, 0xc);
  return v;
}
__device__ __forceinline__ int dpp_scan_or(int v) {
  v |= CG_DPP(v, 0x111, 0xf); v |= CG_DPP(v, 0x112, 0xf); v |= CG_DPP(v, 0x114, 0xf); v |= CG_DPP(v, 0x118, 0xf);
  v |= CG_DPP(v, 0x142, 0xa); v |= CG_DPP(v, 0x143, 0xc);
  return v;
}
__device__ __forceinline__ int wave_or(int v) { return __builtin_amdgcn_readlane(dpp_scan_or(v), 63); }
// sum of a small per-lane count (< 2^bits) with `bits` ballots (SALU popcounts; no LDS permutes)
__device__ __forceinline__ int wave_sum_bits(int v, int bits) {
  int t = 0;
  for (int b = 0; b < bits; ++b) t += __popcll(ballot((v >> b) & 1)) << b;
  return t;
}
__device__ __forceinline__ int wave_sum(int v) { return __builtin_amdgcn_readlane(dpp_scan_add(v), 63); }
__device__ __forceinline__ int wave_incl_scan(int v, int lane) { (void)lane; return dpp_scan_add(v); }
__device__ __forceinline__ int nth_bit(uint64_t m, int r) {  // position of the r-th set bit (uniform)
  for (int i = 0; i < r; ++i) m &= m - 1;
  return __builtin_ctzll(m);
}
__device__ __forceinline__ int nth_bit32(uint32_t m, int r) {
  for (int i = 0; i < r; ++i) m &= m - 1;
  return __builtin_ctz(m);
}
// the same by bisection on popcounts: a fixed ~25 instructions, where the loop above costs three per skipped bit and,
// run per lane, makes every lane wait for the largest r of the wave
__device__ __forceinline__ int nth_bit32_bisect(uint32_t m, int r) {
  int pos = 0;
  int c = __popc(m & 0xFFFFu);
  if (r >= c) { pos = 16; r -= c; m >>= 16; }
  c = __popc(m & 0xFFu);
  if (r >= c) { pos += 8; r -= c; m >>= 8; }
  c = __popc(m & 0xFu);
  if (r >= c) { pos += 4; r -= c; m >>= 4; }
  c = __popc(m & 0x3u);
  if (r >= c) { pos += 2; r -= c; m >>= 2; }
  return pos + ((r >= (int)(m & 1u)) ? 1 : 0);
}

// cg_cdf_lookup (cygym_spec.h) over a table that may live in any address space (kernarg or the constant-space copy)
template <class Tab>
__device__ __forceinline__ int cdf_lookup(uint32_t u, const Tab& thr, int n) {
  int k = 0;
  for (int j = 0; j < n; ++j) k += ((uint64_t)u >= thr[j]) ? 1 : 0;
  return k;
}

// x mod p for non-negative x and p > 0, without the ~35-instruction division sequence when p is a power of two
// (env._evolve_period is 2 in every reference configuration)
__device__ __forceinline__ int umod(int x, int p) {
  return ((p & (p - 1)) == 0) ? (x & (p - 1)) : (x % p);
}
// floor(sqrt(n) / 2) for 1 <= n < 2^22: float estimate, then an exact integer correction (the oracle counts up)
__device__ __forceinline__ int half_isqrt(int n) {
  int h = (int)(__builtin_sqrtf((float)n) * 0.5f);
  while (4 * (h + 1) * (h + 1) <= n) ++h;
  while (h > 0 && 4 * h * h > n) --h;
  return h;
}

// ---- SWAR on 4 device bytes per 32-bit word ----
#define ONES 0x01010101u
__device__ __forceinline__ uint32_t nz01(uint32_t b) {   // 0x01 in every byte of b that is non-zero
  return ((b | ((b & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & ONES;
}

#endif  // CG_WAVE_HPP
