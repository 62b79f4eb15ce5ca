// cg_env.hpp -- Per-wave view of one environment in LDS and the list / bit-range helpers shared by the actions.
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_ENV_HPP
#define CG_ENV_HPP

// ---------------- per-wave environment view ----------------
struct Env {
  // LDS, planes at stride MS = round_up(M, 4); padding bytes of `flags` hold CG_F_NYA
  uint8_t *flags, *busy, *wl, *cby;
  uint32_t* scr;     // [2*Mp] scratch (8 bytes per device)
  uint32_t* blk;     // [EWp] blocked bit per out-slot
  uint32_t* bin;     // [EWp] the same bits in in-CSR entry order
  uint16_t* ring;    // [2*CG_LOG_RING]
  uint32_t* marks;   // [Mp/32 + 2]
  uint16_t* lsrc;    // [Mp] long-row sources of the spread
  int16_t* devl;     // [L] this tick's device lists (all groups, concatenated)
  // edges added by evolve_network (cygym_spec.h: extra-edge list), staged only when the env has any
  uint32_t* xk;      // [K] keys (u << 16 | v), ascending; the first x_cnt() are live
  uint32_t* xb;      // [KW] blocked bit per list entry
  uint64_t* xmo;     // [MC] devices with an added OUT edge (their rows are walked merged)
  uint64_t* xmi;     // [MC] devices with an added edge at either end
  int K;
  bool x_dirty;
  // shared LDS (topology)
  const uint16_t *optr, *ocol, *iptr_l;
  const uint8_t *dst, *vul, *nap;
  const float *osv, *ver, *ano;
  // in-CSR columns + slot<->entry maps: global memory (L2-resident blob), or LDS in the WIDE kernel; read by block/unblock only
  const uint16_t *icol_g, *ieid_g, *oeid_g;
  float4* obs_stage; // [192] 3 KB: the observation's LDS stage (WIDE kernel only, see write_obs_staged)
  uint8_t* stash;    // global [4][M] of this env
  // Device.compromised_by: staged in LDS like the other planes (`cby`), or -- run-time sizes, where the per-env LDS
  // footprint decides how many waves a CU holds -- left in global memory (`cby_g`, plane 3 of the env's live block):
  // only cleans, the stash actions and the domain-controller attribution touch it, a few lanes at a time.
  uint8_t* cby_g;    // non-null: comp_by lives in global memory (word-aligned: M % 4 == 0)
  // misc
  int M, MC, MS, lane, env;
  int cbits;         // bits needed for a per-lane device count
  uint32_t env_id, tick;
  uint64_t seed;
  int eflags;        // CG_I_FLAGS (uniform except BUSY_SAT, OR-reduced at write-back)
  bool blk_dirty, ring_dirty;
  bool multi;        // topology has duplicate (u,v) out-entries
  int log_total;

  __device__ __forceinline__ uint32_t draw(uint32_t site, uint32_t a, uint32_t b) const {
    return cg_draw(seed, env_id, tick, site, a, b);
  }
  __device__ __forceinline__ bool blocked(int slot) const { return (blk[slot >> 5] >> (slot & 31)) & 1u; }
  __device__ __forceinline__ int iptr(int d) const { return iptr_l[d]; }
  __device__ __forceinline__ void set_busy(int d, int v) {
    if (v > 255) { v = 255; eflags |= CG_E_BUSY_SAT; }
    busy[d] = (uint8_t)v;
  }
};

__device__ __forceinline__ void byte_or(uint8_t* base, int d, uint32_t bits) {
  atomicOr((unsigned int*)(base + (d & ~3)), bits << ((d & 3) * 8));
}
// comp_by accessors.  The global form goes through device-scope atomics / atomic loads: lanes of this wave read what other
// lanes wrote earlier in the tick, and plain vector loads may be served from a stale L1 line.
#ifndef CG_CBY_GLOBAL
#define CG_CBY_GLOBAL 1
#endif
__device__ __forceinline__ uint32_t cby_get(const Env& e, int d) {
  if (CG_CBY_GLOBAL && e.cby_g) {
    const uint32_t w = __hip_atomic_load((const uint32_t*)(e.cby_g + (d & ~3)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (w >> ((d & 3) * 8)) & 0xFFu;
  }
  return e.cby[d];
}
__device__ __forceinline__ void cby_clear(Env& e, int d) {
  if (CG_CBY_GLOBAL && e.cby_g) atomicAnd((unsigned int*)(e.cby_g + (d & ~3)), ~(0xFFu << ((d & 3) * 8)));
  else e.cby[d] = 0;
}
__device__ __forceinline__ void cby_or(Env& e, int d, uint32_t bits) {
  if (CG_CBY_GLOBAL && e.cby_g) atomicOr((unsigned int*)(e.cby_g + (d & ~3)), bits << ((d & 3) * 8));
  else byte_or(e.cby, d, bits);
}
__device__ __forceinline__ void cby_put(Env& e, int d, uint32_t v) {   // (one writer per byte and phase)
  if (CG_CBY_GLOBAL && e.cby_g) { cby_clear(e, d); if (v) cby_or(e, d, v); }
  else e.cby[d] = (uint8_t)v;
}

// number of set bits of blk in slot range [a, b)  (uniform; broadcast LDS reads)
__device__ __forceinline__ int range_popc(const uint32_t* blk, int a, int b) {
  if (a >= b) return 0;
  const int w0 = a >> 5, w1 = (b - 1) >> 5;
  int n = 0;
  for (int w = w0; w <= w1; ++w) {
    uint32_t x = blk[w];
    if (w == w0) x &= 0xFFFFFFFFu << (a & 31);
    if (w == w1 && ((b & 31) != 0)) x &= 0xFFFFFFFFu >> (32 - (b & 31));
    n += __popc(x);
  }
  return n;
}
// The same count with a fixed number of words read at once: W independent reads, one LDS latency.  For per-lane callers whose
// rows differ a lot in length (block / unblock pools: a full row next to a two-entry row) the rolled loop above makes every lane
// wait out the longest row, one word per trip.  Every kernel of a compile-time size uses these forms (pool_pick<NW>); rows longer
// than W words cannot occur there: cygym_create sends such a topology to the run-time-size kernels (DevTopo::ct), so there is no
// fallback loop.
#ifndef CG_WIDE_W
#define CG_WIDE_W 9
#endif
// W words read at once: 9 covers any row of <= 256 slots (256 devices), 3 any row of <= 64 slots (64 devices).
// Nine (W) clamped reads, W popcount-accumulates, then arithmetic instead of per-word masks and selects: the last word was read
// (W - words) times and counts once; the first word's bits below a and the last word's bits from b on are taken off.  (Half the
// vector instructions of a masked form -- and under four waves per SIMD a pool pick costs what it issues, not what it waits for.)
template <int W>
__device__ __forceinline__ int range_popc_words(const uint32_t* blk, int a, int b) {
  if (a >= b) return 0;
  const int w0 = a >> 5, w1 = (b - 1) >> 5;
  uint32_t x[W];
#pragma unroll
  for (int j = 0; j < W; ++j) { const int w = w0 + j; x[j] = blk[w <= w1 ? w : w1]; }
  int n = 0;
#pragma unroll
  for (int j = 0; j < W; ++j) n += __popc(x[j]);
  n -= (W - 1 - (w1 - w0)) * __popc(x[W - 1]);
  n -= __popc(x[0] & ~(0xFFFFFFFFu << (a & 31)));
  if (b & 31) n -= __popc(x[W - 1] & (0xFFFFFFFFu << (b & 31)));
  return n;
}
__device__ __forceinline__ int range_popc_wide(const uint32_t* blk, int a, int b) { return range_popc_words<CG_WIDE_W>(blk, a, b); }
// slot of the r-th entry in [a, b) whose blocked bit == want (uniform); r must be in range
__device__ __forceinline__ int range_select(const uint32_t* blk, int a, int b, bool want, int r) {
  const int w0 = a >> 5, w1 = (b - 1) >> 5;
  for (int w = w0; w <= w1; ++w) {
    uint32_t x = want ? blk[w] : ~blk[w];
    if (w == w0) x &= 0xFFFFFFFFu << (a & 31);
    if (w == w1 && ((b & 31) != 0)) x &= 0xFFFFFFFFu >> (32 - (b & 31));
    int c = __popc(x);
    if (r < c) return (w << 5) + nth_bit32_bisect(x, r);
    r -= c;
  }
  return -1;
}

// range_select with W words read at once and the in-word rank by bisection; per-lane callers (block / unblock pools).
// No masks at all: the candidates of the first word BELOW a are skipped by raising the rank by their number; what lies past b --
// the last word's upper bits, the clamped re-reads of that word -- comes after every candidate in range and is never reached
// (r is in range).  Five vector instructions per word.
template <int W>
__device__ __forceinline__ int range_select_words(const uint32_t* blk, int a, int b, bool want, int r) {
  const int w0 = a >> 5, w1 = (b - 1) >> 5;
  const uint32_t inv = want ? 0u : 0xFFFFFFFFu;
  uint32_t x[W > 1 ? W - 1 : 1];
#pragma unroll
  for (int j = 0; j < W - 1; ++j) { const int w = w0 + j; x[j] = blk[w <= w1 ? w : w1] ^ inv; }
  const int rr = r + __popc(x[0] & ~(0xFFFFFFFFu << (a & 31)));
  int cum = 0, wsel = 0, rbase = 0;
#pragma unroll
  for (int j = 0; j < W - 1; ++j) {
    cum += __popc(x[j]);
    if (cum <= rr) { wsel = j + 1; rbase = cum; }   // the candidate lies beyond word j
  }
  const uint32_t xs = blk[w0 + wsel] ^ inv;   // one more read for the word itself
  return ((w0 + wsel) << 5) + nth_bit32_bisect(xs, rr - rbase);
}
__device__ __forceinline__ int range_select_wide(const uint32_t* blk, int a, int b, bool want, int r) { return range_select_words<CG_WIDE_W>(blk, a, b, want, r); }

// multiplicity of every device in a list -> bytes in scr (as uint8 [Mp]); ids >= M ignored.
__device__ __forceinline__ void list_counts(Env& e, const int16_t* dev, int L) {
  uint32_t* w = e.scr;
  for (int i = e.lane; i < (e.MC * WAVE) / 4; i += WAVE) w[i] = 0;
  wsync();
  for (int p = e.lane; p < L; p += WAVE) {
    int d = dev[p];
    if (d >= 0 && d < e.M) atomicAdd(&w[d >> 2], 1u << ((d & 3) * 8));
  }
  wsync();
}

// true when the list fits one wave pass and holds no device twice (uniform)
__device__ __forceinline__ bool list_is_simple(Env& e, const int16_t* dev, int L) {
  if (L > WAVE) return false;
  uint8_t* own = (uint8_t*)e.scr;
  int d = -1;
  if (e.lane < L) { d = dev[e.lane]; if (d < 0 || d >= e.M) d = -1; }
  if (d >= 0) own[d] = (uint8_t)e.lane;
  wsync();
  bool clash = d >= 0 && own[d] != (uint8_t)e.lane;
  bool r = !__any(clash);
  wsync();
  return r;
}

// busy += 1 on every busy device (actions 2 / 10), saturating at 255
__device__ __forceinline__ void bump_busy(Env& e) {
  uint32_t* B = (uint32_t*)e.busy;
  for (int w = e.lane; w < (e.MS >> 2); w += WAVE) {
    uint32_t b = B[w];
    uint32_t full = ~nz01(~b) & ONES;          // bytes equal to 255
    uint32_t inc = nz01(b) & ~full;
    if (nz01(b) & full) e.eflags |= CG_E_BUSY_SAT;
    B[w] = b + inc;
  }
}

#endif  // CG_ENV_HPP
