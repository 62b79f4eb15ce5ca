// cygym_hip.hip -- MI355X (gfx950, wave64) kernels + C ABI of the batched CyGym tick.
//
// Execution model: ONE WAVEFRONT PER ENVIRONMENT.  A workgroup of WPB waves stages
// the shared topology (out-CSR row pointers + columns, static per-device columns)
// into LDS once, then every wave stages its own env's struct-of-arrays state
// (4 live byte planes + blocked-edge bitmask + log ring) into its private LDS
// region, runs the whole tick there, streams the observation out with 16-byte
// coalesced stores and writes the dirty planes back.  No MFMA: the path is
// integer / bit / index / RNG work bound by HBM traffic.
//
// Parallel restatements of the reference's sequential loops (each checked
// bit-for-bit against oracle/cygym_oracle.c, which keeps the reference's order):
//   * attacker spread (volt_typhoon_env.py:1126-1185): sources are processed
//     "in ascending id order, each seeing what earlier sources compromised".
//     Here: every source picks its target in parallel against a per-device
//     first-compromise time T[v] (atomicMin in LDS); iterate to the unique fix
//     point (== the sequential result; proof sketch in DESIGN.md).  Long rows
//     (attacker-owned hubs) are scanned 64 entries per step with ballot/ffs.
//   * "r-th element of a list in dict order" (random.choice over devices):
//     ballot + popcount ranking.
//   * random.sample(candidates, k) (CDSimulator.py:298): k smallest
//     (philox key, id) by a wave-wide radix select.
//   * per-device defender actions: multiplicity counts via LDS atomics, so
//     duplicate / unsorted device lists give the sequential result.
//
// Reference citations are relative to the reference checkout.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <new>
#include "cygym_abi.h"

#ifndef CG_LB
#define CG_LB 6
#endif
#define CG_E_STAR_OK 0x80  // kernel-private: star edges verified for the current owned set

namespace {

constexpr int WAVE = 64;

// The shared topology lives in ONE packed device blob whose layout is also the layout of the
// workgroup-shared LDS section (copied with 16-byte loads): byte offsets o_* into the blob.
//   [optr u16 M+1][ocol u16 E][os f32 M][ver f32 M][ano f32 M][dst u8 M][vul u8 M][nap u8 M] | [iptr u16 M+1][icol u16 E][ieid u16 E][oeid u16 E]
//   (ieid: out-slot of an in-entry; oeid: in-entry of an out-slot)
// The first `lds_bytes` bytes are staged in LDS: everything when it fits (in_lds), else all but the in-CSR.
struct DevTopo {
  int M, X, E, EW, MC, Mp;
  const uint8_t* blob;
  int o_optr, o_ocol, o_os, o_ver, o_ano, o_dst, o_vul, o_nap, o_iptr, o_icol, o_ieid, o_oeid;
  int blob_bytes, lds_bytes, in_lds, multi;
  int K, KW, x_bytes;   // extra-edge list: capacity, blocked-bit words, bytes of its per-wave LDS section
  // global views (host-side convenience; kernels outside the tick use them)
  const uint8_t *dstatic, *vuln, *napps;
  const float *os_val, *version, *anomaly;
  const uint16_t *out_ptr, *out_col;   // u16: E <= 65535, M <= 2048
  const uint16_t *in_ptr, *in_col, *in_eid;
};

struct KParams {
  const KParams* self;  // device copy of this struct (fused kernel re-reads it every tick instead of pinning SGPRs)
  DevTopo t;
  cygym_config c;
  cygym_buffers b;
  cygym_buffers snap;   // snap.flags == nullptr when absent
  cygym_actions a;
  cygym_outputs o;
  int n_envs;
  int n_ticks;          // ticks per launch (cygym_rollout); actions / outputs are [n_ticks][N] arrays
  int wave_lds;         // bytes of LDS per wave
  int shared_lds;       // bytes of the workgroup-shared LDS section
  unsigned long long* dbg;   // diagnostic builds (-DCG_STAMPS): [N][16] s_memtime stamps per env
};

#ifdef CG_STAMPS
#define SUBSTAMP(k) do { if (P.dbg && e.lane == 0) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); P.dbg[(size_t)e.env * 16 + (k)] = _t; } } while (0)
#define SUBVAL(k, v) do { if (P.dbg && e.lane == 0) P.dbg[(size_t)e.env * 16 + (k)] = (unsigned long long)(v); } while (0)
#define STAMP(k) do { if (P.dbg && lane == 0) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); P.dbg[(size_t)env * 16 + (k)] = _t; } } while (0)
#else
#define STAMP(k) do {} while (0)
#define SUBSTAMP(k) do {} while (0)
#define SUBVAL(k, v) do {} while (0)
#endif

#define COLD(c) __builtin_expect(!!(c), 0)   // rarely-taken extra-edge paths: keep them out of the hot layout

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

// ---- SWAR on 4 device bytes per 32-bit word ----
#define ONES 0x01010101u
__device__ __forceinline__ uint32_t nz01(uint32_t b) {   // 0x01 in every byte of b that is non-zero
  return ((b | ((b & 0x7f7f7f7fu) + 0x7f7f7f7fu)) >> 7) & ONES;
}

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
  const uint16_t *optr, *ocol;
  const uint8_t *dst, *vul, *nap;
  const float *osv, *ver, *ano;
  // in-CSR + slot<->entry maps: global memory (L2-resident blob); read by block/unblock and evolve only
  const uint16_t *iptr_g, *icol_g, *ieid_g, *oeid_g;
  uint8_t* stash;    // global [4][M] of this env
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
  __device__ __forceinline__ int iptr(int d) const { return iptr_g[d]; }
  __device__ __forceinline__ void set_busy(int d, int v) {
    if (v > 255) { v = 255; eflags |= CG_E_BUSY_SAT; }
    busy[d] = (uint8_t)v;
  }
};

__device__ __forceinline__ void byte_or(uint8_t* base, int d, uint32_t bits) {
  atomicOr((unsigned int*)(base + (d & ~3)), bits << ((d & 3) * 8));
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
// slot of the r-th entry in [a, b) whose blocked bit == want (uniform); r must be in range
__device__ __forceinline__ int range_select(const uint32_t* blk, int a, int b, bool want, int r) {
  const int w0 = a >> 5, w1 = (b - 1) >> 5;
  for (int w = w0; w <= w1; ++w) {
    uint32_t x = want ? blk[w] : ~blk[w];
    if (w == w0) x &= 0xFFFFFFFFu << (a & 31);
    if (w == w1 && ((b & 31) != 0)) x &= 0xFFFFFFFFu >> (32 - (b & 31));
    int c = __popc(x);
    if (r < c) return (w << 5) + nth_bit32(x, r);
    r -= c;
  }
  return -1;
}

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

// ---------------- edges added by evolve_network (extra-edge list, cygym_spec.h) ----------------
__device__ __forceinline__ int x_cnt(const Env& e) { return (int)((uint32_t)e.eflags >> CG_E_NX_SHIFT); }
__device__ __forceinline__ bool x_isout(const Env& e, int d) { return (e.xmo[d >> 6] >> (d & 63)) & 1ull; }
__device__ __forceinline__ bool x_isinc(const Env& e, int d) { return (e.xmi[d >> 6] >> (d & 63)) & 1ull; }
__device__ __forceinline__ bool x_blocked(const Env& e, int j) { return (e.xb[j >> 5] >> (j & 31)) & 1u; }
// first list entry with key >= k (per lane; the list is short)
__device__ __forceinline__ int x_lower(const Env& e, uint32_t k) {
  int lo = 0, hi = x_cnt(e);
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (e.xk[mid] < k) lo = mid + 1; else hi = mid; }
  return lo;
}
// device masks of the live entries (uniform)
__device__ __forceinline__ void x_masks(Env& e) {
  uint32_t* mo = (uint32_t*)e.xmo; uint32_t* mi = (uint32_t*)e.xmi;
#pragma nounroll
  for (int i = e.lane; i < 2 * e.MC; i += WAVE) { mo[i] = 0; mi[i] = 0; }
  wsync();
  const int n = x_cnt(e);
#pragma nounroll
  for (int j = e.lane; j < n; j += WAVE) {
    const uint32_t k = e.xk[j];
    const int u = (int)(k >> 16), v = (int)(k & 0xFFFFu);
    atomicOr(&mo[u >> 5], 1u << (u & 31));
    atomicOr(&mi[u >> 5], 1u << (u & 31));
    atomicOr(&mi[v >> 5], 1u << (v & 31));
  }
  wsync();
}
// g.add_edges([(u, v)]): sorted insert (uniform); false (and CG_E_TOPO_OVF) when the list is full
__device__ __forceinline__ bool x_add(Env& e, int u, int v) {
  const int n = x_cnt(e);
  if (n >= e.K) { e.eflags |= CG_E_TOPO_OVF; return false; }
  const uint32_t key = ((uint32_t)u << 16) | (uint32_t)v;
  int pos = 0;
#pragma nounroll
  for (int j0 = 0; j0 < n; j0 += WAVE) { const int j = j0 + e.lane; pos += __popcll(ballot(j < n && e.xk[j] < key)); }
#pragma nounroll
  for (int j0 = n > 0 ? ((n - 1) / WAVE) * WAVE : -1; j0 >= 0; j0 -= WAVE) {   // shift the tail up, top chunk first
    const int j = j0 + e.lane;
    const bool mv = j < n && j >= pos;
    const uint32_t k = mv ? e.xk[j] : 0u;
    wsync();
    if (mv) e.xk[j + 1] = k;
    wsync();
  }
  if (e.lane == 0) e.xk[pos] = key;
  wsync();
  e.eflags += 1 << CG_E_NX_SHIFT;
  e.x_dirty = true;
  return true;
}
// per-lane walk over the MERGED out-row of a device: the base CSR row and the device's added edges, by
// ascending neighbour id (what _rebuild_graph_cache makes of igraph's neighbour lists)
struct XWalk {
  int k, o1, j, n, s;
  uint32_t vx;   // neighbour of the pending list entry, 0x10000 = none
  __device__ __forceinline__ void load(const Env& e) {
    const uint32_t key = j < n ? e.xk[j] : 0xFFFFFFFFu;
    vx = (int)(key >> 16) == s ? (key & 0xFFFFu) : 0x10000u;
  }
  __device__ __forceinline__ void init(const Env& e, int src) {
    s = src; k = e.optr[s]; o1 = e.optr[s + 1]; n = x_cnt(e);
    j = x_lower(e, (uint32_t)s << 16);
    load(e);
  }
  __device__ __forceinline__ bool done() const { return k >= o1 && vx == 0x10000u; }
  __device__ __forceinline__ bool at_extra(const Env& e) const { return k >= o1 || vx < (uint32_t)e.ocol[k]; }
  __device__ __forceinline__ void next(const Env& e, bool was_extra) {
    if (was_extra) { ++j; load(e); } else ++k;
  }
};

// ---------------- defender ----------------
__device__ __forceinline__ void def_global(Env& e, const KParams& P, int at, const int16_t* dev, int L, double& cost,
                                           bool& dirty, bool grouped, int32_t* ie, double* fe) {
  const double ds = P.c.def_scale;
  const int M = e.M;
  if (at == 2) {  // :918-926
    ie[CG_I_CKPT_CNT] += 1;
    e.eflags |= CG_E_HAS_CKPT;
    cost += -0.5 * L * ds;
    fe[CG_D_DEF_COST] += 0.5 * L * ds;
    bump_busy(e);
  } else if (at == 3) {  // :928-943
    ie[CG_I_REVERT_CNT] += 1;
    if (e.eflags & CG_E_HAS_CKPT) {
      for (int d = e.lane; d < M; d += WAVE) {
        e.busy[d] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_REVERT, d, 0), 0, P.c.default_high);
        e.wl[d] = 0;
        e.flags[d] &= (uint8_t)~CG_F_WLADV;
      }
      cost += -1.0 * L * ds;
      dirty = true;
    }
  } else if (at == 10) {  // :945-962
    if (!grouped) {
      if (L > 0) {
        int d = dev[0];
        if (d >= 0 && d < M && e.lane == 0) e.set_busy(d, e.busy[d] + 1);
      } else {
        bump_busy(e);
      }
    }
    cost += -1.0 * ds;
    if (e.log_total > 0) { e.eflags |= CG_E_DET_TRAIN; e.eflags &= ~CG_E_DET_RANDOM; }
  } else if (at == 11) {  // :964-976, _device_state :419-428
    if (L > 0) {
      int d = dev[0];
      if (d >= 0 && d < M && e.lane == 0) {
        e.stash[d] = (uint8_t)(CG_S_VALID | (e.flags[d] & CG_S_KEEP));
        e.stash[M + d] = e.busy[d];
        e.stash[2 * M + d] = e.wl[d];
        e.stash[3 * M + d] = e.cby[d];
      }
    }
    ie[CG_I_CKPT_CNT] += 1;
    cost += -0.1 * ds;
    fe[CG_D_DEF_COST] += 0.1 * ds;
  } else {
    return;
  }
  wsync();
}

// action 1 over one device list; `occ` (u8 [Mp], LDS) carries stall occurrence numbers across
// the groups of one step_grouped tick (nullptr for single-action steps).
__device__ __forceinline__ void def_clean(Env& e, const KParams& P, const int16_t* dev, int L, double& cost,
                                          int32_t* ie, double* fe, uint8_t* occ) {
  const double ds = P.c.def_scale;
  int a, b, disc;
  if (list_is_simple(e, dev, L)) {   // list-major: one lane per list entry, one draw per lane
    int d = -1;
    if (e.lane < L) { d = dev[e.lane]; if (d < 0 || d >= e.M) d = -1; }
    uint8_t f = d >= 0 ? e.flags[d] : (uint8_t)CG_F_NYA;
    bool hit = !(f & CG_F_NYA) && !(f & CG_F_OWNED);
    a = __popcll(ballot(hit && (f & CG_F_COMP)));
    b = __popcll(ballot(hit && !(f & CG_F_COMP)));
    int dl = 0;
    if (hit) {
      dl = e.cby[d];
      e.cby[d] = 0;
      e.flags[d] = (uint8_t)(f & ~(CG_F_COMP | CG_F_WLADV));
      int b0 = 0;
      if (occ) { b0 = occ[d]; occ[d] = (uint8_t)(b0 + 1); }
      e.busy[d] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_CLEAN, d, b0), 0, P.c.default_high);
      e.wl[d] = 0;
    }
    disc = 0;
    for (int bit = 0; bit < CG_MAX_EXPLOITS; ++bit) disc |= (ballot((dl >> bit) & 1) != 0ull) << bit;
  } else {   // device-major with multiplicities: duplicates / long lists give the sequential result
    list_counts(e, dev, L);
    const uint8_t* cnt = (const uint8_t*)e.scr;
    int n_first_comp = 0, n_first_clean = 0, n_rest = 0;
    disc = 0;
    for (int c = 0; c < e.MC; ++c) {
      int d = c * WAVE + e.lane;
      bool hit = false;
      int k = 0;
      uint8_t f = 0;
      if (d < e.M) {
        k = cnt[d];
        f = e.flags[d];
        hit = k > 0 && !(f & CG_F_NYA) && !(f & CG_F_OWNED);
      }
      if (hit) {
        if (f & CG_F_COMP) ++n_first_comp; else ++n_first_clean;
        n_rest += k - 1;
        disc |= e.cby[d];
        e.cby[d] = 0;
        e.flags[d] = (uint8_t)(f & ~(CG_F_COMP | CG_F_WLADV));
        int b0 = occ ? occ[d] : 0;
        if (occ) occ[d] = (uint8_t)(b0 + k);
        e.busy[d] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_CLEAN, d, b0 + k - 1), 0, P.c.default_high);
        e.wl[d] = 0;
      }
    }
    a = wave_sum(n_first_comp);
    b = wave_sum(n_first_clean) + wave_sum(n_rest);
    disc = wave_or(disc);
  }
  cost += (0.3 * a - 0.01 * b) * ds;
  fe[CG_D_CLEAN_COST] += (0.3 * a + 0.01 * b) * ds;
  fe[CG_D_DEF_COST] += (0.3 * a + 0.01 * b) * ds;
  ie[CG_I_DISCOVERED] |= disc;
  wsync();
}

// Actions 6 / 9 (volt_typhoon_env.py:501-511, 1071-1100): the pool of device d is its out-entries with the
// wanted blocked state (row order) followed by its in-entries (in-row order).  With the blocked bits kept in
// BOTH orders (blk by out-slot, bin by in-entry) the pool is two contiguous bit ranges, so counting and
// selecting are word operations (no per-edge gathers).
struct Pick { int slot, j, x; };   // out-slot, in-entry and the OTHER endpoint of the chosen edge; slot < 0: empty pool
// All pointers are passed BY VALUE: selecting between addresses of Env members (which the optimiser does when
// two branch arms load through different members) would pin the whole Env struct in scratch memory.
struct PoolPtrs {
  uint32_t *blk, *bin;                   // LDS
  const uint16_t *optr, *ocol;           // LDS
  const uint16_t *icol, *ieid, *oeid;    // global
};
__device__ __forceinline__ Pick pool_pick(const PoolPtrs q, bool want, uint32_t u, int o0, int o1, int i0, int i1) {
  Pick p; p.slot = -1; p.j = -1; p.x = -1;
  const int nbo = range_popc(q.blk, o0, o1), nbi = range_popc(q.bin, i0, i1);
  const int n_out = want ? nbo : (o1 - o0) - nbo;
  const int n_in = want ? nbi : (i1 - i0) - nbi;
  const int n = n_out + n_in;
  if (n == 0) return p;
  const int r = (int)cg_index(u, (uint32_t)n);
  const bool from_out = r < n_out;
  int slot = -1, j = -1;
  if (from_out) slot = range_select(q.blk, o0, o1, want, r);
  else          j = range_select(q.bin, i0, i1, want, r - n_out);
  if (from_out) { p.slot = slot; p.j = q.oeid[slot]; p.x = q.ocol[slot]; }      // independent loads
  else          { p.j = j; p.slot = q.ieid[j]; p.x = q.icol[j]; }
  return p;
}
// toggle edge (su -> sv) given one of its slots; with duplicate (u,v) out-entries all of them share the state
__device__ __forceinline__ void pool_flip(const PoolPtrs q, bool multi, const Pick p, int d, int o0, int o1, bool want) {
  if (!multi) {
    if (!want) { atomicOr(&q.blk[p.slot >> 5], 1u << (p.slot & 31)); atomicOr(&q.bin[p.j >> 5], 1u << (p.j & 31)); }
    else       { atomicAnd(&q.blk[p.slot >> 5], ~(1u << (p.slot & 31))); atomicAnd(&q.bin[p.j >> 5], ~(1u << (p.j & 31))); }
  } else {
    int su = d, sv = p.x;
    if (!(p.slot >= o0 && p.slot < o1)) { su = p.x; sv = d; }
    for (int k = q.optr[su]; k < q.optr[su + 1]; ++k)
      if (q.ocol[k] == sv) {
        const int j = q.oeid[k];
        if (!want) { atomicOr(&q.blk[k >> 5], 1u << (k & 31)); atomicOr(&q.bin[j >> 5], 1u << (j & 31)); }
        else       { atomicAnd(&q.blk[k >> 5], ~(1u << (k & 31))); atomicAnd(&q.bin[j >> 5], ~(1u << (j & 31))); }
      }
  }
}

// Block / unblock on an env whose device list touches an endpoint of an ADDED edge: the incident pools are the
// merged rows (:502-511).  Sequential over the list (the reference's own order), every step wave-cooperative.
// Element r of a merged pool: walk the added-edge candidates in row order; candidate j has rank
// (#added candidates before it) + (#base candidates with a smaller neighbour id).
__device__ __forceinline__ void block_seq(Env& e, const PoolPtrs q, const int16_t* dev, int L, bool want, uint32_t site,
                                          int& n_act, int& n_hit) {
  const int M = e.M;
  uint8_t* occ = (uint8_t*)(e.scr + e.MC * WAVE);
#pragma nounroll
  for (int i = e.lane; i < (e.MC * WAVE) / 4; i += WAVE) ((uint32_t*)occ)[i] = 0;
  wsync();
  const int n = x_cnt(e);
#pragma nounroll
  for (int p = 0; p < L; ++p) {
    const int d = dev[p];
    if (d < 0 || d >= M || (e.flags[d] & CG_F_NYA)) continue;
    ++n_act;
    const int o0 = e.optr[d], o1 = e.optr[d + 1], i0 = e.iptr(d), i1 = e.iptr(d + 1);
    const int nbo = range_popc(q.blk, o0, o1), nbi = range_popc(q.bin, i0, i1);
    const int b_out = want ? nbo : (o1 - o0) - nbo, b_in = want ? nbi : (i1 - i0) - nbi;
    int x_out = 0, x_in = 0;
#pragma nounroll
    for (int j0 = 0; j0 < n; j0 += WAVE) {
      const int j = j0 + e.lane;
      const uint32_t k = j < n ? e.xk[j] : 0u;
      const bool c = j < n && x_blocked(e, j) == want;
      x_out += __popcll(ballot(c && (int)(k >> 16) == d));
      x_in += __popcll(ballot(c && (int)(k & 0xFFFFu) == d));
    }
    const int total = b_out + x_out + b_in + x_in;
    if (total == 0) continue;
    const int occ_d = occ[d];
    int r = (int)cg_index(e.draw(site, d, occ_d), (uint32_t)total);
    const bool from_out = r < b_out + x_out;
    if (!from_out) r -= b_out + x_out;
    int pick_x = -1, x_before = 0;   // chosen added edge, or the number of added candidates ahead of element r
#pragma nounroll
    for (int j0 = 0; j0 < n && pick_x < 0; j0 += WAVE) {
      const int j = j0 + e.lane;
      const uint32_t kk = j < n ? e.xk[j] : 0u;
      uint64_t xm = ballot(j < n && x_blocked(e, j) == want && (from_out ? (int)(kk >> 16) == d : (int)(kk & 0xFFFFu) == d));
#pragma nounroll
      while (xm) {
        const int jj = j0 + __builtin_ctzll(xm);
        xm &= xm - 1;
        const uint32_t key = e.xk[jj];
        const int other = from_out ? (int)(key & 0xFFFFu) : (int)(key >> 16);
        int cb = 0;   // base candidates ahead of this added edge
        if (from_out) {
#pragma nounroll
          for (int k0 = o0; k0 < o1; k0 += WAVE) {
            const int k = k0 + e.lane;
            cb += __popcll(ballot(k < o1 && (((q.blk[k >> 5] >> (k & 31)) & 1u) != 0) == want && (int)q.ocol[k] < other));
          }
        } else {
#pragma nounroll
          for (int k0 = i0; k0 < i1; k0 += WAVE) {
            const int k = k0 + e.lane;
            cb += __popcll(ballot(k < i1 && (((q.bin[k >> 5] >> (k & 31)) & 1u) != 0) == want && (int)q.icol[k] <= other));
          }
        }
        if (r == x_before + cb) { pick_x = jj; break; }
        if (r < x_before + cb) { xm = 0; j0 = n; break; }   // element r is a base candidate
        ++x_before;
      }
    }
    if (pick_x >= 0) {
      if (e.lane == 0) {
        if (!want) e.xb[pick_x >> 5] |= 1u << (pick_x & 31); else e.xb[pick_x >> 5] &= ~(1u << (pick_x & 31));
      }
      e.x_dirty = true;
    } else {
      Pick pk;
      const int t = r - x_before;
      if (from_out) { pk.slot = range_select(q.blk, o0, o1, want, t); pk.j = q.oeid[pk.slot]; pk.x = q.ocol[pk.slot]; }
      else          { pk.j = range_select(q.bin, i0, i1, want, t); pk.slot = q.ieid[pk.j]; pk.x = q.icol[pk.j]; }
      if (e.lane == 0) pool_flip(q, e.multi, pk, d, o0, o1, want);
      e.blk_dirty = true;
    }
    if (e.lane == 0) occ[d] = (uint8_t)(occ_d + 1);
    ++n_hit;
    wsync();
  }
}

template <bool XE>
__device__ __forceinline__ void def_per_device(Env& e, const KParams& P, int at, const int16_t* dev, int L, int app,
                                               double& cost, bool& dirty, int32_t* ie, double* fe) {
  const double ds = P.c.def_scale;
  const int M = e.M;
  if (at == 1) { def_clean(e, P, dev, L, cost, ie, fe, nullptr); return; }
  if (at == 6 || at == 9) {  // sequential semantics: each pick changes the pools of BOTH endpoints
    __builtin_amdgcn_s_setprio(3);   // long path: see the spread
    const uint32_t site = at == 6 ? CG_SITE_PICK_BLOCK : CG_SITE_PICK_UNBLOCK;
    const bool want = (at == 9);
    const bool simple = list_is_simple(e, dev, L);   // no device twice => occurrence number is always 0
    uint8_t* occ = (uint8_t*)(e.scr + e.MC * WAVE);   // second scratch half (first half: first-touch table)
    if (!simple) {
      for (int i = e.lane; i < (e.MC * WAVE) / 4; i += WAVE) ((uint32_t*)occ)[i] = 0;
      wsync();
    }
    uint32_t* fh = e.scr;   // [Mp] first remaining entry (lane) whose flipped edge ends at this device
    PoolPtrs q;
    q.blk = e.blk; q.bin = e.bin; q.optr = e.optr; q.ocol = e.ocol; q.icol = e.icol_g; q.ieid = e.ieid_g; q.oeid = e.oeid_g;
    const bool multi = e.multi;
    int n_act = 0, n_hit = 0;
    bool seq = false;
    if (COLD(XE && x_cnt(e) > 0)) {   // does the list touch an endpoint of an added edge?
      for (int p0 = 0; p0 < L; p0 += WAVE) {
        const int p = p0 + e.lane;
        const int d = p < L ? dev[p] : -1;
        if (__any(d >= 0 && d < M && x_isinc(e, d))) seq = true;
      }
    }
    if constexpr (XE) { if (COLD(seq)) block_seq(e, q, dev, L, want, site, n_act, n_hit); }
    for (int p0 = 0; p0 < L && !seq; p0 += WAVE) {
      // one lane per list entry: device, row bounds and the (occurrence 0) draw
      const int p = p0 + e.lane;
      int d = -1, o0 = 0, o1 = 0, i0 = 0, i1 = 0;
      uint32_t u = 0;
      if (p < L) { d = dev[p]; if (d < 0 || d >= M || (e.flags[d] & CG_F_NYA)) d = -1; }
      if (d >= 0) {
        o0 = e.optr[d]; o1 = e.optr[d + 1]; i0 = e.iptr(d); i1 = e.iptr(d + 1);
        u = e.draw(site, d, 0);
      }
      uint64_t am = ballot(d >= 0);
      n_act += __popcll(am);
      SUBSTAMP(10);
      int n_pass = 0;
      // Speculate: every remaining entry picks on the bitmasks as they stand.  An entry is exact unless an
      // EARLIER remaining entry flips an edge ending at its device (or is the same device); apply the exact
      // prefix in parallel and repeat from the first inexact entry (at least one entry retires per pass).
      while (am) {
        for (int i = e.lane; i < e.MC * WAVE; i += WAVE) fh[i] = 0xFFFFFFFFu;
        wsync();
        const bool mine = (am >> e.lane) & 1ull;
        Pick pk; pk.slot = -1; pk.j = -1; pk.x = -1;
        if (mine) {
          uint32_t uu = u;
          if (!simple) { const int b = occ[d]; if (b > 0) uu = e.draw(site, d, b); }
          pk = pool_pick(q, want, uu, o0, o1, i0, i1);
          if (pk.slot >= 0) atomicMin(&fh[pk.x], (uint32_t)e.lane);
          if (!simple) atomicMin(&fh[d], (uint32_t)e.lane);   // a repeated device must wait for its first occurrence
        }
        wsync();
        const bool taint = mine && fh[d] < (uint32_t)e.lane;
        const uint64_t tm = ballot(taint);
        const int q0 = tm ? __builtin_ctzll(tm) : WAVE;
        const bool apply = mine && pk.slot >= 0 && e.lane < q0;
        if (apply) {
          pool_flip(q, multi, pk, d, o0, o1, want);
          if (!simple) occ[d] += 1;
        }
        const uint64_t apm = ballot(apply);
        n_hit += __popcll(apm);
        if (apm) e.blk_dirty = true;
        wsync();
        am &= q0 < WAVE ? (~0ull << q0) : 0ull;
        ++n_pass;
      }
      SUBSTAMP(11);
      SUBVAL(15, n_pass);
      SUBVAL(14, n_act);
    }
    cost += -0.5 * n_act * ds;
    fe[CG_D_DEF_COST] += 0.5 * n_act * ds;
    if (at == 6) ie[CG_I_EDGES_BLOCKED] += n_hit; else ie[CG_I_EDGES_ADDED] += n_hit;
    if (n_hit) dirty = true;
    __builtin_amdgcn_s_setprio(0);
    return;
  }
  if (at == 12) {  // :1102-1109: the reference restores device_indices[0] once per listed active device
    int d0 = dev[0];
    bool ok = d0 >= 0 && d0 < M;
    const int o = ok ? d0 : 0;
    uint8_t sf = ok ? e.stash[o] : 0;
    if (!(sf & CG_S_VALID)) return;
    uint8_t sb = e.stash[M + o], sw = e.stash[2 * M + o], sc = e.stash[3 * M + o];
    int n_iter = 0;
    for (int p = 0; p < L; ++p) {   // uniform scalar walk: restoring d0 may change ITS Not_yet_added
      int d = dev[p];
      if (d < 0 || d >= M) continue;
      if (e.flags[d] & CG_F_NYA) continue;
      ++n_iter;
      if (n_iter == 1) {
        if (e.lane == 0) {
          e.flags[d0] = (uint8_t)((e.flags[d0] & ~CG_S_KEEP) | (sf & CG_S_KEEP));
          e.busy[d0] = sb; e.wl[d0] = sw; e.cby[d0] = sc;
        }
        wsync();
      }
    }
    cost += -1.0 * n_iter * ds;
    fe[CG_D_DEF_COST] += 1.0 * n_iter * ds;
    return;
  }
  // count-based actions.  n_mult = list entries (with multiplicity) on active devices.
  const bool simple = (at == 4 || at == 7) ? list_is_simple(e, dev, L) : true;
  int n_mult = 0, n_dist = 0;
  if (simple && L <= WAVE) {   // list-major
    int d = -1;
    if (e.lane < L) { d = dev[e.lane]; if (d < 0 || d >= M) d = -1; }
    uint8_t f = d >= 0 ? e.flags[d] : (uint8_t)CG_F_NYA;
    bool hit = !(f & CG_F_NYA);
    n_mult = n_dist = __popcll(ballot(hit));
    if (hit && at == 4) {  // :1013-1018
      if (app >= 0 && app < e.nap[d])
        e.busy[d] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_PATCH, d, 0), 0, P.c.default_high);
    } else if (hit && at == 7) {  // :1082-1089
      e.flags[d] = (uint8_t)((f | CG_F_NYA) & ~(CG_F_COMP | CG_F_WLADV));
      e.cby[d] = 0;
      e.wl[d] = 0;
    }
  } else {   // device-major with multiplicities
    list_counts(e, dev, L);
    const uint8_t* cnt = (const uint8_t*)e.scr;
    for (int c = 0; c < e.MC; ++c) {
      int d = c * WAVE + e.lane;
      if (d < M) {
        int k = cnt[d];
        uint8_t f = e.flags[d];
        if (k > 0 && !(f & CG_F_NYA)) {
          n_mult += k; n_dist += 1;
          if (at == 4) {
            if (app >= 0 && app < e.nap[d])
              e.busy[d] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_PATCH, d, k - 1), 0, P.c.default_high);
          } else if (at == 7) {
            e.flags[d] = (uint8_t)((f | CG_F_NYA) & ~(CG_F_COMP | CG_F_WLADV));
            e.cby[d] = 0;
            e.wl[d] = 0;
          }
        }
      }
    }
    n_mult = wave_sum(n_mult);
    n_dist = wave_sum(n_dist);
  }
  wsync();
  if (at == 4) {
    cost += -1.0 * n_mult * ds;
  } else if (at == 7) {
    cost += -0.5 * n_dist * ds;   // a repeated entry finds the device already removed (:992)
    if (n_dist > 0) dirty = true;
  } else if (at == 5) {  // fast scan :1020-1069
    ie[CG_I_SCAN_CNT] += n_mult;
    int w = e.log_total < CG_SCAN_WINDOW ? e.log_total : CG_SCAN_WINDOW;
    if (w > 0 && n_mult > 0) {
      cost += -0.5 * n_mult * ds;
      fe[CG_D_DEF_COST] += 0.5 * n_mult * ds;
      if (e.eflags & CG_E_DET_RANDOM) {  // Detector.batch_predict coin mode CDSimulator.py:715-716
        const int majority = w / 2 + 1;
        for (int s = 0; s < n_mult; ++s) {
          bool anom = false;
          if (e.lane < w) anom = cg_index(e.draw(CG_SITE_DET_COIN, e.lane, s), 2) == 0;
          uint64_t m = ballot(anom);
          if (__popcll(m) >= majority && anom) {
            uint32_t idx = (uint32_t)(e.log_total - w + e.lane);
            int snd = e.ring[2 * (idx % CG_LOG_RING)];
            atomicAnd((unsigned int*)(e.flags + (snd & ~3)), ~((uint32_t)CG_F_COMP << ((snd & 3) * 8)));
            e.busy[snd] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_SCAN, snd, s), 0, P.c.default_high);
          }
          wsync();
        }
      }
      // untrained detector: all "D" (CDSimulator.py:718-719); trained mode is outside the pinned scope
    }
  } else if (at == 13) {  // :1111-1123 -- acts on device_indices[0] once per listed active device
    int d0 = dev[0];
    if (n_mult > 0 && d0 >= 0 && d0 < M && e.lane == 0) {
      e.flags[d0] &= (uint8_t)~(CG_F_COMP | CG_F_WLADV);
      e.cby[d0] = 0;
      e.wl[d0] = 0;
      e.busy[d0] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_ISOLATE, d0, n_mult - 1), 3, P.c.default_high + 3);
    }
    cost += -3.0 * n_mult * ds;
    fe[CG_D_CLEAN_COST] += 3.0 * n_mult * ds;
    fe[CG_D_DEF_COST] += 3.0 * n_mult * ds;
    wsync();
  }
}

// ---------------- attacker ----------------
#define T_INF 0xFFFFFFFFu
constexpr int LONG_ROW = 8;
#define CG_D_FULLROW 0x04  // library-private static bit: the out-row is "every other device, ascending"

// T[v] = (first-compromise time << 2) | eligibility bits (bit0 reachable, bit1 known & vulnerable to this
// exploit): one LDS word answers "can source s take v".  atomicMin keeps the (constant) low bits intact.
#define T_TIME_INF 0x3FFFFFFFu
__device__ __forceinline__ bool spread_ok(const uint32_t* T, int v, int s) {
  const uint32_t t = T[v];
  return (t & 1u) || ((t & 2u) && ((t >> 2) >= (uint32_t)(s + 1)));
}
__device__ __forceinline__ void spread_take(uint32_t* T, int v, int s) {
  atomicMin(&T[v], ((uint32_t)(s + 1) << 2) | (T[v] & 3u));
}
// first slot k in [from, o1) that source s can take, or o1
__device__ __forceinline__ int spread_scan_lane(const Env& e, const uint32_t* T, int s, bool dc,
                                                int from, int o1) {
  for (int k = from; k < o1; ++k) {
    if (e.blocked(k)) continue;
    if (dc || spread_ok(T, e.ocol[k], s)) return k;
  }
  return o1;
}
// same for a FULL row (slot k <-> device v = k - o0 + (k - o0 >= s)): walk the candidate-device bitmask
// instead of the row; `cand` holds reach | (known & vulnerable & not compromised at the start)
__device__ __forceinline__ int spread_scan_full(const Env& e, const uint32_t* T, const uint64_t* cand,
                                                int s, int from, int o0, int o1) {
  if (from >= o1) return o1;
  int v_from = from - o0; if (v_from >= s) ++v_from;
  for (int w = v_from >> 6; w < e.MC; ++w) {
    uint64_t m = cand[w];
    if (w == (v_from >> 6)) m &= ~0ull << (v_from & 63);
    while (m) {
      int v = (w << 6) + __builtin_ctzll(m);
      m &= m - 1;
      if (v == s) continue;
      int k = o0 + v - (v > s ? 1 : 0);
      if (e.blocked(k)) continue;
      if (spread_ok(T, v, s)) return k;
    }
  }
  return o1;
}
__device__ __forceinline__ int spread_scan_coop(const Env& e, const uint32_t* T, int s, bool dc,
                                                int from, int o1) {
  for (int k0 = from; k0 < o1; k0 += WAVE) {
    int k = k0 + e.lane;
    bool p = false;
    if (k < o1 && !e.blocked(k)) p = dc || spread_ok(T, e.ocol[k], s);
    uint64_t m = ballot(p);
    if (m) return k0 + __builtin_ctzll(m);
  }
  return o1;
}

// One fix-point round for the sources whose rows carry ADDED edges: per-lane walk of the merged row;
// cur[s] = o0 + index in the merged row (o0 + merged length = nothing to take).  Kept out of the main
// round loop so that its registers are not live there.
__device__ __forceinline__ bool spread_x_round(const Env& e, uint32_t* T, uint16_t* cur, const uint16_t* slist, int n_src, int round) {
  bool changed = false;
#pragma nounroll
  for (int b0 = 0; b0 < n_src; b0 += WAVE) {
    const int i = b0 + e.lane;
    if (i >= n_src) continue;
    const int s = slist[i];
    if (!x_isout(e, s)) continue;
    const int o0 = e.optr[s], k0 = cur[s];
    const bool dc = e.dst[s] & CG_D_DC;
    XWalk w; w.init(e, s);
    int m = 0, hit = -1, hv = 0;
    const int m_from = k0 - o0;
    while (!w.done()) {
      const bool ex = w.at_extra(e);
      const int v = ex ? (int)w.vx : (int)e.ocol[w.k];
      const bool bl = ex ? x_blocked(e, w.j) : e.blocked(w.k);
      if (m >= m_from && !bl && (dc || spread_ok(T, v, s))) { hit = m; hv = v; break; }
      w.next(e, ex); ++m;
    }
    const int k = o0 + m;   // m == merged length when nothing can be taken
    if (k != k0) { cur[s] = (uint16_t)k; changed = true; }
    if (hit >= 0 && (round == 0 || k != k0)) spread_take(T, hv, s);
  }
  return changed;
}
// log entries of those sources (unblocked merged entries up to and including the pick) + DC attribution
__device__ __forceinline__ int spread_x_counts(Env& e, const uint16_t* cur, const uint16_t* slist, uint16_t* cntv, int n_src, uint8_t ebit) {
  int total = 0;
#pragma nounroll
  for (int b0 = 0; b0 < n_src; b0 += WAVE) {
    const int i = b0 + e.lane;
    int n = 0;
    if (i < n_src && x_isout(e, slist[i])) {
      const int s = slist[i];
      XWalk w; w.init(e, s);
      const int last = (int)cur[s] - (int)e.optr[s];
      const bool dc = e.dst[s] & CG_D_DC;
      for (int m = 0; !w.done() && m <= last; ++m) {
        const bool ex = w.at_extra(e);
        n += !(ex ? x_blocked(e, w.j) : e.blocked(w.k));
        if (m == last && dc) byte_or(e.cby, ex ? (int)w.vx : (int)e.ocol[w.k], ebit);   // DC attribution :1163-1168
        w.next(e, ex);
      }
      cntv[i] = (uint16_t)n;
    }
    total += wave_sum(n);
  }
  return total;
}

template <bool XE>
__device__ __forceinline__ void attacker_spread(Env& e, const KParams& P, const int32_t* expl, int n_expl,
                                                uint64_t* srcb) {
  const int M = e.M, MC = e.MC, Mp = MC * WAVE;
  uint32_t* T = e.scr;                          // [Mp] first-compromise time (source id + 1)
  uint16_t* cur = (uint16_t*)(e.scr + Mp);      // [Mp] current pick (slot) per source DEVICE, row end = none
  uint16_t* cntv = cur + Mp;                    // [Mp] log entries per COMPACT source index
  uint16_t* slist = e.lsrc;                     // [Mp] the sources in id order (snapshot :1127)
  uint64_t* cand = (uint64_t*)e.marks;          // [MC] candidate-device bitmask for full rows
  int n_src = 0;
#pragma unroll
  for (int c = 0; c < MC; ++c) {   // the sources in id order (compact list)
    const uint64_t sm = srcb[c];
    if ((sm >> e.lane) & 1ull) slist[n_src + below(sm)] = (uint16_t)(c * WAVE + e.lane);
    n_src += __popcll(sm);
  }
  wsync();
  const bool xany = XE && x_cnt(e) > 0;   // this env has added edges: their sources walk merged rows
  int zocc = 0;
  for (int j = 0; j < n_expl; ++j) {
    int raw = expl[j];
    if (P.c.zero_day) {  // :1131-1146
      uint32_t mask = (uint32_t)P.c.zero_day_owned_mask;
      bool in = raw >= 0 && raw < 32 && ((mask >> raw) & 1u);
      if (!in) {
        int cnt = __popc(mask);
        if (cnt == 0) continue;
        int r = (int)cg_index(e.draw(CG_SITE_ZERODAY, zocc++, 0), (uint32_t)cnt);
        raw = nth_bit32(mask, r);
      }
    }
    if (raw < 0 || raw >= P.t.X) continue;
    const uint8_t ebit = (uint8_t)(1u << raw);
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      int d = c * WAVE + e.lane;
      uint8_t f = d < M ? e.flags[d] : 0;
      uint8_t x = (uint8_t)(((f & CG_F_REACH) ? 1 : 0) | (((f & CG_F_KNOWN) && d < M && (e.vul[d] & ebit)) ? 2 : 0));
      T[d] = (((f & CG_F_COMP) ? 0u : T_TIME_INF) << 2) | x;
      cur[d] = d < M ? e.optr[d] : 0;
      uint64_t cm = ballot((x & 1) || ((x & 2) && !(f & CG_F_COMP)));
      if (e.lane == 0) cand[c] = cm;
    }
    wsync();
    SUBSTAMP(10);
    int n_rounds = 0;
    // fix-point rounds: a source re-examines its pick and, if an earlier source took it, resumes the scan
    for (int round = 0; round <= M + 1; ++round) {
      ++n_rounds;
      bool changed = round == 0;
      for (int b0 = 0; b0 < n_src; b0 += WAVE) {
        const int i = b0 + e.lane;
        bool coop = false;
        int s = 0, o0 = 0, o1 = 0, k0 = 0;
        uint8_t st = 0;
        if (i < n_src) {
          s = slist[i];
          o0 = e.optr[s]; o1 = e.optr[s + 1]; k0 = cur[s]; st = e.dst[s];
          const bool dc = st & CG_D_DC;
          int k = k0;
          if (COLD(xany && x_isout(e, s))) {   // row with added edges: handled by spread_x_round below
          } else if (o1 - o0 <= LONG_ROW) k = spread_scan_lane(e, T, s, dc, k0, o1);
          else if ((st & CG_D_FULLROW) && !dc) {
            if (!(k0 < o1 && !e.blocked(k0) && spread_ok(T, e.ocol[k0], s)))
              k = spread_scan_full(e, T, cand, s, round == 0 ? k0 : k0 + 1, o0, o1);
          } else {
            coop = round == 0 || (k0 < o1 && !(dc || spread_ok(T, e.ocol[k0], s)));
          }
          if (!coop && !(xany && x_isout(e, s))) {
            if (k != k0) { cur[s] = (uint16_t)k; changed = true; }
            if (k < o1 && (round == 0 || k != k0)) spread_take(T, e.ocol[k], s);
          }
        }
        uint64_t nm = ballot(coop);   // long rows that are not "full": wave-cooperative (re)scan
        while (nm) {
          const int src_lane = __builtin_ctzll(nm);
          nm &= nm - 1;
          const int ls = __shfl(s, src_lane), lo1 = __shfl(o1, src_lane), lk0 = __shfl(k0, src_lane);
          const int lst = __shfl((int)st, src_lane);
          int k = spread_scan_coop(e, T, ls, lst & CG_D_DC, round == 0 ? lk0 : lk0 + 1, lo1);
          if (k != lk0) changed = true;
          if (e.lane == 0) {
            cur[ls] = (uint16_t)k;
            if (k < lo1) spread_take(T, e.ocol[k], ls);
          }
        }
      }
      if constexpr (XE) { if (COLD(xany)) { if (spread_x_round(e, T, cur, slist, n_src, round)) changed = true; } }
      wsync();
      if (!__any(changed)) break;
    }
    SUBSTAMP(11);
    SUBVAL(15, n_rounds);
    // log entries of every source: unblocked out-entries up to and including its pick
    int total_new = 0;
    for (int b0 = 0; b0 < n_src; b0 += WAVE) {
      const int i = b0 + e.lane;
      int n = 0;
      if (i < n_src) {
        int s = slist[i];
        int o0 = e.optr[s], o1 = e.optr[s + 1];
        if (COLD(xany && x_isout(e, s))) {   // counted by spread_x_counts below
        } else {
          int end = cur[s] < o1 ? cur[s] + 1 : o1;
          n = (end - o0) - range_popc(e.blk, o0, end);
        }
      }
      cntv[i < Mp ? i : 0] = (uint16_t)n;
      total_new += wave_sum(n);
    }
    wsync();
    if constexpr (XE) { if (COLD(xany)) { total_new += spread_x_counts(e, cur, slist, cntv, n_src, ebit); wsync(); } }
    SUBSTAMP(12);
    // ring: only the last CG_LOG_RING entries (global order: source id, then row order) matter
    if (total_new > 0) {
      const uint32_t base = (uint32_t)e.log_total;
      const uint32_t end = base + (uint32_t)total_new;
      const uint32_t lo = end > CG_LOG_RING ? end - CG_LOG_RING : 0;
      uint32_t after = end;   // global index just past the current block of sources
      for (int b0 = ((n_src - 1) / WAVE) * WAVE; b0 >= 0 && after > lo; b0 -= WAVE) {
        const int i = b0 + e.lane;
        int n = i < n_src ? cntv[i] : 0;
        int incl = wave_incl_scan(n, e.lane);
        int blk_total = __shfl(incl, 63);
        uint32_t cbase = after - (uint32_t)blk_total;
        uint32_t off = cbase + (uint32_t)(incl - n);
        bool mine = n > 0 && off + (uint32_t)n > lo;
        int s = mine ? (int)slist[i] : 0;
        const bool xs = mine && xany && x_isout(e, s);
        bool is_long = mine && !xs && (e.optr[s + 1] - e.optr[s]) > LONG_ROW;
        if (COLD(xs)) {
          uint32_t idx = off;
          XWalk w; w.init(e, s);
          const int last = (int)cur[s] - (int)e.optr[s];
          for (int m = 0; !w.done() && m <= last; ++m) {
            const bool ex = w.at_extra(e);
            const int v = ex ? (int)w.vx : (int)e.ocol[w.k];
            if (!(ex ? x_blocked(e, w.j) : e.blocked(w.k))) {
              if (idx >= lo) { e.ring[2 * (idx % CG_LOG_RING)] = (uint16_t)s; e.ring[2 * (idx % CG_LOG_RING) + 1] = (uint16_t)v; }
              ++idx;
            }
            w.next(e, ex);
          }
        }
        if (mine && !is_long && !xs) {
          uint32_t idx = off;
          int o1 = e.optr[s + 1];
          int last = cur[s] < o1 ? (int)cur[s] : o1 - 1;
          for (int k = e.optr[s]; k <= last; ++k) {
            if (e.blocked(k)) continue;
            if (idx >= lo) { e.ring[2 * (idx % CG_LOG_RING)] = (uint16_t)s; e.ring[2 * (idx % CG_LOG_RING) + 1] = e.ocol[k]; }
            ++idx;
          }
        }
        uint64_t lm = ballot(is_long);
        while (lm) {
          int ll = __builtin_ctzll(lm);
          lm &= lm - 1;
          int ls = __shfl(s, ll);
          uint32_t idx0 = __shfl(off, ll);
          int o0 = e.optr[ls], o1 = e.optr[ls + 1];
          int last = cur[ls] < o1 ? (int)cur[ls] : o1 - 1;
          for (int k0 = o0; k0 <= last; k0 += WAVE) {
            int k = k0 + e.lane;
            bool p = k <= last && !e.blocked(k);
            uint64_t m = ballot(p);
            uint32_t idx = idx0 + (uint32_t)below(m);
            if (p && idx >= lo) { e.ring[2 * (idx % CG_LOG_RING)] = (uint16_t)ls; e.ring[2 * (idx % CG_LOG_RING) + 1] = e.ocol[k]; }
            idx0 += (uint32_t)__popcll(m);
          }
        }
        after = cbase;
      }
      e.log_total = (int)end;
      e.ring_dirty = true;
    }
    wsync();
    SUBSTAMP(13);
    // apply: compromise flags + DC attribution (:1163-1185)
    for (int d = e.lane; d < M; d += WAVE)
      if ((T[d] >> 2) != T_TIME_INF && (T[d] >> 2) != 0u) e.flags[d] |= CG_F_COMP;
    for (int i = e.lane; i < n_src; i += WAVE) {
      int s = slist[i];
      if (!(e.dst[s] & CG_D_DC)) continue;
      if (COLD(xany && x_isout(e, s))) continue;   // attributed while its log entries were counted
      if (cur[s] < e.optr[s + 1]) byte_or(e.cby, e.ocol[cur[s]], ebit);
    }
    wsync();
    SUBSTAMP(14);
  }
}

template <bool XE>
__device__ __forceinline__ void attacker_probe(Env& e, const uint64_t* srcb, double& cost) {
  int n_src = 0;
  for (int c = 0; c < e.MC; ++c) n_src += __popcll(srcb[c]);
  if (n_src == 0) return;
  int r = (int)cg_index(e.draw(CG_SITE_PROBE_SRC, 0, 0), (uint32_t)n_src);
  int s = -1;
  for (int c = 0; c < e.MC; ++c) {
    int k = __popcll(srcb[c]);
    if (r < k) { s = c * WAVE + nth_bit(srcb[c], r); break; }
    r -= k;
  }
  if (COLD(XE && x_cnt(e) > 0 && x_isout(e, s))) {   // row with added edges: uniform walk of the merged row
    XWalk w; w.init(e, s);
    int v = -1;
    while (!w.done()) {
      const bool ex = w.at_extra(e);
      const int vv = ex ? (int)w.vx : (int)e.ocol[w.k];
      if (!(ex ? x_blocked(e, w.j) : e.blocked(w.k)) && !(e.flags[vv] & CG_F_KNOWN)) { v = vv; break; }
      w.next(e, ex);
    }
    wsync();
    if (v >= 0) {
      if (e.lane == 0) e.flags[v] |= CG_F_KNOWN;
      cost += 0.1;
    }
    wsync();
    return;
  }
  const int o0 = e.optr[s], o1 = e.optr[s + 1];
  for (int k0 = o0; k0 < o1; k0 += WAVE) {
    int k = k0 + e.lane;
    bool p = k < o1 && !e.blocked(k) && !(e.flags[e.ocol[k]] & CG_F_KNOWN);
    uint64_t m = ballot(p);
    if (m) {
      int v = e.ocol[k0 + __builtin_ctzll(m)];
      wsync();
      if (e.lane == 0) e.flags[v] |= CG_F_KNOWN;
      cost += 0.1;   // :1199 (not scaled)
      break;
    }
  }
  wsync();
}

// ---------------- arrivals: CDSimulator.generate_workloads :244-348 ----------------
__device__ __forceinline__ void gen_workloads(Env& e, const KParams& P, int num, bool server, int n_active) {
  const int M = e.M, MC = e.MC;
  if (n_active <= 0) return;
  if (P.c.workload_cap >= 0 && num > P.c.workload_cap) num = P.c.workload_cap;
  if (num > n_active) num = n_active;
  if (num <= 0) return;
  const uint32_t site = server ? CG_SITE_ARR_SERVER : CG_SITE_ARR_CLIENT;
  uint32_t* key = e.scr;                     // [Mp]
  uint32_t* candb = e.scr + MC * WAVE;       // candidate bit per device as ballots [2*MC]
  int n = 0;
  for (int c = 0; c < MC; ++c) {
    int d = c * WAVE + e.lane;
    bool cand = false;
    if (d < M) {
      uint8_t f = e.flags[d];
      cand = !(f & CG_F_NYA) && e.wl[d] == 0 && e.busy[d] == 0 && (((e.dst[d] & CG_D_SERVER) != 0) == server);
    }
    uint64_t m = ballot(cand);
    if (e.lane == 0) { candb[2 * c] = (uint32_t)m; candb[2 * c + 1] = (uint32_t)(m >> 32); }
    n += __popcll(m);
  }
  wsync();
  if (n == 0) return;
  int k = num < n ? num : n;
  bool all = (k == n);
  uint32_t thr_key = 0xFFFFFFFFu;
  int n_less = 0;
  if (!all) {
    for (int c = 0; c < MC; ++c) {
      int d = c * WAVE + e.lane;
      uint64_t m = (uint64_t)candb[2 * c] | ((uint64_t)candb[2 * c + 1] << 32);
      if ((m >> e.lane) & 1ull) key[d] = e.draw(site, d, 0);
    }
    wsync();
    // radix select: value of the k-th smallest key
    uint32_t prefix = 0;
    int kk = k;   // 1-based rank still to locate among matching candidates
    for (int bit = 31; bit >= 0; --bit) {
      uint32_t hi_mask = bit == 31 ? 0u : (0xFFFFFFFFu << (bit + 1));
      int cnt0 = 0;
      for (int c = 0; c < MC; ++c) {
        int d = c * WAVE + e.lane;
        uint64_t m = (uint64_t)candb[2 * c] | ((uint64_t)candb[2 * c + 1] << 32);
        bool p = ((m >> e.lane) & 1ull) && ((key[d] & hi_mask) == prefix) && !((key[d] >> bit) & 1u);
        cnt0 += __popcll(ballot(p));
      }
      if (kk > cnt0) { kk -= cnt0; prefix |= (1u << bit); }
    }
    thr_key = prefix;
    for (int c = 0; c < MC; ++c) {
      int d = c * WAVE + e.lane;
      uint64_t m = (uint64_t)candb[2 * c] | ((uint64_t)candb[2 * c + 1] << 32);
      n_less += __popcll(ballot(((m >> e.lane) & 1ull) && key[d] < thr_key));
    }
  }
  // select: key < thr, plus the first (k - n_less) candidates with key == thr in id order
  int need_eq = k - n_less;
  int seen_eq = 0;
  for (int c = 0; c < MC; ++c) {
    int d = c * WAVE + e.lane;
    uint64_t m = (uint64_t)candb[2 * c] | ((uint64_t)candb[2 * c + 1] << 32);
    bool cand = (m >> e.lane) & 1ull;
    bool take = false;
    if (all) take = cand;
    else {
      bool eq = cand && key[d] == thr_key;
      uint64_t em = ballot(eq);
      take = (cand && key[d] < thr_key) || (eq && (seen_eq + below(em)) < need_eq);
      seen_eq += __popcll(em);
    }
    if (take) {
      e.wl[d] = (uint8_t)(1 + cg_cdf_lookup(e.draw(CG_SITE_ARR_TIME, d, 0), P.c.tri_thr, CG_TRI_TABLE));
      e.flags[d] &= (uint8_t)~CG_F_WLADV;
    }
  }
  wsync();
}

// volt_typhoon_env.py:575-596; the three counts come from the fused pass of the tick
__device__ __forceinline__ void arrivals(Env& e, const KParams& P, int step_num, int n_active, int idle, int free_s) {
  int free_c = idle - free_s;
  int n1 = n_active > 1 ? n_active : 1;
  int half = 0;
  while (4 * (half + 1) * (half + 1) <= n1) ++half;   // int(0.5*sqrt(n)) (:141-145)
  int period = P.c.workload_period_base + half;
  if (period < 10) period = 10;
  if (period > P.c.workload_period_max) period = P.c.workload_period_max;
  if (step_num % period != 0) return;
  if (n_active == 0 || 10 * idle < n_active) return;   // _idle_fraction() < 0.10
  int nC, nS;
  if (P.c.scaling_vulnerability) {   // _scaled_numloads(100, 10), anchor 50 (:266-293)
    int req_c = 2 * n_active;                       // round(100*n/50)
    int q = n_active / 5, r = n_active % 5;          // round(10*n/50) = round(n/5); no exact halves
    int req_s = q + (2 * r > 5 ? 1 : 0);
    if (req_c < 1) req_c = 1;
    if (req_s < 1) req_s = 1;
    int cap_c = free_c > 1 ? free_c : 1, cap_s = free_s > 1 ? free_s : 1;
    nC = req_c < cap_c ? req_c : cap_c;
    nS = req_s < cap_s ? req_s : cap_s;
  } else { nC = 100; nS = 10; }
  if (P.c.workload_cap > 0) {
    int total = nC + nS;
    if (total > P.c.workload_cap) {
      double ratio = (double)P.c.workload_cap / (double)total;
      nC = (int)(nC * ratio); if (nC < 0) nC = 0;
      nS = (int)(nS * ratio); if (nS < 0) nS = 0;
    }
  }
  gen_workloads(e, P, nC, false, n_active);
  gen_workloads(e, P, nS, true, n_active);
}

// ---------------- evolve_network: CyberDefenseEnv.py:583-875 ----------------
__device__ __forceinline__ int rank_select(const Env& e, uint8_t mask, uint8_t want, int r) {
  for (int c = 0; c < e.MC; ++c) {
    int d = c * WAVE + e.lane;
    uint64_t m = ballot(d < e.M && (e.flags[d] & mask) == want);
    int k = __popcll(m);
    if (r < k) return c * WAVE + nth_bit(m, r);
    r -= k;
  }
  return -1;
}
// g.get_eid(u, v, directed=True, error=False) != -1 over the base CSR and the env's added edges (uniform)
template <bool XE>
__device__ __forceinline__ bool edge_exists(const Env& e, int u, int v) {
  const int o0 = e.optr[u], o1 = e.optr[u + 1];
#pragma nounroll
  for (int k0 = o0; k0 < o1; k0 += WAVE) {
    const int k = k0 + e.lane;
    if (ballot(k < o1 && e.ocol[k] == v)) return true;
  }
  const int n = XE ? x_cnt(e) : 0;
  const uint32_t key = ((uint32_t)u << 16) | (uint32_t)v;
#pragma nounroll
  for (int j0 = 0; j0 < n; j0 += WAVE) {
    const int j = j0 + e.lane;
    if (ballot(j < n && e.xk[j] == key)) return true;
  }
  return false;
}

template <bool XE>
__device__ __forceinline__ void evolve(Env& e, const KParams& P) {
  const int M = e.M, MC = e.MC;
  if (!(e.eflags & CG_E_EVO_INIT)) {   // :654-659
    uint32_t* F = (uint32_t*)e.flags;
    for (int w = e.lane; w < (e.MS >> 2); w += WAVE) {
      uint32_t f = F[w];
      F[w] = (f & ~(ONES * CG_F_EVOACT)) | ((~(f >> 4) & ONES) << 5);
    }
    e.eflags |= CG_E_EVO_INIT;
    wsync();
  }
  int n_ev = 0;
  if (P.c.poisson_thr[0] < (1ull << 32))   // lambda_events == 0: the table says "always zero events"
    n_ev = cg_cdf_lookup(e.draw(CG_SITE_EVO_POISSON, 0, 0), P.c.poisson_thr, CG_POISSON_TABLE);
  bool any_new = false;
  uint32_t* newly = e.marks;   // bit per device
  if (n_ev > 0) {
    for (int i = e.lane; i <= (MC * WAVE) / 32; i += WAVE) newly[i] = 0;
    wsync();
  }
  const int floor_n = P.c.num_of_device > P.c.min_network_size ? P.c.num_of_device : P.c.min_network_size;
  for (int ev = 0; ev < n_ev; ++ev) {
    if (cg_bernoulli(e.draw(CG_SITE_EVO_COIN, ev, 0), P.c.p_add_thr)) {
      int n_in = 0;
      for (int c = 0; c < MC; ++c) {
        int d = c * WAVE + e.lane;
        n_in += __popcll(ballot(d < M && !(e.flags[d] & CG_F_EVOACT)));
      }
      if (n_in > 0) {
        int d = rank_select(e, CG_F_EVOACT, 0, (int)cg_index(e.draw(CG_SITE_EVO_PICK_IN, ev, 0), (uint32_t)n_in));
        bool att = cg_bernoulli(e.draw(CG_SITE_EVO_ATT, ev, 0), P.c.p_attacker_thr);
        const uint8_t f0 = e.flags[d];
        wsync();
        if (e.lane == 0) {
          uint8_t f = (uint8_t)((f0 & ~CG_F_NYA) | CG_F_EVOACT);
          if (att) f |= (CG_F_COMP | CG_F_OWNED | CG_F_KNOWN);
          e.flags[d] = f;
          newly[d >> 5] |= 1u << (d & 31);
        }
        if (att || (f0 & CG_F_OWNED)) e.eflags &= ~CG_E_STAR_OK;
        any_new = true;
      }
    } else {
      int n_act = 0;
      for (int c = 0; c < MC; ++c) {
        int d = c * WAVE + e.lane;
        n_act += __popcll(ballot(d < M && (e.flags[d] & CG_F_EVOACT)));
      }
      if (n_act > floor_n) {
        int d = rank_select(e, CG_F_EVOACT, CG_F_EVOACT, (int)cg_index(e.draw(CG_SITE_EVO_PICK_ACT, ev, 0), (uint32_t)n_act));
        const uint8_t f0 = e.flags[d];
        wsync();
        if (f0 & CG_F_OWNED) e.eflags &= ~CG_E_STAR_OK;
        if (e.lane == 0) {
          e.flags[d] = (uint8_t)((f0 | CG_F_NYA) & ~(CG_F_EVOACT | CG_F_WLADV));
          e.wl[d] = 0;
          e.busy[d] = 0;
          newly[d >> 5] &= ~(1u << (d & 31));
        }
      }
    }
    wsync();
  }
  bool changed = false;
  // star reconnection (:738-774): hub = first active attacker-owned device; missing hub<->owner links are
  // ADDED to the env's extra-edge list (a full list raises CG_E_TOPO_OVF and leaves the check pending).
  // Cold code: loops are kept rolled and every helper has one call site (instruction-cache footprint).
  if (!(e.eflags & CG_E_STAR_OK)) {
    int hub = rank_select(e, CG_F_OWNED | CG_F_EVOACT, CG_F_OWNED | CG_F_EVOACT, 0);
    bool ok = true;
    if (hub >= 0) {
#pragma nounroll
      for (int c = 0; c < MC; ++c) {
        int d = c * WAVE + e.lane;
        uint64_t m = ballot(d < M && d != hub && (e.flags[d] & (CG_F_OWNED | CG_F_EVOACT)) == (CG_F_OWNED | CG_F_EVOACT));
#pragma nounroll
        for (int it = 0; m; ++it) {   // two directed edges per owner: hub -> o, then o -> hub
          const int o = c * WAVE + __builtin_ctzll(m);
          const int u = (it & 1) ? o : hub, v = (it & 1) ? hub : o;
          if (it & 1) m &= m - 1;
          if (COLD(!edge_exists<XE>(e, u, v))) { if (XE && x_add(e, u, v)) changed = true; else { ok = false; e.eflags |= CG_E_TOPO_OVF; } }
        }
      }
    }
    if (ok) e.eflags |= CG_E_STAR_OK;
  }
  // preferential attachment of isolated newcomers (:776-843): one degree snapshot (after the star edges),
  // weights degree + 1 over the active devices in ascending id, r = random.uniform(0, total) (:817)
  if (any_new) {
    const int n0 = x_cnt(e);
    bool iso = false;
#pragma nounroll
    for (int c = 0; c < MC; ++c) {
      int d = c * WAVE + e.lane;
      if (d < M && ((newly[d >> 5] >> (d & 31)) & 1u) && !(e.flags[d] & (CG_F_NYA | CG_F_OWNED))) {
        int deg = (e.optr[d + 1] - e.optr[d]) + (e.iptr(d + 1) - e.iptr(d));
        if (deg < 1) iso = true;   // candidates only: the added edges are counted below
      }
    }
    if (!XE && __any(iso)) e.eflags |= CG_E_TOPO_OVF;   // no extra-edge list in this build of the kernel
    if (COLD(XE && __any(iso))) {
      uint32_t* cdf = e.scr;   // [Mp] inclusive weight sums
      int total = 0;
#pragma nounroll
      for (int c = 0; c < MC; ++c) {
        const int d = c * WAVE + e.lane;
        int w = 0;
        if (d < M && (e.flags[d] & CG_F_EVOACT)) {
          w = (e.optr[d + 1] - e.optr[d]) + (e.iptr(d + 1) - e.iptr(d)) + 1;
#pragma nounroll
          for (int j = 0; j < n0; ++j) { const uint32_t k = e.xk[j]; w += ((int)(k >> 16) == d) + ((int)(k & 0xFFFFu) == d); }
        }
        const int incl = wave_incl_scan(w, e.lane);
        cdf[d] = (uint32_t)(total + incl);
        total += __shfl(incl, 63);
      }
      wsync();
#pragma nounroll
      for (int c = 0; c < MC && total > 0; ++c) {
        const int d0 = c * WAVE + e.lane;
        uint64_t m = ballot(d0 < M && ((newly[d0 >> 5] >> (d0 & 31)) & 1u) && !(e.flags[d0] & (CG_F_NYA | CG_F_OWNED)) &&
                            (e.optr[d0 + 1] - e.optr[d0]) + (e.iptr(d0 + 1) - e.iptr(d0)) < 1);
#pragma nounroll
        while (m) {
          const int d = c * WAVE + __builtin_ctzll(m);
          m &= m - 1;
          // live degree (:809): an edge added earlier in this call may already touch d
          int deg = 0;
          const int n = x_cnt(e);
#pragma nounroll
          for (int j0 = 0; j0 < n; j0 += WAVE) {
            const int j = j0 + e.lane;
            const uint32_t k = j < n ? e.xk[j] : 0xFFFFFFFFu;
            deg += __popcll(ballot(j < n && ((int)(k >> 16) == d || (int)(k & 0xFFFFu) == d)));
          }
          if (deg >= 1) continue;
          const uint64_t r = (uint64_t)total * (uint64_t)e.draw(CG_SITE_EVO_PA, d, 0);
          int tgt = -1;
#pragma nounroll
          for (int c2 = 0; c2 < MC && tgt < 0; ++c2) {   // bisect_left(cdf, r) over the active devices
            const int a = c2 * WAVE + e.lane;
            const uint64_t hit = ballot(a < M && (e.flags[a] & CG_F_EVOACT) && ((uint64_t)cdf[a] << 32) >= r);
            if (hit) tgt = c2 * WAVE + __builtin_ctzll(hit);
          }
          if (tgt >= 0 && !edge_exists<XE>(e, d, tgt)) { if (x_add(e, d, tgt)) changed = true; }
        }
      }
    }
  }
  if (COLD(XE && changed)) {   // _rebuild_graph_cache (volt_typhoon_env.py:456-481) starts from an empty _blocked set
    for (int w = e.lane; w < P.t.EW; w += WAVE) { e.blk[w] = 0; e.bin[w] = 0; }
    for (int w = e.lane; w < P.t.KW; w += WAVE) e.xb[w] = 0;
    e.blk_dirty = true;
    x_masks(e);
  }
  wsync();
}

// ---------------- the tick ----------------
// MT: devices per env when known at compile time (64, 256: chunk loops unroll and their LDS latencies
// overlap), 0 = any M at run time.
// Per-wave LDS carve + pointer table of one env (must match wave_lds_bytes on the host).
struct WaveAux { uint64_t* srcb; int32_t* park; };
__device__ __forceinline__ WaveAux env_setup(Env& e, uint8_t* smem, const KParams& P, int M, int MC, int Mp, int MS,
                                             int wave, int lane, int env) {
  uint8_t* wb = smem + P.shared_lds + (size_t)wave * P.wave_lds;
  e.flags = wb; e.busy = wb + MS; e.wl = wb + 2 * MS; e.cby = wb + 3 * MS;
  e.scr = (uint32_t*)(wb + ((4 * MS + 15) & ~15));
  e.blk = e.scr + 2 * Mp;
  e.bin = e.blk + ((P.t.EW + 3) & ~3);
  e.ring = (uint16_t*)(e.bin + ((P.t.EW + 3) & ~3));
  e.marks = (uint32_t*)(e.ring + 2 * CG_LOG_RING);
  WaveAux x;
  x.srcb = (uint64_t*)(e.marks + ((Mp / 32 + 2) & ~1));
  e.lsrc = (uint16_t*)(x.srcb + MC);
  e.devl = (int16_t*)(e.lsrc + Mp);
  x.park = (int32_t*)(wb + P.wave_lds - 128);   // [16 i32 + 3 f64] per-env scalars between fused ticks
  e.xk = (uint32_t*)(wb + P.wave_lds - 128 - P.t.x_bytes);
  e.xb = e.xk + P.t.K;
  e.xmo = (uint64_t*)(e.xb + ((P.t.KW + 1) & ~1));
  e.xmi = e.xmo + MC;
  e.K = P.t.K;
  e.optr = (const uint16_t*)(smem + P.t.o_optr); e.ocol = (const uint16_t*)(smem + P.t.o_ocol);
  e.osv = (const float*)(smem + P.t.o_os); e.ver = (const float*)(smem + P.t.o_ver); e.ano = (const float*)(smem + P.t.o_ano);
  e.dst = smem + P.t.o_dst; e.vul = smem + P.t.o_vul; e.nap = smem + P.t.o_nap;
  e.iptr_g = (const uint16_t*)(P.t.blob + P.t.o_iptr); e.icol_g = (const uint16_t*)(P.t.blob + P.t.o_icol);
  e.ieid_g = (const uint16_t*)(P.t.blob + P.t.o_ieid); e.oeid_g = (const uint16_t*)(P.t.blob + P.t.o_oeid);
  e.M = M; e.MC = MC; e.MS = MS; e.lane = lane; e.env = env;
  e.cbits = 32 - __builtin_clz((unsigned)(4 * ((MS / 4 + WAVE - 1) / WAVE)));
  e.env_id = (uint32_t)(P.c.env_id_base + env);
  e.seed = P.c.seed;
  e.multi = P.t.multi != 0;
  e.stash = P.b.stash + (size_t)env * 4 * M;
  return x;
}

// FUSED: cygym_rollout (n_ticks > 1).  The per-env scalars are parked in LDS between ticks so that they are
// not loop-carried registers; the single-tick instantiation has a compile-time trip count of 1.
template <int WPB, int MT, bool FUSED, bool XE>
__global__ __launch_bounds__(WPB * WAVE, FUSED ? 4 : (XE ? CG_LB : 1)) void step_kernel(const KParams P0) {
  extern __shared__ __align__(16) uint8_t smem[];
  // every use below goes through `P`: the kernarg copy for the single-tick kernel, the device copy for the
  // fused one (so that it can be re-read, opaquely, at the top of every tick)
  const KParams* pk;
  if constexpr (FUSED) pk = P0.self; else pk = &P0;
#define P (*pk)
  const int M = MT ? MT : P.t.M, MC = MT ? (MT + WAVE - 1) / WAVE : P.t.MC, Mp = MC * WAVE, MS = (M + 3) & ~3;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int env = uni(blockIdx.x * WPB + wave);
  const bool live = env < P.n_envs;
  const int G = P.a.max_groups, L = P.a.max_devs;

  Env e;
  WaveAux aux = env_setup(e, smem, P, M, MC, Mp, MS, wave, lane, live ? env : 0);
  uint64_t* srcb = aux.srcb;
  int32_t* park = aux.park;
  e.env = env;
  e.blk_dirty = e.ring_dirty = e.x_dirty = false;

  STAMP(0);
  // ---- issue every global load of this tick up front (one memory latency, not a chain) ----
  const size_t so = (size_t)(live ? env : 0) * 4 * M;
  const uint8_t* g_live = P.b.live + so;
  int32_t ie[CG_I_COUNT];
  double fe[CG_D_COUNT];
  int mode = 0, ng = 0, at0 = 8, cnt0 = 0, nexp0 = 0, app0 = -1;
  uint4 r0 = make_uint4(0, 0, 0, 0);
  uint32_t ringw = 0;
  constexpr int PF_BLK = 2, PF_DEV = 1;   // words / list entries per lane prefetched into registers
  uint32_t bw[PF_BLK], bwi[PF_BLK];
  int16_t dv[PF_DEV];
  const bool vec = (M & 3) == 0;
  const int items = M >> 2;   // uint4 items of the [4][M] live block when M % 4 == 0
  if (live) {
    const int32_t* g = P.b.ienv + (size_t)env * CG_I_COUNT;
#pragma unroll
    for (int i = 0; i < CG_I_COUNT; ++i) ie[i] = g[i];
    const double* gf = P.b.fenv + (size_t)env * CG_D_COUNT;
#pragma unroll
    for (int i = 0; i < CG_D_COUNT; ++i) fe[i] = gf[i];
    mode = P.a.mode[env];
    ng = P.a.n_groups[env];
    at0 = P.a.atype[(size_t)env * G];
    cnt0 = P.a.dev_cnt[(size_t)env * G];
    nexp0 = P.a.n_exploit[(size_t)env * G];
    app0 = P.a.app[(size_t)env * G];
    if (vec && lane < items) r0 = ((const uint4*)g_live)[lane];
    if (lane < CG_LOG_RING) ringw = ((const uint32_t*)(P.b.ring + (size_t)env * CG_LOG_RING * 2))[lane];
#pragma unroll
    for (int j = 0; j < PF_BLK; ++j) {
      int w = lane + j * WAVE;
      bw[j] = w < P.t.EW ? P.b.blocked[(size_t)env * P.t.EW + w] : 0u;
      bwi[j] = w < P.t.EW ? P.b.blocked_in[(size_t)env * P.t.EW + w] : 0u;
    }
#pragma unroll
    for (int j = 0; j < PF_DEV; ++j) { int q = lane + j * WAVE; dv[j] = q < L ? P.a.dev_idx[(size_t)env * L + q] : (int16_t)0; }
  }
  // ---- workgroup-shared topology blob -> LDS: every 16-byte load is issued before the first store ----
  {
    const uint4* src = (const uint4*)P.t.blob;
    uint4* dstp = (uint4*)smem;
    const int n16 = P.t.lds_bytes >> 4, stride = WPB * WAVE;
    constexpr int PF_BLOB = 4;
    uint4 br[PF_BLOB];
#pragma unroll
    for (int j = 0; j < PF_BLOB; ++j) { const int i = threadIdx.x + j * stride; br[j] = src[i < n16 ? i : n16 - 1]; }   // unconditional: stays in registers
#pragma unroll
    for (int j = 0; j < PF_BLOB; ++j) { const int i = threadIdx.x + j * stride; if (i < n16) dstp[i] = br[j]; }
    for (int i = threadIdx.x + PF_BLOB * stride; i < n16; i += stride) dstp[i] = src[i];
  }
  if (live) {
    if (vec) {
      if (lane < items) ((uint4*)e.flags)[lane] = r0;
      for (int i = lane + WAVE; i < items; i += WAVE) ((uint4*)e.flags)[i] = ((const uint4*)g_live)[i];
    } else {
      for (int pl = 0; pl < 4; ++pl)
        for (int i = lane; i < MS; i += WAVE) e.flags[pl * MS + i] = i < M ? g_live[pl * M + i] : (pl == 0 ? (uint8_t)CG_F_NYA : (uint8_t)0);
    }
    if (lane < CG_LOG_RING) ((uint32_t*)e.ring)[lane] = ringw;
    const uint32_t* gb = P.b.blocked + (size_t)env * P.t.EW;
#pragma unroll
    for (int j = 0; j < PF_BLK; ++j) { int w = lane + j * WAVE; if (w < P.t.EW) { e.blk[w] = bw[j]; e.bin[w] = bwi[j]; } }
    for (int w = lane + PF_BLK * WAVE; w < P.t.EW; w += WAVE) { e.blk[w] = gb[w]; e.bin[w] = P.b.blocked_in[(size_t)env * P.t.EW + w]; }
    const int16_t* gd = P.a.dev_idx + (size_t)env * L;
#pragma unroll
    for (int j = 0; j < PF_DEV; ++j) { int q = lane + j * WAVE; if (q < L) e.devl[q] = dv[j]; }
    for (int q = lane + PF_DEV * WAVE; q < L; q += WAVE) e.devl[q] = gd[q];
  }
  __syncthreads();   // the only workgroup barrier: waves diverge per env from here on
  if (!live) return;
  STAMP(1);

  const int NW = MS >> 2;

  // ---- ticks of this launch: 1 for cygym_step, T for cygym_rollout (state stays in LDS / registers;
  // no cross-env synchronisation between ticks) ----
  const int n_ticks = FUSED ? P.n_ticks : 1;
  for (int tk = 0; tk < n_ticks; ++tk) {
  const size_t te = (size_t)tk * P.n_envs + env;   // row of this (tick, env) in the action / output arrays
  if (FUSED && tk > 0) {   // tick 0's header and list were prefetched with the state
    // Re-derive everything uniform from the device copy of the parameters: keeping ~200 loop-invariant
    // scalars alive across the tick body would spill SGPRs into VGPRs and halve the occupancy.
    {
      const uint64_t pv = (uint64_t)P0.self;
      uint32_t plo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
      uint32_t phi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pv >> 32));
      asm volatile("" : "+s"(plo), "+s"(phi));   // opaque: nothing derived from it is hoisted out of the tick loop
      pk = (const KParams*)(((uint64_t)phi << 32) | plo);
    }
    aux = env_setup(e, smem, P, M, MC, Mp, MS, wave, lane, env);
    srcb = aux.srcb; park = aux.park;
#pragma unroll
    for (int i = 0; i < CG_I_COUNT; ++i) ie[i] = park[i];
#pragma unroll
    for (int i = 0; i < CG_D_COUNT; ++i) fe[i] = ((const double*)(park + CG_I_COUNT))[i];
    mode = P.a.mode[te];
    ng = P.a.n_groups[te];
    at0 = P.a.atype[te * G];
    cnt0 = P.a.dev_cnt[te * G];
    nexp0 = P.a.n_exploit[te * G];
    app0 = P.a.app[te * G];
    const int16_t* gd = P.a.dev_idx + te * L;
    for (int q = lane; q < L; q += WAVE) e.devl[q] = gd[q];
    wsync();
  }
  if (ng < 0) {   // n_groups < 0: this env does not tick (per-env stepping inside a batch)
    if (FUSED && tk == 0 && lane == 0) {
#pragma unroll
      for (int i = 0; i < CG_I_COUNT; ++i) park[i] = ie[i];
#pragma unroll
      for (int i = 0; i < CG_D_COUNT; ++i) ((double*)(park + CG_I_COUNT))[i] = fe[i];
    }
    continue;
  }
  const int16_t* devs = e.devl;
  uint32_t* const F = (uint32_t*)e.flags;
  uint32_t* const Bz = (uint32_t*)e.busy;
  uint32_t* const Wl = (uint32_t*)e.wl;
  const uint32_t* const Ds = (const uint32_t*)e.dst;
  const bool partial = (mode & CG_MODE_PARTIAL) && ng == 0;   // step(action, agent_cnt != len(net)) :1207
  mode &= 0xFF;
  e.tick = (uint32_t)ie[CG_I_RNG_TICK];
  e.eflags = ie[CG_I_FLAGS];
  e.log_total = ie[CG_I_LOG_TOTAL];
  if (COLD(XE && (!FUSED || tk == 0) && x_cnt(e) > 0)) {   // this env carries edges evolve_network added: stage its list
    const uint32_t* xg = P.b.extra + (size_t)env * (P.t.K + P.t.KW);
    const int nx = x_cnt(e) < P.t.K ? x_cnt(e) : P.t.K;
    for (int j = lane; j < nx; j += WAVE) e.xk[j] = xg[j];
    for (int w = lane; w < P.t.KW; w += WAVE) e.xb[w] = xg[P.t.K + w];
    wsync();
    x_masks(e);
  }
  double cost = 0.0;
  bool dirty = false;
  int last_atype = -1;

  if (ng == 0) {   // ---- step(action) volt_typhoon_env.py:818-1333 ----
    int at = at0;
    int Ld = cnt0;
    if (Ld > L) Ld = L;
    if (Ld < 0) Ld = 0;
    if (mode == CG_MODE_DEFENDER) { if (!(at >= 0 && at < P.c.n_def_actions)) at = 8; }
    else                          { if (!(at >= 0 && at < P.c.n_att_actions)) at = 3; }
    for (int w = lane; w < NW; w += WAVE) {   // :904-908 decay of the cached busy set
      uint32_t b = Bz[w];
      Bz[w] = b - (((F[w] >> 6) & ONES) & nz01(b));
    }
    wsync();
    if (mode == CG_MODE_DEFENDER) {
      if (P.c.baseline != 0) at = 8;   // :913-914
      def_global(e, P, at, devs, Ld, cost, dirty, false, ie, fe);
      if (at == 1 || at == 4 || at == 5 || at == 6 || at == 7 || at == 9 || at == 12 || at == 13)
        if (Ld > 0) def_per_device<XE>(e, P, at, devs, Ld, app0, cost, dirty, ie, fe);
    } else if (P.c.baseline != 3 && (at == 1 || at == 2)) {
#pragma unroll
      for (int c = 0; c < MC; ++c) {   // :1127 snapshot of the sources
        int d = c * WAVE + lane;
        uint64_t m = ballot(d < M && (e.flags[d] & (CG_F_COMP | CG_F_OWNED)));
        if (lane == 0) srcb[c] = m;
      }
      wsync();
      if (at == 1) {
        int ne = nexp0;
        if (ne > CG_MAX_EXPLOITS) ne = CG_MAX_EXPLOITS;
        __builtin_amdgcn_s_setprio(3);   // the spread bounds the launch: win issue arbitration over short envs
        attacker_spread<XE>(e, P, P.a.exploit + te * G * CG_MAX_EXPLOITS, ne, srcb);
        __builtin_amdgcn_s_setprio(0);
      } else {
        attacker_probe<XE>(e, srcb, cost);
      }
    }
    last_atype = at;
  } else {   // ---- step_grouped(groups) :694-779 via _step_apply_only :612-692 ----
    uint8_t* occ = (uint8_t*)(e.scr + Mp);   // second scratch half: clean-stall occurrence numbers
    for (int i = lane; i < Mp / 4; i += WAVE) ((uint32_t*)occ)[i] = 0;
    wsync();
    const int16_t* dp = devs;
    int used = 0;
    for (int g = 0; g < ng && g < G; ++g) {
      int at = P.a.atype[te * G + g];
      int Ld = P.a.dev_cnt[te * G + g];
      if (Ld < 0) Ld = 0;
      if (used + Ld > L) Ld = L - used;
      if (mode == CG_MODE_DEFENDER && at == 0) at = 8;
      else if (mode == CG_MODE_ATTACKER && at == 0) at = 3;
      if (mode == CG_MODE_DEFENDER) {
        if (P.c.baseline != 0) at = 8;
        def_global(e, P, at, dp, Ld, cost, dirty, true, ie, fe);
        if (at == 1 && Ld > 0) def_clean(e, P, dp, Ld, cost, ie, fe, occ);
      }
      dp += Ld; used += Ld;
    }
    for (int w = lane; w < NW; w += WAVE) { uint32_t b = Bz[w]; Bz[w] = b - nz01(b); }   // _tick_busy_time_once :607
    wsync();
  }

  STAMP(2);
  // ---- fused word pass: workload advance (:1242-1261 / :705-725) + every per-tick count ----
  int c_fin = 0, c_act = 0, c_idle = 0, c_fsrv = 0, c_comp = 0, c_cdc = 0;
  for (int w = lane; w < NW; w += WAVE) {
    uint32_t f = F[w], b = Bz[w], l = Wl[w], st = Ds[w];
    const uint32_t nya = (f >> 4) & ONES;
    const uint32_t step = partial ? 0u : (~nz01(b) & ~nya & nz01(l) & ONES);   // idle-of-stall, active, has a job
    l -= step;
    const uint32_t fin = step & ~nz01(l);
    const uint32_t adv = (f >> 7) & ONES;
    f &= ~((fin & adv) << 7);
    Wl[w] = l; F[w] = f;
    const uint32_t act = ~nya & ONES;
    const uint32_t idl = act & ~nz01(b) & ~nz01(l);
    const uint32_t cmp = f & ~nya & ~(f >> 1) & ONES;          // compromised, active, not attacker-owned
    c_fin += __popc(fin & ~adv);
    c_act += __popc(act);
    c_idle += __popc(idl);
    c_fsrv += __popc(idl & (st >> 1));
    c_comp += __popc(cmp);
    c_cdc += __popc(cmp & st);
  }
  const int current_work = wave_sum_bits(c_fin, e.cbits);
  const int n_active = wave_sum_bits(c_act, e.cbits);
  const int n_idle = wave_sum_bits(c_idle, e.cbits);
  const int n_fsrv = wave_sum_bits(c_fsrv, e.cbits);
  const int n_comp = wave_sum_bits(c_comp, e.cbits);
  const int n_comp_dc = wave_sum_bits(c_cdc, e.cbits);
  ie[CG_I_WORK_DONE] += current_work;
  wsync();
  if (!partial) arrivals(e, P, ie[CG_I_STEP_NUM], n_active, n_idle, n_fsrv);   // changes wl only: the counts above stand

  STAMP(3);
  // ---- rewards (:1267-1304 / :732-748) ----
  if (ng == 0) ie[CG_I_COMP_CNT] += n_comp;
  ie[CG_I_LAST_NCOMP] = n_comp;
  double raw, shaped;
  {
    double def_work = P.c.work_scale * current_work;
    if (mode == CG_MODE_DEFENDER) {
      raw = cost + def_work - n_comp * P.c.comp_scale;
      shaped = raw;
    } else {
      double r = cost + P.c.comp_scale * (n_comp + 10 * n_comp_dc);
      double phi = (double)n_comp / (double)M;
      if (!(e.eflags & CG_E_PREV_SET)) { fe[CG_D_PREV_ATT_POT] = phi; e.eflags |= CG_E_PREV_SET; }
      double inc = P.c.gamma * phi - fe[CG_D_PREV_ATT_POT];
      double bonus = 0.1 * inc + 0.0;
      fe[CG_D_PREV_ATT_POT] = P.c.gamma * phi;
      raw = r;
      shaped = r + bonus;
    }
  }

  STAMP(4);
  // ---- observation (_get_state CyberDefenseEnv.py:146-191), before evolve.
  // One lane per device PAIR: 12 floats = three 16-byte stores; static columns read as float2.
  if (!(M & 1)) {
    float4* out4 = (float4*)(P.o.obs + te * M * 6);
    const int npairs = M >> 1;
    const uint16_t* F2 = (const uint16_t*)e.flags;
    const float2* os2 = (const float2*)e.osv;
    const float2* ve2 = (const float2*)e.ver;
    const float2* an2 = (const float2*)e.ano;
    for (int p = lane; p < npairs; p += WAVE) {
      const uint32_t f2 = F2[p];
      const float2 o = os2[p], v = ve2[p], a = an2[p];
      const uint32_t fa = f2 & 0xFFu, fb = f2 >> 8;
      out4[3 * p + 0] = make_float4(o.x, v.x, (float)(fa & 1u), a.x);
      out4[3 * p + 1] = make_float4((float)((fa >> 2) & 1u), (float)((fa >> 4) & 1u), o.y, v.y);
      out4[3 * p + 2] = make_float4((float)(fb & 1u), a.y, (float)((fb >> 2) & 1u), (float)((fb >> 4) & 1u));
    }
  } else {   // odd M: rows are not 16-byte aligned across envs
    float* o = P.o.obs + te * M * 6;
    for (int d = lane; d < M; d += WAVE) {
      const uint32_t f = e.flags[d];
      o[6 * d + 0] = e.osv[d]; o[6 * d + 1] = e.ver[d]; o[6 * d + 2] = (float)(f & 1u); o[6 * d + 3] = e.ano[d];
      o[6 * d + 4] = (float)((f >> 2) & 1u); o[6 * d + 5] = (float)((f >> 4) & 1u);
    }
  }

  STAMP(5);
  if (!partial) {   // :1307-1312
    ie[CG_I_STEP_NUM] += 1;
    if (mode == CG_MODE_ATTACKER) ie[CG_I_ATT_STEP] += 1; else ie[CG_I_DEF_STEP] += 1;
  }
  const bool done = ie[CG_I_STEP_NUM] > P.c.episode_limit;
  if (dirty || (ie[CG_I_STEP_NUM] % P.c.evolve_period) == 0) evolve<XE>(e, P);
  if (ng == 0) {   // :1330 rebuild of the cached busy set
    for (int w = lane; w < NW; w += WAVE) F[w] = (F[w] & ~(ONES * CG_F_BUSYC)) | (nz01(Bz[w]) << 6);
  }
  wsync();
  ie[CG_I_RNG_TICK] += 1;
  ie[CG_I_LAST_ATYPE] = last_atype;
  ie[CG_I_LOG_TOTAL] = e.log_total;
  ie[CG_I_FLAGS] = e.eflags | (__any(e.eflags & CG_E_BUSY_SAT) ? CG_E_BUSY_SAT : 0);

  if (lane == 0) {
    P.o.raw[te] = raw;
    P.o.shaped[te] = shaped;
    P.o.done[te] = done ? 1 : 0;
  }

  if (done && P.c.auto_reset && P.snap.live) {   // reload the initial snapshot; the RNG tick stays monotone
    const int si = P.snap.n_envs == 1 ? 0 : env;
    const size_t ss = (size_t)si * 4 * M;
    wsync();
    if (vec) {
      for (int i = lane; i < items; i += WAVE) ((uint4*)e.flags)[i] = ((const uint4*)(P.snap.live + ss))[i];
    } else {
      for (int pl = 0; pl < 4; ++pl)
        for (int i = lane; i < M; i += WAVE) e.flags[pl * MS + i] = P.snap.live[ss + pl * M + i];
    }
    for (int i = lane; i < 4 * M; i += WAVE) P.b.stash[so + i] = P.snap.stash[ss + i];
    for (int w = lane; w < P.t.EW; w += WAVE) {
      e.blk[w] = P.snap.blocked[(size_t)si * P.t.EW + w];
      e.bin[w] = P.snap.blocked_in[(size_t)si * P.t.EW + w];
    }
    if (lane < CG_LOG_RING) ((uint32_t*)e.ring)[lane] = ((const uint32_t*)(P.snap.ring + (size_t)si * CG_LOG_RING * 2))[lane];
    e.blk_dirty = e.ring_dirty = true;
    if (COLD(XE && P.t.K > 0)) {   // the snapshot's extra-edge list (normally empty) replaces the episode's
      const int ns = P.snap.extra ? (int)((uint32_t)P.snap.ienv[(size_t)si * CG_I_COUNT + CG_I_FLAGS] >> CG_E_NX_SHIFT) : 0;
      const uint32_t* xs = P.snap.extra + (size_t)si * (P.t.K + P.t.KW);
      for (int j = lane; j < ns; j += WAVE) e.xk[j] = xs[j];
      for (int w = lane; w < P.t.KW; w += WAVE) e.xb[w] = ns > 0 ? xs[P.t.K + w] : 0u;
      e.eflags = (e.eflags & 0xFFFF) | (ns << CG_E_NX_SHIFT);
      wsync();
      x_masks(e);
      e.x_dirty = true;
    }
    const int32_t keep_tick = ie[CG_I_RNG_TICK];
    const int32_t* g = P.snap.ienv + (size_t)si * CG_I_COUNT;
#pragma unroll
    for (int i = 0; i < CG_I_COUNT; ++i) ie[i] = g[i];
    ie[CG_I_RNG_TICK] = keep_tick;
    if (!(P.t.K > 0 && P.snap.extra)) ie[CG_I_FLAGS] &= 0xFFFF;
    const double* gf = P.snap.fenv + (size_t)si * CG_D_COUNT;
#pragma unroll
    for (int i = 0; i < CG_D_COUNT; ++i) fe[i] = gf[i];
    wsync();
  }
  if (FUSED && tk + 1 < n_ticks) {   // park the scalars for the next tick
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < CG_I_COUNT; ++i) park[i] = ie[i];
#pragma unroll
      for (int i = 0; i < CG_D_COUNT; ++i) ((double*)(park + CG_I_COUNT))[i] = fe[i];
    }
    wsync();
  }
  }   // for tk

  STAMP(6);
  // ---- write back: the whole [4][M] live block with 16-byte stores ----
  if (vec) {
    for (int i = lane; i < items; i += WAVE) ((uint4*)(P.b.live + so))[i] = ((const uint4*)e.flags)[i];
  } else {
    for (int pl = 0; pl < 4; ++pl)
      for (int i = lane; i < M; i += WAVE) P.b.live[so + pl * M + i] = e.flags[pl * MS + i];
  }
  if (e.blk_dirty)
    for (int w = lane; w < P.t.EW; w += WAVE) {
      P.b.blocked[(size_t)env * P.t.EW + w] = e.blk[w];
      P.b.blocked_in[(size_t)env * P.t.EW + w] = e.bin[w];
    }
  if (e.ring_dirty && lane < CG_LOG_RING)
    ((uint32_t*)(P.b.ring + (size_t)env * CG_LOG_RING * 2))[lane] = ((const uint32_t*)e.ring)[lane];
  if (COLD(XE && e.x_dirty)) {
    uint32_t* xg = P.b.extra + (size_t)env * (P.t.K + P.t.KW);
    const int nx = (int)((uint32_t)ie[CG_I_FLAGS] >> CG_E_NX_SHIFT);
    for (int j = lane; j < nx; j += WAVE) xg[j] = e.xk[j];
    for (int w = lane; w < P.t.KW; w += WAVE) xg[P.t.K + w] = e.xb[w];
  }
  if (lane == 0) {
    int32_t* g = P.b.ienv + (size_t)env * CG_I_COUNT;
#pragma unroll
    for (int i = 0; i < CG_I_COUNT; ++i) g[i] = ie[i];
    double* gf = P.b.fenv + (size_t)env * CG_D_COUNT;
#pragma unroll
    for (int i = 0; i < CG_D_COUNT; ++i) gf[i] = fe[i];
  }
  STAMP(7);
#ifdef CG_STAMPS
  if (P.dbg && lane == 0) { P.dbg[(size_t)env * 16 + 8] = (unsigned long long)(long long)ie[CG_I_LAST_ATYPE]; P.dbg[(size_t)env * 16 + 9] = (unsigned long long)mode; }
#endif
#undef P
}

// ---------------- reset / randomize / observe / action script ----------------
__global__ void reset_kernel(KParams P, const int32_t* env_ids, int n) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n) return;
  const int env = env_ids ? env_ids[wave] : wave;
  if (env < 0 || env >= P.n_envs) return;
  const int M = P.t.M;
  const int si = P.snap.n_envs == 1 ? 0 : env;
  const size_t so = (size_t)env * 4 * M, ss = (size_t)si * 4 * M;
  int32_t tick = P.b.ienv[(size_t)env * CG_I_COUNT + CG_I_RNG_TICK];
  for (int i = lane; i < 4 * M; i += WAVE) {
    P.b.live[so + i] = P.snap.live[ss + i];
    P.b.stash[so + i] = P.snap.stash[ss + i];
  }
  for (int w = lane; w < P.t.EW; w += WAVE) {
    P.b.blocked[(size_t)env * P.t.EW + w] = P.snap.blocked[(size_t)si * P.t.EW + w];
    P.b.blocked_in[(size_t)env * P.t.EW + w] = P.snap.blocked_in[(size_t)si * P.t.EW + w];
  }
  if (lane < CG_LOG_RING)
    ((uint32_t*)(P.b.ring + (size_t)env * CG_LOG_RING * 2))[lane] = ((const uint32_t*)(P.snap.ring + (size_t)si * CG_LOG_RING * 2))[lane];
  const int XW = P.t.K + P.t.KW;
  if (P.t.K > 0 && P.snap.extra)
    for (int w = lane; w < XW; w += WAVE) P.b.extra[(size_t)env * XW + w] = P.snap.extra[(size_t)si * XW + w];
  if (lane < CG_I_COUNT) {
    int32_t v = P.snap.ienv[(size_t)si * CG_I_COUNT + lane];
    if (lane == CG_I_RNG_TICK) v = tick;   // the draw counter is monotone across episodes
    if (lane == CG_I_FLAGS && !(P.t.K > 0 && P.snap.extra)) v &= 0xFFFF;   // no extra-edge list to restore
    P.b.ienv[(size_t)env * CG_I_COUNT + lane] = v;
  }
  if (lane < CG_D_COUNT) P.b.fenv[(size_t)env * CG_D_COUNT + lane] = P.snap.fenv[(size_t)si * CG_D_COUNT + lane];
}

// randomize_compromise_and_ownership volt_typhoon_env.py:330-383; wave per env, global memory only.
__global__ void randomize_kernel(KParams P, const int32_t* env_ids, int n, uint32_t* keybuf) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n) return;
  const int env = env_ids ? env_ids[wave] : wave;
  if (env < 0 || env >= P.n_envs) return;
  const int M = P.t.M, MC = P.t.MC;
  uint8_t* flags = P.b.live + (size_t)env * 4 * M;
  int32_t* ie = P.b.ienv + (size_t)env * CG_I_COUNT;
  const uint32_t tick = (uint32_t)ie[CG_I_RNG_TICK];
  const uint32_t env_id = (uint32_t)(P.c.env_id_base + env);
  uint32_t* key = keybuf + (size_t)wave * MC * WAVE;
  int cnt = 0, k_owned = 0, k_comp = 0;
  for (int c = 0; c < MC; ++c) {
    int d = c * WAVE + lane;
    bool el = d < M && !(flags[d] & CG_F_NYA) && !(P.t.dstatic[d] & CG_D_DC);
    if (d < MC * WAVE) key[d] = el ? cg_draw(P.c.seed, env_id, tick, CG_SITE_SHUFFLE, d, 0) : 0u;
    cnt += __popcll(ballot(el));
    k_owned += __popcll(ballot(el && (flags[d] & CG_F_OWNED)));
    k_comp += __popcll(ballot(el && (flags[d] & CG_F_COMP)));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (lane == 0) ie[CG_I_RNG_TICK] = (int32_t)(tick + 1);
  if (lane == 0) ie[CG_I_FLAGS] &= ~CG_E_STAR_OK;
  if (cnt == 0 || (k_owned == 0 && k_comp == 0)) return;
  int extra = k_comp - k_owned; if (extra < 0) extra = 0;
  for (int c = 0; c < MC; ++c) {
    int d = c * WAVE + lane;
    bool el = d < M && !(flags[d] & CG_F_NYA) && !(P.t.dstatic[d] & CG_D_DC);
    if (!el) continue;
    uint32_t kd = key[d];
    int rank = 0;
    for (int o = 0; o < M; ++o) {
      bool eo = !(flags[o] & CG_F_NYA) && !(P.t.dstatic[o] & CG_D_DC);
      uint32_t ko = key[o];
      rank += (eo && (ko < kd || (ko == kd && o < d))) ? 1 : 0;
    }
    uint8_t f = (uint8_t)(flags[d] & ~(CG_F_OWNED | CG_F_COMP | CG_F_KNOWN));
    if (rank < k_owned) f |= (CG_F_OWNED | CG_F_COMP | CG_F_KNOWN);
    else if (rank < k_owned + extra) f |= (CG_F_COMP | CG_F_KNOWN);
    // flags are rewritten after every lane has read the eligibility bits (NYA/DC do not change)
    flags[d] = f;
  }
}

// blocked_in[j] = blocked[in_eid[j]]: the derived in-order mirror of the blocked bits (wave per env)
__global__ void derive_kernel(KParams P, cygym_buffers bufs) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= bufs.n_envs) return;
  const uint32_t* blk = bufs.blocked + (size_t)wave * P.t.EW;
  uint32_t* bin = bufs.blocked_in + (size_t)wave * P.t.EW;
  for (int j0 = 0; j0 < P.t.EW * 32; j0 += WAVE) {
    const int j = j0 + lane;
    bool bit = false;
    if (j < P.t.E) { const int k = P.t.in_eid[j]; bit = (blk[k >> 5] >> (k & 31)) & 1u; }
    const uint64_t m = __ballot(bit);
    if (lane == 0) {
      bin[j0 >> 5] = (uint32_t)m;
      if ((j0 >> 5) + 1 < P.t.EW) bin[(j0 >> 5) + 1] = (uint32_t)(m >> 32);
    }
  }
}

// role views: CyberDefenseEnv.py:146-257
__global__ void observe_kernel(KParams P, int role, float* out) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= P.n_envs) return;
  const int env = wave, M = P.t.M;
  const uint8_t* flags = P.b.live + (size_t)env * 4 * M;
  if (role == 0 || role == 1) {
    float* o = out + (size_t)env * 6 * M;
    for (int i = lane; i < 6 * M; i += WAVE) {
      int d = i / 6, col = i - d * 6;
      uint8_t f = flags[d];
      float x = col == 0 ? P.t.os_val[d] : col == 1 ? P.t.version[d] : col == 2 ? ((f & CG_F_COMP) ? 1.f : 0.f)
              : col == 3 ? P.t.anomaly[d] : col == 4 ? ((f & CG_F_KNOWN) ? 1.f : 0.f) : ((f & CG_F_NYA) ? 1.f : 0.f);
      if (role == 1 && ((f & CG_F_NYA) || !(f & CG_F_OWNED) || col == 2)) x = -1.f;
      o[i] = x;
    }
  } else {
    const int W = 4 * M + P.c.max_exploits;
    float* o = out + (size_t)env * W;
    for (int i = lane; i < W; i += WAVE) {
      float x;
      if (i < 4 * M) {
        int d = i >> 2, col = i & 3;
        uint8_t f = flags[d];
        bool vis = (f & CG_F_KNOWN) && !(f & CG_F_NYA) && (f & CG_F_OWNED);
        x = !vis ? -1.f : col == 0 ? P.t.os_val[d] : col == 1 ? P.t.version[d]
          : col == 2 ? ((f & CG_F_COMP) ? 1.f : 0.f) : ((f & CG_F_KNOWN) ? 1.f : 0.f);
      } else {
        x = (i - 4 * M) < P.t.X ? 1.f : 0.f;
      }
      o[i] = x;
    }
  }
}

// Synthetic action script of bench.py: alternating defender / attacker turns.
// Mirrored in numpy by cygym_amd/actions.py (tests check equality).
__global__ void gen_actions_kernel(KParams P, int tick, int32_t* mode, int32_t* n_groups, int32_t* atype,
                                   int32_t* n_exploit, int32_t* exploit, int32_t* app, int32_t* dev_cnt,
                                   int16_t* dev_idx, int max_devs) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= P.n_envs) return;
  const int M = P.t.M;
  const uint32_t env_id = (uint32_t)(P.c.env_id_base + env);
  cg_u32x4 r = cg_philox4x32_10(env_id, (uint32_t)tick, CG_SITE_ACTGEN, 0u, (uint32_t)P.c.seed, (uint32_t)(P.c.seed >> 32));
  const int m = tick & 1;
  mode[env] = m;
  n_groups[env] = 0;
  n_exploit[env] = 1;
  for (int j = 0; j < CG_MAX_EXPLOITS; ++j) exploit[(size_t)env * CG_MAX_EXPLOITS + j] = -1;
  exploit[(size_t)env * CG_MAX_EXPLOITS] = (int)cg_index(r.v[1], (uint32_t)(P.t.X > 0 ? P.t.X : 1));
  app[env] = (int)cg_index(r.v[2], 4u);
  int k = 0;
  if (m == CG_MODE_DEFENDER) {
    const int types[11] = {1, 4, 5, 6, 7, 8, 9, 11, 12, 13, 2};
    atype[env] = types[cg_index(r.v[0], 11u)];
    int kmax = M / 8 > 1 ? M / 8 : 1;
    if (kmax > max_devs) kmax = max_devs;
    k = 1 + (int)cg_index(r.v[3], (uint32_t)kmax);
    cg_u32x4 q = cg_philox4x32_10(env_id, (uint32_t)tick, CG_SITE_ACTGEN, 1u, (uint32_t)P.c.seed, (uint32_t)(P.c.seed >> 32));
    int a = (int)cg_index(q.v[0], (uint32_t)M);
    int stride = ((M & (M - 1)) == 0 && M > 1) ? (int)(2u * cg_index(q.v[1], (uint32_t)(M / 2)) + 1u) : 1;
    for (int j = 0; j < k; ++j) dev_idx[(size_t)env * max_devs + j] = (int16_t)((a + (long long)j * stride) % M);
  } else {
    atype[env] = 1 + (int)cg_index(r.v[0], 3u);
  }
  dev_cnt[env] = k;
}

}  // namespace

// =====================================================================
// C ABI
// =====================================================================
struct cygym_handle {
  int device_id;
  int n_envs;
  DevTopo t;
  cygym_config c;
  cygym_buffers b;
  cygym_buffers snap;
  bool bound, has_snap;
  void* dev_blob;       // one allocation holding the topology copies
  uint32_t* keybuf;     // randomize scratch [n_envs][Mp]
  int wpb, max_devs;
  int wave_lds, shared_lds;
  hipEvent_t ev0, ev1;
  KParams* dparams;     // device copy of the launch parameters (read by the fused kernel)
  unsigned long long* dbg;
  char err[256];
};

static char g_err[256] = "";

static int fail(cygym_handle* h, int code, const char* fmt, const char* detail) {
  char* dst = h ? h->err : g_err;
  snprintf(dst, 256, fmt, detail ? detail : "");
  if (h) snprintf(g_err, 256, "%s", dst);
  return code;
}
#define HIPCHK(h, call)                                                         \
  do {                                                                          \
    hipError_t _e = (call);                                                     \
    if (_e != hipSuccess) return fail(h, CYGYM_EHIP, #call ": %s", hipGetErrorString(_e)); \
  } while (0)

template <int MT, bool FUSED, bool XE>
static const void* kernel_for(int wpb) {
  switch (wpb) {
    case 16: return (const void*)step_kernel<16, MT, FUSED, XE>;
    case 8: return (const void*)step_kernel<8, MT, FUSED, XE>;
    case 4: return (const void*)step_kernel<4, MT, FUSED, XE>;
    case 2: return (const void*)step_kernel<2, MT, FUSED, XE>;
    default: return (const void*)step_kernel<1, MT, FUSED, XE>;
  }
}
template <bool FUSED, bool XE>
static const void* kernel_for_m(const cygym_handle* h) {
  if (h->t.M == 256) return kernel_for<256, FUSED, XE>(h->wpb);
  if (h->t.M == 64) return kernel_for<64, FUSED, XE>(h->wpb);
  return kernel_for<0, FUSED, XE>(h->wpb);
}
// XE: the kernel that follows the edges evolve_network adds (max_extra_edges > 0).  With no extra-edge list
// the lean instantiation runs: none of that code is in it.
static const void* pick_kernel(const cygym_handle* h, bool fused) {
  const bool xe = h->t.K > 0;
  if (fused) return xe ? kernel_for_m<true, true>(h) : kernel_for_m<true, false>(h);
  return xe ? kernel_for_m<false, true>(h) : kernel_for_m<false, false>(h);
}
static hipError_t set_lds_attr(cygym_handle* h) {
  const int lds = h->shared_lds + h->wave_lds * h->wpb;
  hipError_t e = hipFuncSetAttribute(pick_kernel(h, false), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(pick_kernel(h, true), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

extern "C" {

int cygym_version(void) { return CYGYM_ABI_VERSION; }
const char* cygym_last_error(const cygym_handle* h) { return h ? h->err : g_err; }

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// LDS budget: shared blob prefix + WPB per-wave regions.  Prefers staging the in-CSR too.
static size_t wave_lds_bytes(const DevTopo& t, int max_devs) {
  size_t w = align_up((size_t)4 * ((t.M + 3) & ~3), 16) + (size_t)t.Mp * 8 + (size_t)((t.EW + 3) & ~3) * 4 * 2 + CG_LOG_RING * 4 +
             (size_t)((t.Mp / 32 + 2) & ~1) * 4 + (size_t)t.MC * 8 + (size_t)t.Mp * 2 +
             align_up((size_t)max_devs * 2, 16) + (size_t)t.x_bytes + 128 /* scalar parking of the fused kernel */;
  return align_up(w, 16);
}
// The in-CSR (iptr/icol/ieid/oeid, ~2/3 of the blob) is read by block/unblock only (~9 % of env-ticks):
// it stays in global memory (L2-resident) and only the first o_iptr bytes are staged in LDS every tick.
static int choose_launch(cygym_handle* h, int max_devs) {
  DevTopo& t = h->t;
  const size_t lds_cap = 160 * 1024;
  const size_t wave = wave_lds_bytes(t, max_devs);
  const size_t shared = (size_t)t.o_iptr;
  int best = 0, best_waves = 0;
  const char* force = getenv("CYGYM_WPB");   // tuning aid: force the waves-per-workgroup choice
  const int forced = force ? atoi(force) : 0;
  for (int wpb = 16; wpb >= 1; wpb >>= 1) {
    if (forced && wpb != forced) continue;
    const size_t per_wg = shared + wave * wpb;
    if (per_wg > lds_cap) continue;
    int waves = (int)(lds_cap / per_wg) * wpb;
    if (waves > 32) waves = 32;
    // ties: two 8-wave workgroups per CU beat one 16-wave workgroup (their phases interleave)
    if (waves > best_waves || (waves == best_waves && wpb == 8)) { best_waves = waves; best = wpb; }
  }
  if (!best) return -1;
  h->wpb = best; h->wave_lds = (int)wave; h->shared_lds = (int)shared;
  t.lds_bytes = (int)shared; t.in_lds = 0;
  h->max_devs = max_devs;
  return 0;
}

int cygym_create(const cygym_topology* topo, const cygym_config* cfg, int32_t n_envs, int32_t device_id,
                 cygym_handle** out) {
  if (!topo || !cfg || !out || n_envs <= 0) return fail(nullptr, CYGYM_EINVAL, "cygym_create: bad argument%s", "");
  const int M = topo->n_devices, E = topo->n_edges, X = topo->n_exploits;
  if (M < 1 || M > 2048) return fail(nullptr, CYGYM_EUNSUPPORTED, "n_devices must be in [1, 2048]%s", "");
  if (E < 0 || E > 65535) return fail(nullptr, CYGYM_EUNSUPPORTED, "n_edges must be <= 65535%s", "");
  if (X < 0 || X > CG_MAX_EXPLOITS) return fail(nullptr, CYGYM_EINVAL, "n_exploits out of range%s", "");
  if (!cfg->fast_scan) return fail(nullptr, CYGYM_EUNSUPPORTED, "fast_scan=False (per-log scan path) is not implemented%s", "");
  if (cfg->num_of_device > 5000) return fail(nullptr, CYGYM_EUNSUPPORTED, "numOfDevice > 5000 (ready-set path) is not implemented%s", "");
  // host-side validation of the CSR: a malformed topology must never reach a kernel
  for (int i = 0; i <= M; ++i) {
    if (topo->out_ptr[i] < 0 || topo->out_ptr[i] > E || topo->in_ptr[i] < 0 || topo->in_ptr[i] > E ||
        (i && (topo->out_ptr[i] < topo->out_ptr[i - 1] || topo->in_ptr[i] < topo->in_ptr[i - 1])))
      return fail(nullptr, CYGYM_EINVAL, "malformed CSR row pointers%s", "");
  }
  if (topo->out_ptr[0] != 0 || topo->out_ptr[M] != E || topo->in_ptr[0] != 0 || topo->in_ptr[M] != E)
    return fail(nullptr, CYGYM_EINVAL, "CSR row pointers do not span the edge array%s", "");
  for (int k = 0; k < E; ++k) {
    if (topo->out_col[k] < 0 || topo->out_col[k] >= M || topo->in_col[k] < 0 || topo->in_col[k] >= M ||
        topo->in_eid[k] < 0 || topo->in_eid[k] >= E)
      return fail(nullptr, CYGYM_EINVAL, "CSR column / edge id out of range%s", "");
  }
  {   // in_eid must be a bijection in-entry -> out-slot that agrees with both CSRs
    unsigned char* seen = (unsigned char*)calloc((size_t)(E > 0 ? E : 1), 1);
    if (!seen) return fail(nullptr, CYGYM_EINVAL, "out of host memory%s", "");
    bool ok = true;
    for (int v = 0; v < M && ok; ++v)
      for (int j = topo->in_ptr[v]; j < topo->in_ptr[v + 1] && ok; ++j) {
        const int k = topo->in_eid[j], u = topo->in_col[j];
        ok = !seen[k] && topo->out_col[k] == v && k >= topo->out_ptr[u] && k < topo->out_ptr[u + 1];
        seen[k] = 1;
      }
    free(seen);
    if (!ok) return fail(nullptr, CYGYM_EINVAL, "in_eid does not match the out-CSR%s", "");
  }
  {   // edges evolve_network may add are kept per env and merged into the rows by neighbour id
    const int K = topo->max_extra_edges;
    if (K < 0 || K > 4096 || E + K > 65535) return fail(nullptr, CYGYM_EINVAL, "max_extra_edges must be in [0, 4096] and n_edges + max_extra_edges <= 65535%s", "");
    bool sorted = true;
    for (int u = 0; u < M && sorted && K > 0; ++u) {
      for (int k = topo->out_ptr[u] + 1; k < topo->out_ptr[u + 1] && sorted; ++k) sorted = topo->out_col[k - 1] <= topo->out_col[k];
      for (int k = topo->in_ptr[u] + 1; k < topo->in_ptr[u + 1] && sorted; ++k) sorted = topo->in_col[k - 1] <= topo->in_col[k];
    }
    if (!sorted) return fail(nullptr, CYGYM_EINVAL, "max_extra_edges > 0 needs adjacency rows sorted by neighbour id%s", "");
  }
  cygym_handle* h = new (std::nothrow) cygym_handle();
  if (!h) return fail(nullptr, CYGYM_EINVAL, "out of host memory%s", "");
  memset(h, 0, sizeof(*h));
  h->device_id = device_id; h->n_envs = n_envs; h->c = *cfg;
  hipError_t e0 = hipSetDevice(device_id);
  if (e0 != hipSuccess) { fail(nullptr, CYGYM_EHIP, "hipSetDevice: %s", hipGetErrorString(e0)); delete h; return CYGYM_EHIP; }
  DevTopo& t = h->t;
  t.M = M; t.X = X; t.E = E; t.EW = (E + 31) / 32 > 0 ? (E + 31) / 32 : 1;
  t.MC = (M + WAVE - 1) / WAVE; t.Mp = t.MC * WAVE;
  t.K = topo->max_extra_edges; t.KW = (t.K + 31) / 32;
  t.x_bytes = t.K > 0 ? (int)align_up((size_t)4 * (t.K + ((t.KW + 1) & ~1)) + (size_t)16 * t.MC, 16) : 0;
  // one blob, laid out exactly as the LDS-shared section (see DevTopo)
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 16); return (int)o; };
  t.o_optr = take((size_t)(M + 1) * 2); t.o_ocol = take((size_t)(E > 0 ? E : 1) * 2);
  t.o_os = take((size_t)M * 4); t.o_ver = take((size_t)M * 4); t.o_ano = take((size_t)M * 4);
  t.o_dst = take(M); t.o_vul = take(M); t.o_nap = take(M);
  t.o_iptr = take((size_t)(M + 1) * 2); t.o_icol = take((size_t)(E > 0 ? E : 1) * 2); t.o_ieid = take((size_t)(E > 0 ? E : 1) * 2);
  t.o_oeid = take((size_t)(E > 0 ? E : 1) * 2);
  t.blob_bytes = (int)off;
  t.multi = 0;   // duplicate (u,v) out-entries? (env._blocked holds pairs, so duplicates share their state)
  for (int u = 0; u < M && !t.multi; ++u)
    for (int k = topo->out_ptr[u]; k < topo->out_ptr[u + 1] && !t.multi; ++k)
      for (int k2 = k + 1; k2 < topo->out_ptr[u + 1]; ++k2)
        if (topo->out_col[k2] == topo->out_col[k]) { t.multi = 1; break; }
  if (choose_launch(h, M > 8 ? M / 8 : 1) != 0) { delete h; return fail(nullptr, CYGYM_EUNSUPPORTED, "topology does not fit in LDS%s", ""); }
  uint8_t* host = (uint8_t*)calloc(1, off);
  if (!host) { delete h; return fail(nullptr, CYGYM_EINVAL, "out of host memory%s", ""); }
  memcpy(host + t.o_dst, topo->dstatic, M); memcpy(host + t.o_vul, topo->vuln, M); memcpy(host + t.o_nap, topo->napps, M);
  for (int u = 0; u < M; ++u) {   // library-private static bit: out-row == every other device, ascending
    host[t.o_dst + u] &= (uint8_t)(CG_D_DC | CG_D_SERVER);
    if (topo->out_ptr[u + 1] - topo->out_ptr[u] != M - 1) continue;
    bool full = true;
    for (int j = 0; j < M - 1 && full; ++j) full = topo->out_col[topo->out_ptr[u] + j] == j + (j >= u ? 1 : 0);
    if (full) host[t.o_dst + u] |= 0x04;
  }
  memcpy(host + t.o_os, topo->os_val, (size_t)M * 4); memcpy(host + t.o_ver, topo->version, (size_t)M * 4);
  memcpy(host + t.o_ano, topo->anomaly, (size_t)M * 4);
  for (int i = 0; i <= M; ++i) { ((uint16_t*)(host + t.o_optr))[i] = (uint16_t)topo->out_ptr[i]; ((uint16_t*)(host + t.o_iptr))[i] = (uint16_t)topo->in_ptr[i]; }
  for (int k = 0; k < E; ++k) {
    ((uint16_t*)(host + t.o_ocol))[k] = (uint16_t)topo->out_col[k];
    ((uint16_t*)(host + t.o_icol))[k] = (uint16_t)topo->in_col[k];
    ((uint16_t*)(host + t.o_ieid))[k] = (uint16_t)topo->in_eid[k];
    ((uint16_t*)(host + t.o_oeid))[topo->in_eid[k]] = (uint16_t)k;   // inverse map: in-entry of every out-slot
  }
  hipError_t e1 = hipMalloc(&h->dev_blob, off);
  if (e1 == hipSuccess) e1 = hipMemcpy(h->dev_blob, host, off, hipMemcpyHostToDevice);
  free(host);
  if (e1 == hipSuccess) e1 = hipMalloc((void**)&h->keybuf, (size_t)n_envs * t.Mp * 4);
  if (e1 == hipSuccess) e1 = hipMalloc((void**)&h->dparams, sizeof(KParams));
  if (e1 == hipSuccess) e1 = hipEventCreate(&h->ev0);
  if (e1 == hipSuccess) e1 = hipEventCreate(&h->ev1);
  if (e1 != hipSuccess) { fail(nullptr, CYGYM_EHIP, "cygym_create: %s", hipGetErrorString(e1)); cygym_destroy(h); return CYGYM_EHIP; }
  uint8_t* d = (uint8_t*)h->dev_blob;
  t.blob = d;
  t.dstatic = d + t.o_dst; t.vuln = d + t.o_vul; t.napps = d + t.o_nap;
  t.os_val = (const float*)(d + t.o_os); t.version = (const float*)(d + t.o_ver); t.anomaly = (const float*)(d + t.o_ano);
  t.out_ptr = (const uint16_t*)(d + t.o_optr); t.out_col = (const uint16_t*)(d + t.o_ocol);
  t.in_ptr = (const uint16_t*)(d + t.o_iptr); t.in_col = (const uint16_t*)(d + t.o_icol); t.in_eid = (const uint16_t*)(d + t.o_ieid);
  // opt in to large dynamic LDS for every instantiation we may launch
  hipError_t e2 = set_lds_attr(h);
  if (e2 != hipSuccess) { fail(nullptr, CYGYM_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e2)); cygym_destroy(h); return CYGYM_EHIP; }
  *out = h;
  return CYGYM_OK;
}

void cygym_destroy(cygym_handle* h) {
  if (!h) return;
  if (h->dev_blob) (void)hipFree(h->dev_blob);
  if (h->keybuf) (void)hipFree(h->keybuf);
  if (h->dparams) (void)hipFree(h->dparams);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
}

int cygym_set_config(cygym_handle* h, const cygym_config* cfg) {
  if (!h || !cfg) return fail(h, CYGYM_EINVAL, "cygym_set_config: bad argument%s", "");
  if (!cfg->fast_scan) return fail(h, CYGYM_EUNSUPPORTED, "fast_scan=False is not implemented%s", "");
  h->c = *cfg;
  return CYGYM_OK;
}

static int check_buffers(cygym_handle* h, const cygym_buffers* b, bool snapshot) {
  if (!b || !b->live || !b->stash || !b->blocked || !b->blocked_in || !b->ring || !b->ienv || !b->fenv)
    return fail(h, CYGYM_EINVAL, "buffer struct has a null plane%s", "");
  if (h->t.K > 0 && !b->extra) return fail(h, CYGYM_EINVAL, "max_extra_edges > 0 needs the `extra` plane%s", "");
  if (snapshot ? (b->n_envs != 1 && b->n_envs != h->n_envs) : (b->n_envs != h->n_envs))
    return fail(h, CYGYM_EINVAL, "buffer struct has the wrong leading dimension%s", "");
  return CYGYM_OK;
}

int cygym_bind(cygym_handle* h, const cygym_buffers* state) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_bind: null handle%s", "");
  int rc = check_buffers(h, state, false);
  if (rc) return rc;
  h->b = *state;
  h->bound = true;
  return CYGYM_OK;
}

static KParams make_params(cygym_handle* h) {
  KParams P;
  memset(&P, 0, sizeof(P));
  P.t = h->t; P.c = h->c; P.b = h->b; P.n_envs = h->n_envs;
  P.wave_lds = h->wave_lds; P.shared_lds = h->shared_lds;
  P.dbg = h->dbg;
  return P;
}

int cygym_derive(cygym_handle* h, const cygym_buffers* bufs, void* stream) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_derive: null handle%s", "");
  int rc = check_buffers(h, bufs, true);
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(derive_kernel, dim3((bufs->n_envs + waves_per_block - 1) / waves_per_block), dim3(threads), 0,
                     (hipStream_t)stream, P, *bufs);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_reset(cygym_handle* h, const cygym_buffers* snapshot, const int32_t* env_ids, int32_t n, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_reset: handle not bound%s", "");
  if (!snapshot) {
    if (!h->has_snap) return fail(h, CYGYM_EINVAL, "cygym_reset: no snapshot given or registered%s", "");
    snapshot = &h->snap;
  }
  int rc = check_buffers(h, snapshot, true);
  if (rc) return rc;
  if (!env_ids) n = h->n_envs;
  if (n <= 0) return CYGYM_OK;
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  P.snap = *snapshot;
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(reset_kernel, dim3((n + waves_per_block - 1) / waves_per_block), dim3(threads), 0,
                     (hipStream_t)stream, P, env_ids, n);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_randomize(cygym_handle* h, const int32_t* env_ids, int32_t n, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_randomize: handle not bound%s", "");
  if (!env_ids) n = h->n_envs;
  if (n <= 0) return CYGYM_OK;
  if (n > h->n_envs) return fail(h, CYGYM_EINVAL, "cygym_randomize: more ids than envs%s", "");
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(randomize_kernel, dim3((n + waves_per_block - 1) / waves_per_block), dim3(threads), 0,
                     (hipStream_t)stream, P, env_ids, n, h->keybuf);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_set_snapshot(cygym_handle* h, const cygym_buffers* snapshot) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_set_snapshot: null handle%s", "");
  if (!snapshot) { h->has_snap = false; memset(&h->snap, 0, sizeof(h->snap)); return CYGYM_OK; }
  int rc = check_buffers(h, snapshot, true);
  if (rc) return rc;
  h->snap = *snapshot;
  h->has_snap = true;
  return CYGYM_OK;
}

int cygym_step(cygym_handle* h, const cygym_actions* a, const cygym_outputs* o, void* stream) {
  return cygym_rollout(h, 1, a, o, stream);
}

int cygym_rollout(cygym_handle* h, int32_t n_ticks, const cygym_actions* a, const cygym_outputs* o, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_step: handle not bound%s", "");
  if (n_ticks < 1) return fail(h, CYGYM_EINVAL, "cygym_rollout: n_ticks must be >= 1%s", "");
  if (!a || !o || !a->mode || !a->n_groups || !a->atype || !a->n_exploit || !a->exploit || !a->app ||
      !a->dev_cnt || !a->dev_idx || !o->obs || !o->raw || !o->shaped || !o->done)
    return fail(h, CYGYM_EINVAL, "cygym_step: null action / output pointer%s", "");
  if (a->max_groups < 1 || a->max_devs < 1) return fail(h, CYGYM_EINVAL, "cygym_step: max_groups / max_devs must be >= 1%s", "");
  if (a->max_devs > 32767) return fail(h, CYGYM_EINVAL, "cygym_step: max_devs too large%s", "");
  if (a->max_devs > h->max_devs) {   // the device list lives in LDS: re-plan the launch for a longer list
    if (choose_launch(h, a->max_devs) != 0) return fail(h, CYGYM_EUNSUPPORTED, "device list does not fit in LDS%s", "");
    HIPCHK(h, set_lds_attr(h));
  }
  if (h->c.auto_reset && !h->has_snap) return fail(h, CYGYM_EINVAL, "auto_reset needs cygym_set_snapshot first%s", "");
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  P.a = *a; P.o = *o;
  P.n_ticks = n_ticks;
  P.snap = h->snap;
  P.self = h->dparams;
  if (n_ticks > 1) HIPCHK(h, hipMemcpyAsync(h->dparams, &P, sizeof(KParams), hipMemcpyHostToDevice, (hipStream_t)stream));
  const int lds = h->shared_lds + h->wave_lds * h->wpb;
  const dim3 grid((h->n_envs + h->wpb - 1) / h->wpb), block(h->wpb * WAVE);
  hipStream_t s = (hipStream_t)stream;
  {
    void* args[] = {(void*)&P};
    HIPCHK(h, hipLaunchKernel(pick_kernel(h, n_ticks > 1), grid, block, args, (size_t)lds, s));
  }
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_observe(cygym_handle* h, int32_t role, float* out, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_observe: handle not bound%s", "");
  if (!out || role < 0 || role > 2) return fail(h, CYGYM_EINVAL, "cygym_observe: bad argument%s", "");
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(observe_kernel, dim3((h->n_envs + waves_per_block - 1) / waves_per_block), dim3(threads), 0,
                     (hipStream_t)stream, P, role, out);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_gen_actions(cygym_handle* h, int32_t tick, int32_t* mode, int32_t* n_groups, int32_t* atype,
                      int32_t* n_exploit, int32_t* exploit, int32_t* app, int32_t* dev_cnt, int16_t* dev_idx,
                      int32_t max_devs, void* stream) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_gen_actions: null handle%s", "");
  if (!mode || !n_groups || !atype || !n_exploit || !exploit || !app || !dev_cnt || !dev_idx || max_devs < 1)
    return fail(h, CYGYM_EINVAL, "cygym_gen_actions: bad argument%s", "");
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  hipLaunchKernelGGL(gen_actions_kernel, dim3((h->n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, tick,
                     mode, n_groups, atype, n_exploit, exploit, app, dev_cnt, dev_idx, max_devs);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

/* diagnostic builds only (-DCG_STAMPS): per-env phase stamps, [N][16] uint64 device buffer (NULL to disable) */
int cygym_set_debug(cygym_handle* h, void* buf) {
  if (!h) return fail(h, CYGYM_EINVAL, "null handle%s", "");
  h->dbg = (unsigned long long*)buf;
  return CYGYM_OK;
}

int cygym_timer_start(cygym_handle* h, void* stream) {
  if (!h) return fail(h, CYGYM_EINVAL, "null handle%s", "");
  HIPCHK(h, hipEventRecord(h->ev0, (hipStream_t)stream));
  return CYGYM_OK;
}
int cygym_timer_stop(cygym_handle* h, void* stream, float* ms) {
  if (!h || !ms) return fail(h, CYGYM_EINVAL, "null argument%s", "");
  HIPCHK(h, hipEventRecord(h->ev1, (hipStream_t)stream));
  HIPCHK(h, hipEventSynchronize(h->ev1));
  HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  return CYGYM_OK;
}

}  // extern "C"
