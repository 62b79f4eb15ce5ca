// cg_extra_edges.hpp -- Edges added by evolve_network: the per-env extra-edge list and the merged-row walk.
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_EXTRA_EDGES_HPP
#define CG_EXTRA_EDGES_HPP

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

#endif  // CG_EXTRA_EDGES_HPP
