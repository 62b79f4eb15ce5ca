// cg_evolve.hpp -- evolve_network (CyberDefenseEnv.py:583-875).
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_EVOLVE_HPP
#define CG_EVOLVE_HPP

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

template <bool XE, class KP>
__device__ __forceinline__ void evolve(Env& e, const KP& P) {
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
    n_ev = cdf_lookup(e.draw(CG_SITE_EVO_POISSON, 0, 0), P.c.poisson_thr, CG_POISSON_TABLE);
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
      // The common answer is "every link is there": two bitmasks of the hub's neighbours (its out-row, its in-row and
      // the added edges) answer all owners at once.  Only owners with a missing link take the edge-by-edge path below,
      // in the reference's order.  (One edge_exists per link was 100 k cycles per check at 2048 devices with ~100
      // owners -- the slowest env of every launch that held one.)
      const int MW = (MC * WAVE) / 32;
      uint32_t* om = e.scr;        // [MW] devices the hub points to
      uint32_t* im = e.scr + MW;   // [MW] devices pointing to the hub
#pragma nounroll
      for (int i = e.lane; i < 2 * MW; i += WAVE) e.scr[i] = 0u;
      wsync();
      const int ho0 = e.optr[hub], ho1 = e.optr[hub + 1], hi0 = e.iptr(hub), hi1 = e.iptr(hub + 1);
#pragma nounroll
      for (int k = ho0 + e.lane; k < ho1; k += WAVE) { const int v = e.ocol[k]; atomicOr(&om[v >> 5], 1u << (v & 31)); }
#pragma nounroll
      for (int k = hi0 + e.lane; k < hi1; k += WAVE) { const int u = e.icol_g[k]; atomicOr(&im[u >> 5], 1u << (u & 31)); }
      if constexpr (XE) {
        const int n = x_cnt(e);
#pragma nounroll
        for (int j = e.lane; j < n; j += WAVE) {
          const uint32_t key = e.xk[j];
          const int u = (int)(key >> 16), v = (int)(key & 0xFFFFu);
          if (u == hub) atomicOr(&om[v >> 5], 1u << (v & 31));
          if (v == hub) atomicOr(&im[u >> 5], 1u << (u & 31));
        }
      }
      wsync();
#pragma nounroll
      for (int c = 0; c < MC; ++c) {
        int d = c * WAVE + e.lane;
        const bool owner = d < M && d != hub && (e.flags[d] & (CG_F_OWNED | CG_F_EVOACT)) == (CG_F_OWNED | CG_F_EVOACT);
        const bool linked = ((om[d >> 5] >> (d & 31)) & (im[d >> 5] >> (d & 31)) & 1u) != 0u;
        uint64_t m = ballot(owner && !linked);
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
        total += __builtin_amdgcn_readlane(incl, 63);
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

#endif  // CG_EVOLVE_HPP
