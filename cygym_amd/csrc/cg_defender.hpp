// cg_defender.hpp -- Defender actions (volt_typhoon_env.py:918-1123).
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_DEFENDER_HPP
#define CG_DEFENDER_HPP

// ---------------- defender ----------------
template <class KP, class IE, class FE>   // IE / FE: the env's scalars -- arrays, or lane views of the scalars register (EnvI / EnvF, cg_wave.hpp)
__device__ __forceinline__ void def_global(Env& e, const KP& P, int at, const int16_t* dev, int L, double& cost,
                                           bool& dirty, bool grouped, IE ie, FE fe) {
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
    if (e.log_total > 0) {   // Detector.train(non-empty logs) CDSimulator.py:692-695: the fit is the host's job
      e.eflags |= CG_E_DET_TRAIN | CG_E_DET_PENDING; e.eflags &= ~CG_E_DET_RANDOM;
      if (COLD(P.b.forest != nullptr) && e.lane == 0) {   // the request: which tick asked, and on how many logs
        uint32_t* fo = P.b.forest + (size_t)e.env * CG_FOREST_WORDS;
        fo[6] = (fo[3] == e.tick && fo[6] > 0) ? fo[6] + 1 : 1;   // several groups of one step_grouped tick may ask
        fo[3] = e.tick; fo[4] = (uint32_t)e.log_total;
      }
    }
  } else if (at == 11) {  // :964-976, _device_state :419-428
    if (L > 0) {
      int d = dev[0];
      if (d >= 0 && d < M && e.lane == 0) {
        e.stash[d] = (uint8_t)(CG_S_VALID | (e.flags[d] & CG_S_KEEP));
        e.stash[M + d] = e.busy[d];
        e.stash[2 * M + d] = e.wl[d];
        e.stash[3 * M + d] = (uint8_t)cby_get(e, d);
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
template <class KP, class IE, class FE>
__device__ __forceinline__ void def_clean(Env& e, const KP& P, const int16_t* dev, int L, double& cost,
                                          IE ie, FE fe, uint8_t* occ) {
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
      dl = (int)cby_get(e, d);
      cby_clear(e, d);
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
        disc |= (int)cby_get(e, d);
        cby_clear(e, d);
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
// NW: words a row's bits can span, known at compile time (9 at 256 devices, 3 at 64: cygym_create sends a topology with a longer
// row -- duplicate edges -- to the run-time-size kernels), or 0: run-time sizes, the rolled per-lane loops (both forms in one kernel
// cost it a wave per SIMD: 114 VGPRs).
template <int NW>
__device__ __forceinline__ Pick pool_pick(const PoolPtrs q, bool want, uint32_t u, int o0, int o1, int i0, int i1) {
  Pick p; p.slot = -1; p.j = -1; p.x = -1;
  int nbo, nbi;
  if constexpr (NW > 0) { nbo = range_popc_words<NW>(q.blk, o0, o1); nbi = range_popc_words<NW>(q.bin, i0, i1); }
  else { nbo = range_popc(q.blk, o0, o1); nbi = range_popc(q.bin, i0, i1); }
  const int n_out = want ? nbo : (o1 - o0) - nbo;
  const int n_in = want ? nbi : (i1 - i0) - nbi;
  const int n = n_out + n_in;
  if (n == 0) return p;
  const int r = (int)cg_index(u, (uint32_t)n);
  const bool from_out = r < n_out;
  int slot = -1, j = -1;
  if constexpr (NW > 0) {
    if (from_out) slot = range_select_words<NW>(q.blk, o0, o1, want, r);
    else          j = range_select_words<NW>(q.bin, i0, i1, want, r - n_out);
  } else {
    if (from_out) slot = range_select(q.blk, o0, o1, want, r);
    else          j = range_select(q.bin, i0, i1, want, r - n_out);
  }
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

// Block / unblock of ONE list entry whose device is an endpoint of an ADDED edge: its pools are the merged rows
// (:502-511).  Wave-cooperative, uniform arguments; runs at the entry's turn in list order (every earlier entry has
// been applied), so it reads the bitmasks as the reference would.
// Element r of a merged pool: walk the added-edge candidates in row order; candidate j has rank
// (#added candidates before it) + (#base candidates with a smaller neighbour id).
// occ: occurrence numbers per device (nullptr: the list holds no device twice).
__device__ __forceinline__ void block_one(Env& e, const PoolPtrs q, int d, bool want, uint32_t site, uint8_t* occ, int& n_hit) {
  const int n = x_cnt(e);
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
  if (total == 0) return;
  const int occ_d = occ ? (int)occ[d] : 0;
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
  if (occ && e.lane == 0) occ[d] = (uint8_t)(occ_d + 1);
  ++n_hit;
  wsync();
}

// Action 5 with fast_scan == False (volt_typhoon_env.py:1030-1050): every scan predicts the last <= 256 log entries one
// by one (Detector.predict), pays 0.5 * def_scale per entry, and an entry predicted "A" discovers the exploits its
// SENDER was compromised by, cleans and stalls the sender; the scanned device's anomaly_score becomes the score of the
// last entry (None -> -1 unless the detector is trained; 0.0 in turbo mode).  Wave-cooperative, one lane per entry,
// 64 entries per step, read from the long history (global memory).  Uniform arguments.
//   n_mult: scans (list entries on active devices); the entries are the same for each of them, so are the trained
//   detector's predictions -- only the coin mode draws per scan -- and a flagged sender keeps the stall of the last
//   scan that flagged it (the draw is addressed by sender and scan ordinal, like on the fast path).
template <class KP, class IE, class FE>
__device__ __forceinline__ float slow_scan(Env& e, const KP& P, int n_mult, double& cost, IE ie, FE fe) {
  const int M = e.M;
  const double ds = P.c.def_scale;
  const int w = e.log_total < CG_SLOW_SCAN_WINDOW ? e.log_total : CG_SLOW_SCAN_WINDOW;
  if (w <= 0 || n_mult <= 0) return __builtin_nanf("");   // no entry looked at: the scores stand
  const double c = 0.5 * (double)w * (double)n_mult * ds;
  cost += -c;
  fe[CG_D_DEF_COST] += c;
  float score = -1.f;   // Detector.predict(..., return_score=True) gives None unless trained
  if (P.c.turbo) return 0.f;   // :1036-1038 predicted_kind = None, anomaly_score = 0.0
  const uint16_t* hist = P.b.hist + (size_t)e.env * CG_HIST_RING * 2;
  const bool coin = (e.eflags & CG_E_DET_RANDOM) != 0;
  bool trained = !coin && (e.eflags & CG_E_DET_TRAIN);
  const uint32_t* fo = P.b.forest ? P.b.forest + (size_t)e.env * CG_FOREST_WORDS : nullptr;
  const double* apl = P.t.apl;
  if (trained && (!fo || !apl || (e.eflags & CG_E_DET_PENDING) || fo[2] == 0u)) {
    e.eflags |= CG_E_UNPINNED;   // no current forest: all "D", flagged (cygym_spec.h)
    trained = false;
  }
  if (!coin && !trained) return score;   // untrained: every entry "D", score None
  int disc = 0;
#pragma nounroll
  for (int s = 0; s < (coin ? n_mult : 1); ++s) {
    const int ord = coin ? s : n_mult - 1;
#pragma nounroll
    for (int p0 = 0; p0 < w; p0 += WAVE) {
      const int p = p0 + e.lane;
      bool anom = false;
      int snd = 0;
      if (p < w) {
        const uint32_t idx = (uint32_t)(e.log_total - w + p);
        snd = hist[2 * (idx % CG_HIST_RING)];
        const uint32_t to = hist[2 * (idx % CG_HIST_RING) + 1];
        if (coin) {
          anom = cg_index(e.draw(CG_SITE_DET_COIN, p, s), 2) == 0;
        } else {
          double depths = 0.0;
#pragma nounroll
          for (int t = 0; t < CG_FOREST_TREES; ++t) {
            const uint32_t* tr = fo + CG_FOREST_HDR + t * CG_FOREST_NODES;
            uint32_t nd = tr[0];
#pragma nounroll
            for (int it = 0; it < 16 && !CG_FN_LEAF(nd); ++it) {
              const uint32_t x = CG_FN_FEAT(nd) ? to : (uint32_t)snd;
              nd = tr[x <= CG_FN_THR(nd) ? CG_FN_LEFT(nd) : CG_FN_RIGHT(nd)];
            }
            uint32_t ns = CG_FN_NSAMP(nd);
            if (ns >= CG_DET_APL_N) ns = CG_DET_APL_N - 1;
            depths += ((double)CG_FN_DEPTH(nd) + apl[ns]) - 1.0;
          }
          anom = depths < __hiloint2double((int)fo[1], (int)fo[0]);
          if (p == w - 1) {   // the scanned device keeps the decision_function value of the last entry
            uint32_t ms = fo[7];
            if (ms >= CG_DET_APL_N) ms = CG_DET_APL_N - 1;
            const double den = (double)CG_FOREST_TREES * apl[ms];
            score = (float)(0.5 - exp2(-(den != 0.0 ? depths / den : 1.0)));
          }
        }
      }
      if (anom && snd < M) {   // (a sender id outside the network: the reference would raise KeyError)
        disc |= (int)cby_get(e, snd);
        cby_clear(e, snd);
        atomicAnd((unsigned int*)(e.flags + (snd & ~3)), ~((uint32_t)CG_F_COMP << ((snd & 3) * 8)));
        e.busy[snd] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_SCAN, snd, ord), 0, P.c.default_high);
      }
      wsync();
    }
  }
  ie[CG_I_DISCOVERED] |= wave_or(disc);
  if (!coin) score = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(score), (w - 1) & 63));
  return score;
}

// SLOW: the instantiation carries the per-log scan path (full-feature per-tick kernels; the rollout kernels sit at their
// 128-VGPR cap and do not -- cygym_rollout issues a fast_scan=False handle's ticks as single-tick launches)
template <bool XE, bool WIDE, bool SLOW, int NW, class KP, class IE, class FE>
__device__ __forceinline__ void def_per_device(Env& e, const KP& P, int at, const int16_t* dev, int L, int app,
                                               double& cost, bool& dirty, IE ie, FE fe) {
  const double ds = P.c.def_scale;
  const int M = e.M;
  if (at == 1) { def_clean(e, P, dev, L, cost, ie, fe, nullptr); return; }
  if (at == 6 || at == 9) {  // sequential semantics: each pick changes the pools of BOTH endpoints
    __builtin_amdgcn_s_setprio(3);   // long path: see the spread
    // (every per-lane address of this path is derived from the lane id HERE: hoisted to the prologue -- "list + 2 * lane",
    // "scratch + 4 * lane" -- and kept live to this point they were spilled registers of the 80-VGPR kernels)
    asm volatile("" : "+v"(e.lane));
    const uint32_t site = at == 6 ? CG_SITE_PICK_BLOCK : CG_SITE_PICK_UNBLOCK;
    const bool want = (at == 9);
    const bool simple = list_is_simple(e, dev, L);   // no device twice => occurrence number is always 0
    uint8_t* occ = (uint8_t*)e.lsrc;   // occurrence numbers (the spread's source list is idle here; the scratch area holds the first-touch table)
    if (!simple) {
      for (int i = e.lane; i < (e.MC * WAVE) / 4; i += WAVE) ((uint32_t*)occ)[i] = 0;
      wsync();
    }
    uint32_t* fh = e.scr;   // [Mp] first remaining entry (lane) whose flipped edge ends at this device
    PoolPtrs q;
    q.blk = e.blk; q.bin = e.bin; q.optr = e.optr; q.ocol = e.ocol; q.icol = e.icol_g; q.ieid = e.ieid_g; q.oeid = e.oeid_g;
    const bool multi = e.multi;
    int n_act = 0, n_hit = 0;
    [[maybe_unused]] int n_pass_all = 0;   // diagnostic builds
    const bool xany = XE && x_cnt(e) > 0;   // this env carries added edges: entries on their endpoints take block_one
    for (int p0 = 0; p0 < L; p0 += WAVE) {
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
      uint64_t xm = 0ull;   // remaining entries whose device is an endpoint of an added edge
      if constexpr (XE) { if (COLD(xany)) xm = ballot(d >= 0 && x_isinc(e, d)); }
      if (p0 == 0) SUBSTAMP(10);
      [[maybe_unused]] int n_pass = 0;   // read by the stamps of diagnostic builds
      // Speculate: every remaining entry picks on the bitmasks as they stand.  An entry is exact unless an
      // EARLIER remaining entry flips an edge ending at its device (or is the same device); apply the exact
      // prefix in parallel and repeat from the first inexact entry (at least one entry retires per pass).
      // The first-touch table is stamped with the pass number (high bits), so it is cleared once per chunk of the
      // list, not once per pass: an entry of an older pass compares as "nobody touched this device yet".
      for (int i = e.lane; i < e.MC * WAVE; i += WAVE) fh[i] = 0u;
      wsync();
      uint32_t epoch = 0;
      while (am) {
        int lim = WAVE;   // speculate only ahead of the next merged-pool entry
        if constexpr (XE) {
          if (COLD(xm != 0ull)) {
            const int first = __builtin_ctzll(am);
            if ((xm >> first) & 1ull) {   // next in list order: alone, on the bitmasks every earlier entry has left
              block_one(e, q, __builtin_amdgcn_readlane(d, first), want, site, simple ? nullptr : occ, n_hit);
              am &= am - 1; xm &= xm - 1;
              // (the per-lane row bounds are re-read rather than kept in registers across the cooperative pick)
              asm volatile("" : "+v"(d));
              if (d >= 0) { o0 = e.optr[d]; o1 = e.optr[d + 1]; i0 = e.iptr(d); i1 = e.iptr(d + 1); }
              continue;
            }
            lim = __builtin_ctzll(xm);
          }
        }
        epoch += 0x100u;   // (pass + 1) << 8, above the 6 lane bits
        const bool mine = ((am >> e.lane) & 1ull) && e.lane < lim;
        Pick pk; pk.slot = -1; pk.j = -1; pk.x = -1;
        if (mine) {
          uint32_t uu = u;
          if (!simple) { const int b = occ[d]; if (b > 0) uu = e.draw(site, d, b); }
          pk = pool_pick<NW>(q, want, uu, o0, o1, i0, i1);
          // key = epoch | (63 - lane): atomicMax keeps the newest pass and, within it, the smallest lane
          if (pk.slot >= 0) atomicMax(&fh[pk.x], epoch | (uint32_t)(63 - e.lane));
          if (!simple) atomicMax(&fh[d], epoch | (uint32_t)(63 - e.lane));   // a repeated device must wait for its first occurrence
        }
        wsync();
        const uint32_t ft = mine ? fh[d] : 0u;
        const bool taint = mine && (ft & ~0xFFu) == epoch && (63u - (ft & 0x3Fu)) < (uint32_t)e.lane;
        const uint64_t tm = ballot(taint);
        int q0 = tm ? __builtin_ctzll(tm) : WAVE;
        if (q0 > lim) q0 = lim;
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
      n_pass_all += n_pass;
      SUBSTAMP(11);
      SUBVAL(15, n_pass_all);
      SUBVAL(14, n_act);
      SUBVAL(13, (simple ? 0 : 1) | (xany ? 2 : 0));
      SUBVAL(16, L);
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
          e.busy[d0] = sb; e.wl[d0] = sw; cby_put(e, d0, sc);
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
      cby_clear(e, d);
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
            cby_clear(e, d);
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
  } else if (COLD(at == 5 && !P.c.fast_scan)) {  // per-log scan :1030-1050 (full-feature kernels: history + anomaly plane bound)
    ie[CG_I_SCAN_CNT] += n_mult;
    if constexpr (SLOW) {
      const float sc = slow_scan(e, P, n_mult, cost, ie, fe);
      if (sc == sc && P.b.anomaly) {   // (NaN: no entry was looked at)  every scanned active device takes the last entry's score
        float* an = P.b.anomaly + (size_t)e.env * M;
        if (simple && L <= WAVE) {
          int d = -1;
          if (e.lane < L) { d = dev[e.lane]; if (d < 0 || d >= M) d = -1; }
          if (d >= 0 && !(e.flags[d] & CG_F_NYA)) an[d] = sc;
        } else {
          const uint8_t* cnt = (const uint8_t*)e.scr;   // multiplicities of list_counts above
          for (int d = e.lane; d < M; d += WAVE)
            if (cnt[d] > 0 && !(e.flags[d] & CG_F_NYA)) an[d] = sc;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // the observation writers of this wave read the plane back
      }
    }
    wsync();
  } else if (at == 5) {  // fast scan :1020-1069
    ie[CG_I_SCAN_CNT] += n_mult;
    int w = e.log_total < CG_SCAN_WINDOW ? e.log_total : CG_SCAN_WINDOW;
    if (w > 0 && n_mult > 0) {
      cost += -0.5 * n_mult * ds;
      fe[CG_D_DEF_COST] += 0.5 * n_mult * ds;
      if (COLD(P.c.turbo)) {   // turbo: predictions = [] (volt_typhoon_env.py:1055) -- the scan costs, nothing is flagged
      } else if (e.eflags & CG_E_DET_RANDOM) {  // Detector.batch_predict coin mode CDSimulator.py:715-716
        const int majority = w / 2 + 1;
        for (int s = 0; s < n_mult; ++s) {
          bool anom = false;
          if (e.lane < w) anom = cg_index(e.draw(CG_SITE_DET_COIN, e.lane, s), 2) == 0;
          uint64_t m = ballot(anom);
          if (__popcll(m) >= majority && anom) {
            uint32_t idx = (uint32_t)(e.log_total - w + e.lane);
            int snd = e.ring[2 * (idx % CG_LOG_RING)];
            if (snd < M) {
              atomicAnd((unsigned int*)(e.flags + (snd & ~3)), ~((uint32_t)CG_F_COMP << ((snd & 3) * 8)));
              e.busy[snd] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_SCAN, snd, s), 0, P.c.default_high);
            }
          }
          wsync();
        }
      }
      else if (COLD(e.eflags & CG_E_DET_TRAIN)) {   // trained detector: IsolationForest.predict == -1 (CDSimulator.py:721-723)
        const uint32_t* fo = nullptr;
        const double* apl = nullptr;
        if constexpr (XE) { if (P.b.forest) fo = P.b.forest + (size_t)e.env * CG_FOREST_WORDS; apl = P.t.apl; }
        if (!fo || !apl || (e.eflags & CG_E_DET_PENDING) || fo[2] == 0u) {   // (node-count word 0: no trees were ever installed)
          e.eflags |= CG_E_UNPINNED;   // no current forest: all "D", flagged (cygym_spec.h)
        } else {
          // The window, the forest and hence the predictions are the same for every scan of this tick; a flagged
          // sender ends up with the stall drawn by the LAST scan (ordinal n_mult - 1).  One lane per window entry
          // walks both flat trees (<= 9 levels each, words from L2).
          bool anom = false;
          int snd = 0;
          if (e.lane < w) {
            const uint32_t idx = (uint32_t)(e.log_total - w + e.lane);
            snd = e.ring[2 * (idx % CG_LOG_RING)];
            const uint32_t to = e.ring[2 * (idx % CG_LOG_RING) + 1];
            double depths = 0.0;
#pragma nounroll
            for (int t = 0; t < CG_FOREST_TREES; ++t) {
              const uint32_t* tr = fo + CG_FOREST_HDR + t * CG_FOREST_NODES;
              uint32_t nd = tr[0];
#pragma nounroll
              for (int it = 0; it < 16 && !CG_FN_LEAF(nd); ++it) {
                const uint32_t x = CG_FN_FEAT(nd) ? to : (uint32_t)snd;
                nd = tr[x <= CG_FN_THR(nd) ? CG_FN_LEFT(nd) : CG_FN_RIGHT(nd)];
              }
              uint32_t ns = CG_FN_NSAMP(nd);
              if (ns >= CG_DET_APL_N) ns = CG_DET_APL_N - 1;
              depths += ((double)CG_FN_DEPTH(nd) + apl[ns]) - 1.0;
            }
            const double sstar = __hiloint2double((int)fo[1], (int)fo[0]);
            anom = depths < sstar;   // foreign sender ids (>= M, a ring loaded from the host) count towards the majority
          }                          // like in the coin path and the oracle; only the write skips them
          const uint64_t m = ballot(anom);
          if (__popcll(m) >= w / 2 + 1 && anom && snd < M) {
            atomicAnd((unsigned int*)(e.flags + (snd & ~3)), ~((uint32_t)CG_F_COMP << ((snd & 3) * 8)));
            e.busy[snd] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_SCAN, snd, n_mult - 1), 0, P.c.default_high);
          }
          wsync();
        }
      }
      // untrained detector: all "D" (CDSimulator.py:718-719)
    }
  } else if (at == 13) {  // :1111-1123 -- acts on device_indices[0] once per listed active device
    int d0 = dev[0];
    if (n_mult > 0 && d0 >= 0 && d0 < M && e.lane == 0) {
      e.flags[d0] &= (uint8_t)~(CG_F_COMP | CG_F_WLADV);
      cby_clear(e, d0);
      e.wl[d0] = 0;
      e.busy[d0] = (uint8_t)cg_randint(e.draw(CG_SITE_STALL_ISOLATE, d0, n_mult - 1), 3, P.c.default_high + 3);
    }
    cost += -3.0 * n_mult * ds;
    fe[CG_D_CLEAN_COST] += 3.0 * n_mult * ds;
    fe[CG_D_DEF_COST] += 3.0 * n_mult * ds;
    wsync();
  }
}

#endif  // CG_DEFENDER_HPP
