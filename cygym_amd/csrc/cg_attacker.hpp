// cg_attacker.hpp -- Attacker actions: spread fix point and probe (volt_typhoon_env.py:1126-1202).
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_ATTACKER_HPP
#define CG_ATTACKER_HPP

// ---------------- attacker ----------------
#define T_INF 0xFFFFFFFFu
constexpr int LONG_ROW = 8;
#define CG_D_FULLROW 0x04  // library-private static bit: the out-row is "every other device, ascending"

// T[v] = (first-compromise time << 2) | eligibility bits (bit0 reachable, bit1 known & vulnerable to this
// exploit): one LDS word answers "can source s take v".  atomicMin keeps the (constant) low bits intact.
// The word is 32 bits at the compile-time sizes and 16 bits at run-time sizes (times <= 2049 there: M <= 2048), where the
// table's 4 KB decide whether a CU holds four or five envs of 2048 devices with an extra-edge list: LDS has no 16-bit
// atomic minimum, so a take is a compare-and-swap loop on the 32-bit word that holds the entry.
#define T_TIME_INF 0x3FFFFFFFu
template <bool RT> struct TWord { using type = uint32_t; };
template <> struct TWord<true> { using type = uint16_t; };
template <class TW> __device__ __forceinline__ constexpr uint32_t t_inf() { return sizeof(TW) == 2 ? 0x3FFFu : T_TIME_INF; }
template <class TW>
__device__ __forceinline__ bool spread_ok(const TW* T, int v, int s) {
  const uint32_t t = T[v];
  return (t & 1u) || ((t & 2u) && ((t >> 2) >= (uint32_t)(s + 1)));
}
// minimum into T[v]; returns the old entry
__device__ __forceinline__ uint32_t t_min(uint32_t* T, int v, uint32_t val) { return atomicMin(&T[v], val); }
__device__ __forceinline__ uint32_t t_min(uint16_t* T, int v, uint32_t val) {
  uint32_t* w = (uint32_t*)(T + (v & ~1));
  const int sh = (v & 1) * 16;
  uint32_t old = *w;
  for (;;) {
    const uint32_t cur = (old >> sh) & 0xFFFFu;
    if (cur <= val) return cur;
    const uint32_t prev = atomicCAS(w, old, (old & ~(0xFFFFu << sh)) | (val << sh));
    if (prev == old) return cur;
    old = prev;
  }
}
template <class TW>
__device__ __forceinline__ void spread_take(TW* T, int v, int s) {
  t_min(T, v, ((uint32_t)(s + 1) << 2) | ((uint32_t)T[v] & 3u));
}
// A take that reports whether `v` had ALREADY been taken by another source in this tick (its old time is neither "never"
// nor "compromised before the action"): one of the two then holds a pick the other one overrides -- a conflict that a
// verification sweep has to settle.  A sweep without a single conflict leaves every pick standing: it IS the fix point,
// no confirming sweep needed (see attacker_spread).
template <class TW>
__device__ __forceinline__ bool spread_take_c(TW* T, int v, int s, uint32_t low) {
  const uint32_t old = t_min(T, v, ((uint32_t)(s + 1) << 2) | low) >> 2;
  return old != t_inf<TW>() && old != 0u;
}
// first slot k in [from, o1) that source s can take, or o1
template <class TW>
__device__ __forceinline__ int spread_scan_lane(const Env& e, const TW* T, int s, bool dc,
                                                int from, int o1) {
  if (from >= o1) return o1;
  // the row (<= LONG_ROW slots) spans at most two words of the blocked bitmask: read them once, not per slot
  const int w0 = from >> 5;
  uint64_t bits = e.blk[w0];
  if (((o1 - 1) >> 5) != w0) bits |= (uint64_t)e.blk[w0 + 1] << 32;
  bits >>= (from & 31);   // bit i <-> slot from + i
  for (int k = from; k < o1; ++k, bits >>= 1) {
    if (bits & 1ull) continue;
    if (dc || spread_ok(T, e.ocol[k], s)) return k;
  }
  return o1;
}
// The same with v / low = the device and the constant low bits of T[v], STAGED (run-time sizes, where one resident wave
// per SIMD leaves every LDS round trip exposed; at 256 devices with 4 waves per SIMD it measured -1 %): the row's blocked words and its first four neighbours are read together, then the four T words -- two LDS
// round trips for the usual two- or three-slot row instead of two per slot.
template <class TW>
__device__ __forceinline__ int spread_scan_lane_st(const Env& e, const TW* T, int s, bool dc,
                                                int from, int o1, int& v, uint32_t& low) {
  if (from >= o1) return o1;
  // the row (<= LONG_ROW slots) spans at most two words of the blocked bitmask
  const int w0 = from >> 5;
  const bool two = ((o1 - 1) >> 5) != w0;
  const uint32_t b_lo = e.blk[w0], b_hi = e.blk[two ? w0 + 1 : w0];
  int vv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) vv[j] = e.ocol[from + j < o1 ? from + j : o1 - 1];
  uint32_t tt[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) tt[j] = T[vv[j]];
#pragma unroll
  for (int j = 0; j < 4; ++j) PIN(tt[j]);
  uint64_t bits = (uint64_t)b_lo | (two ? (uint64_t)b_hi << 32 : 0ull);
  bits >>= (from & 31);   // bit i <-> slot from + i
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const bool okv = (tt[j] & 1u) || ((tt[j] & 2u) && ((tt[j] >> 2) >= (uint32_t)(s + 1)));
    if (from + j < o1 && !((bits >> j) & 1ull) && (dc || okv)) { v = vv[j]; low = tt[j] & 3u; return from + j; }
  }
  bits >>= 4;
  for (int k = from + 4; k < o1; ++k, bits >>= 1) {
    if (bits & 1ull) continue;
    const int vk = e.ocol[k];
    const uint32_t tk = T[vk];
    if (dc || (tk & 1u) || ((tk & 2u) && ((tk >> 2) >= (uint32_t)(s + 1)))) { v = vk; low = tk & 3u; return k; }
  }
  return o1;
}
// same for a FULL row (slot k <-> device v = k - o0 + (k - o0 >= s)): walk the candidate-device bitmask
// instead of the row; `cand` holds reach | (known & vulnerable & not compromised at the start)
template <class TW>
__device__ __forceinline__ int spread_scan_full(const Env& e, const TW* T, const uint64_t* cand,
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
template <class TW>
__device__ __forceinline__ int spread_scan_coop(const Env& e, const TW* T, int s, bool dc,
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
template <class TW>
__device__ __forceinline__ bool spread_x_round(const Env& e, TW* T, uint16_t* cur, const uint16_t* slist, int n_src, int round) {
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
        if (m == last && dc) cby_or(e, ex ? (int)w.vx : (int)e.ocol[w.k], ebit);   // DC attribution :1163-1168
        w.next(e, ex);
      }
      cntv[i] = (uint16_t)n;
    }
    total += wave_sum(n);
  }
  return total;
}

// CR: long rows scanned per cooperative step (4 where the registers allow it: run-time sizes in workgroups of <= 8 waves;
// CR > 1 also selects the staged per-lane scan)
// GS: chunks per staged group of a chunk loop (4, or the whole env at a compile-time size)
// WIDE: the log counts read a row's blocked words nine at a time (rows of <= 256 slots, see range_popc_wide)
template <bool XE, int CR, int GS, bool WIDE, class TW, class KP>
__device__ __forceinline__ void attacker_spread(Env& e, const KP& P, const int32_t* expl, int ex0, int n_expl,
                                                uint64_t* srcb) {
  const int M = e.M, MC = e.MC, Mp = MC * WAVE;
  TW* T = (TW*)e.scr;                           // [Mp] first-compromise time (source id + 1); TW: see t_min above
  uint16_t* cur = (uint16_t*)(T + Mp);          // [Mp] current pick (slot) per source DEVICE, row end = none
  uint16_t* cntv = (uint16_t*)T;                // [Mp] log entries per COMPACT source index: reuses T, which is applied
                                                //      to the flags right after the sweeps (4 KB less LDS per 2048-device env)
  uint16_t* slist = e.lsrc;                     // [Mp] the sources in id order (snapshot :1127)
  uint64_t* cand = (uint64_t*)e.marks;          // [MC] candidate-device bitmask for full rows
  int n_src = 0;
#pragma nounroll
  for (int c0 = 0; c0 < MC; c0 += GS) {   // the sources in id order (compact list); staged like every chunk loop
    uint64_t smj[GS];
#pragma unroll
    for (int j = 0; j < GS; ++j) smj[j] = c0 + j < MC ? srcb[c0 + j] : 0ull;
#pragma unroll
    for (int j = 0; j < GS; ++j) {
      if ((smj[j] >> e.lane) & 1ull) slist[n_src + below(smj[j])] = (uint16_t)((c0 + j) * WAVE + e.lane);
      n_src += __popcll(smj[j]);
    }
  }
  wsync();
  const bool xany = XE && x_cnt(e) > 0;   // this env has added edges: their sources walk merged rows
  int zocc = 0;
  for (int j = 0; j < n_expl; ++j) {
    int raw = j == 0 ? ex0 : expl[j];   // expl[0] was fetched with the tick's header
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
#pragma nounroll
    for (int c0 = 0; c0 < MC; c0 += GS) {
      uint32_t fj[GS], vj[GS], oj[GS];   // (32-bit scalars: byte arrays were left in scratch by one instantiation)
#pragma unroll
      for (int j = 0; j < GS; ++j) {
        const int d = (c0 + j) * WAVE + e.lane, dc = d < M ? d : 0;
        fj[j] = e.flags[dc]; vj[j] = e.vul[dc]; oj[j] = e.optr[dc];
      }
#pragma unroll
      for (int j = 0; j < GS; ++j) {
        const int d = (c0 + j) * WAVE + e.lane;
        const uint8_t f = d < M ? (uint8_t)fj[j] : (uint8_t)0;
        const uint8_t x = (uint8_t)(((f & CG_F_REACH) ? 1 : 0) | (((f & CG_F_KNOWN) && (vj[j] & ebit)) ? 2 : 0));
        const uint64_t cm = ballot((x & 1) || ((x & 2) && !(f & CG_F_COMP)));
        if (c0 + j < MC) {
          T[d] = (TW)((((f & CG_F_COMP) ? 0u : t_inf<TW>()) << 2) | x);
          cur[d] = d < M ? (uint16_t)oj[j] : (uint16_t)0;
          if (e.lane == 0) cand[c0 + j] = cm;
        }
      }
    }
    wsync();
    SUBSTAMP(10);
    [[maybe_unused]] int n_rounds = 0;   // read by the stamps of diagnostic builds
#ifdef CG_STAMPS
    int dg_full = 0, dg_coop = 0, dg_lost = 0, dg_resc = 0;   // full-row sources, cooperative rows, conflicting takes, rescans in sweeps >= 1
#endif
    // Sweep 0: every source scans its row from the start and takes its pick.  Sweeps >= 1: a source whose pick an
    // EARLIER source took (the only way to lose one: dc sources and reachable targets never lose) resumes behind it.
    // The sources are in id order and a take only matters to later sources, so within a sweep the blocks run in
    // ascending order and see each other's takes.  Termination: a pick can only be invalidated by ANOTHER source taking
    // the same target, and every take reports whether its target had been taken before (atomicMin's old value): a sweep
    // in which no take hit a taken target -- very often sweep 0 itself -- has left every pick standing, which is the fix
    // point; only sweeps with such a conflict are followed by a verification sweep (round 2 always ran a confirming one).
    for (int sweep = 0; sweep <= M + 1; ++sweep) {
      ++n_rounds;
      bool lost_any = false;    // some pick is not settled yet: another sweep is needed
#ifndef CG_SPREAD_PRECHECK
#define CG_SPREAD_PRECHECK 1
#endif
      // Sweeps >= 1 first ask, for ALL blocks with the loads of four blocks in flight together, which sources lost their pick
      // (bit b of `need` <-> block b: at most 32 blocks); the block loop below then runs only for blocks that hold one.  A
      // verification sweep used to walk every block through its four dependent round trips -- 14 blocks, ~40 k cycles at 2048
      // devices -- to find the three or four sources that have to resume (profiles/r04_tail_hist_cfg5.txt).  A resumed source's
      // take that reaches a LATER source's target is reported as a conflict by the take itself, so the next sweep re-checks.
      uint32_t need = 0xFFFFFFFFu;
      if (CG_SPREAD_PRECHECK && sweep > 0) {
        need = 0u;
        constexpr int PB = 4;
#pragma nounroll
        for (int bb = 0; bb * WAVE < n_src; bb += PB) {
          int sj[PB], kj[PB], oj0[PB], oj1[PB];
          uint32_t stj[PB];
#pragma unroll
          for (int j = 0; j < PB; ++j) { const int i = (bb + j) * WAVE + e.lane; sj[j] = slist[i < n_src ? i : 0]; }
#pragma unroll
          for (int j = 0; j < PB; ++j) { oj0[j] = e.optr[sj[j]]; oj1[j] = e.optr[sj[j] + 1]; stj[j] = e.dst[sj[j]]; kj[j] = (int)cur[sj[j]]; }
          int vj[PB];
#pragma unroll
          for (int j = 0; j < PB; ++j) {
            const bool full = (oj1[j] - oj0[j] > LONG_ROW) && (stj[j] & CG_D_FULLROW) && !(stj[j] & CG_D_DC);
            const int kc = kj[j] < oj1[j] ? kj[j] : (oj1[j] > 0 ? oj1[j] - 1 : 0);
            vj[j] = full ? (kj[j] - oj0[j]) + ((kj[j] - oj0[j]) >= sj[j] ? 1 : 0) : (int)e.ocol[kc];
            if (vj[j] >= M) vj[j] = M - 1;
          }
          uint32_t tj[PB];
#pragma unroll
          for (int j = 0; j < PB; ++j) tj[j] = T[vj[j]];
#pragma unroll
          for (int j = 0; j < PB; ++j) {
            const int i = (bb + j) * WAVE + e.lane;
            const bool okv = (tj[j] & 1u) || ((tj[j] & 2u) && ((tj[j] >> 2) >= (uint32_t)(sj[j] + 1)));
            const bool lost = i < n_src && kj[j] < oj1[j] && !(stj[j] & CG_D_DC) && !okv;
            const bool xr = XE && COLD(xany) && i < n_src && x_isout(e, sj[j]);   // (rows with added edges: spread_x_round's business)
            if (__any(lost && !xr)) need |= 1u << ((bb + j) & 31);
          }
        }
      }
      for (int b0 = 0; b0 < n_src; b0 += WAVE) {
        if (!((need >> ((b0 / WAVE) & 31)) & 1u)) continue;   // (uniform) nobody in this block has to resume
        const int i = b0 + e.lane;
        bool coop = false;
        int s = 0, o1 = 0, k0 = 0;
        uint8_t st = 0;
        if (i < n_src) {
          s = slist[i];
          const int o0 = e.optr[s];
          o1 = e.optr[s + 1]; st = e.dst[s];
          k0 = sweep == 0 ? o0 : (int)cur[s];
          const bool dc = st & CG_D_DC;
          const bool shortrow = o1 - o0 <= LONG_ROW;
          const bool full = !shortrow && (st & CG_D_FULLROW) && !dc;
          const bool xrow = COLD(xany && x_isout(e, s));   // row with added edges: handled by spread_x_round below
          bool rescan = sweep == 0;
          if (sweep > 0 && k0 < o1 && !dc && !xrow) {
            const int vc = full ? (k0 - o0) + ((k0 - o0) >= s ? 1 : 0) : (int)e.ocol[k0];
            rescan = !spread_ok(T, vc, s);
          }
          if (rescan && !xrow) {
#ifndef CG_LONG_PREFIX
#define CG_LONG_PREFIX 1
#endif
            // A long row that is not "full" (a hub's 47 slots at 2048 devices) first has its next four slots scanned by its own lane,
            // with -- and by the same instructions as -- the short rows of the block; only if none of them is a pick does the rest
            // of the row take a cooperative step.  (Those steps, four rows each and ~90 rows per exploit, were 40 us of a 165 us
            // tick at 4096 x 2048: timing ablation, PERFLOG.md; the usual pick is among a hub's first neighbours.)
            const int from = sweep == 0 ? o0 : k0 + 1;
            const bool prefix = CG_LONG_PREFIX && CR > 1 && !shortrow && !full;
            if (shortrow || full || prefix) {
              const int lim1 = prefix ? (from + 4 < o1 ? from + 4 : o1) : o1;
              uint32_t low = 0;
              int k, v = 0;
              if (!full) {
                if constexpr (CR > 1) k = spread_scan_lane_st(e, T, s, dc, from, lim1, v, low);
                else { k = spread_scan_lane(e, T, s, dc, from, lim1); if (k < lim1) { v = e.ocol[k]; low = T[v] & 3u; } }
              }
              else {
                k = spread_scan_full(e, T, cand, s, from, o0, o1);
                if (k < o1) { v = (k - o0) + ((k - o0) >= s ? 1 : 0); low = T[v] & 3u; }
              }
              if (prefix && k >= lim1 && lim1 < o1) {
#ifndef CG_ABL_NO_COOP_ROWS
                coop = true; k0 = lim1;   // the rest of the row, cooperatively, from here
#endif
              } else {
                cur[s] = (uint16_t)k;
                if (k < o1 && spread_take_c(T, v, s, low)) lost_any = true;
              }
            } else {
#ifndef CG_ABL_NO_COOP_ROWS
              coop = true; k0 = from;
#endif
            }
#ifdef CG_STAMPS
            if (sweep == 0) { dg_full += full ? 1 : 0; dg_coop += coop ? 1 : 0; } else dg_resc += 1;
#endif
          }
        }
        uint64_t nm = ballot(coop);   // long rows that are not "full": wave-cooperative (re)scan
        // Up to CR (four) rows per step: the first 64 slots of each are scanned by one ballot, and the four scans are
        // independent LDS chains that overlap (one row at a time paid ~700 cycles of dependent latency per row:
        // at 2048 devices ~100 such rows per sweep were half of the spread).  Rows of one step do not see each
        // other's takes -- like the lanes of the per-lane path, the verification sweeps settle that.
        while (nm) {
          int rs[CR], ro1[CR], rfrom[CR], rk[CR];
          bool rdc[CR], rhave[CR];
#pragma unroll
          for (int j = 0; j < CR; ++j) {
            rhave[j] = nm != 0;
            const int src_lane = rhave[j] ? __builtin_ctzll(nm) : 0;
            nm &= nm - 1;   // (0 stays 0)
            rs[j] = __builtin_amdgcn_readlane(s, src_lane);
            ro1[j] = rhave[j] ? __builtin_amdgcn_readlane(o1, src_lane) : 0;
            rfrom[j] = __builtin_amdgcn_readlane(k0, src_lane);   // (k0 of a cooperative lane: the slot its scan starts at)
            rdc[j] = __builtin_amdgcn_readlane((int)st, src_lane) & CG_D_DC;
          }
          // staged and branch-free: the loads of one stage (blocked word + neighbour, then its T word) are issued for
          // all four rows before the first is used (clamped slots; a short-circuit chain would serialise them)
          uint64_t rm[CR];
          uint32_t rbw[CR], rtv[CR];
          int rkc[CR], rv[CR];
#pragma unroll
          for (int j = 0; j < CR; ++j) {
            const int k = rfrom[j] + e.lane;
            rkc[j] = k < ro1[j] ? k : (ro1[j] > 0 ? ro1[j] - 1 : 0);
            rbw[j] = e.blk[rkc[j] >> 5];
            rv[j] = e.ocol[rkc[j]];
          }
#pragma unroll
          for (int j = 0; j < CR; ++j) rtv[j] = T[rv[j]];
#pragma unroll
          for (int j = 0; j < CR; ++j) {
            const int k = rfrom[j] + e.lane;
            const bool okv = (rtv[j] & 1u) || ((rtv[j] & 2u) && ((rtv[j] >> 2) >= (uint32_t)(rs[j] + 1)));
            rm[j] = ballot(k < ro1[j] && !((rbw[j] >> (rkc[j] & 31)) & 1u) && (rdc[j] || okv));
          }
#pragma unroll
          for (int j = 0; j < CR; ++j) {
            if (rm[j]) rk[j] = rfrom[j] + __builtin_ctzll(rm[j]);
            else rk[j] = rfrom[j] + WAVE < ro1[j] ? spread_scan_coop(e, T, rs[j], rdc[j], rfrom[j] + WAVE, ro1[j]) : ro1[j];
          }
          if (e.lane < CR) {   // lane j records the pick of row j
            const int j = e.lane;
            bool hv = rhave[0]; int ls = rs[0], k = rk[0], lo1 = ro1[0];
#pragma unroll
            for (int q = 1; q < CR; ++q) if (j == q) { hv = rhave[q]; ls = rs[q]; k = rk[q]; lo1 = ro1[q]; }
            if (hv) {
              cur[ls] = (uint16_t)k;
              if (k < lo1) { const int tv = e.ocol[k]; if (spread_take_c(T, tv, ls, T[tv] & 3u)) lost_any = true; }
            }
          }
        }
        if (sweep > 0) wsync();   // the next block must see this block's takes
      }
      // (rows with added edges keep the conservative rule: any change of such a pick, and their whole first sweep, asks for
      // a verification sweep)
      if constexpr (XE) { if (COLD(xany)) { if (spread_x_round(e, T, cur, slist, n_src, sweep) || sweep == 0) lost_any = true; } }
      wsync();
#ifdef CG_STAMPS
      dg_lost += lost_any ? 1 : 0;
#endif
      if (!__any(lost_any)) break;
    }
    SUBSTAMP(11);
    SUBVAL(15, n_rounds);
#ifdef CG_STAMPS
    { const int a = wave_sum(dg_full), b = wave_sum(dg_coop), c = wave_sum(dg_lost), d = wave_sum(dg_resc);
      SUBVAL(16, n_src); SUBVAL(17, a | (b << 16)); SUBVAL(18, c | (d << 16)); }
#endif
    // apply the compromise flags (:1163-1185) now: T is dead from here on (cntv reuses its storage)
#pragma nounroll
    for (int c0 = 0; c0 < MC; c0 += GS) {
      uint32_t tj[GS], fj[GS];
#pragma unroll
      for (int j = 0; j < GS; ++j) { const int d = (c0 + j) * WAVE + e.lane, dc = d < M ? d : 0; tj[j] = T[dc]; fj[j] = e.flags[dc]; }
#pragma unroll
      for (int j = 0; j < GS; ++j) {
        const int d = (c0 + j) * WAVE + e.lane;
        if (d < M && (tj[j] >> 2) != t_inf<TW>() && (tj[j] >> 2) != 0u) e.flags[d] = (uint8_t)(fj[j] | CG_F_COMP);
      }
    }
    wsync();
    // log entries of every source: unblocked out-entries up to and including its pick
    // ... and, in the same pass, the domain-controller attribution (:1163-1185): a DC source marks its pick
    int total_new = 0;
#ifndef CG_COUNTS_STAGED
#define CG_COUNTS_STAGED 1
#endif
    if constexpr (CG_COUNTS_STAGED && CR > 1) {
      // four blocks of sources per step, each stage's loads in flight together (source id -> row bounds + pick + static byte ->
      // the blocked-word pair of the prefix): three round trips per FOUR blocks where the loop below pays five per block; a prefix
      // of more than 33 slots (a hub without a target) takes the rolled count, only when some lane of the step has one.  One
      // wave sum per step.  (14 blocks per exploit at 2048 devices: 17 k cycles of a spread env's 119 k.)
      constexpr int CB = 4;
      const int wl = ((P.t.EW + 3) & ~3) - 1;
#pragma nounroll
      for (int bb = 0; bb * WAVE < n_src; bb += CB) {
        int sj[CB], oj0[CB], oj1[CB], cj[CB];
        uint32_t stj[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) { const int i = (bb + j) * WAVE + e.lane; sj[j] = slist[i < n_src ? i : 0]; }
#pragma unroll
        for (int j = 0; j < CB; ++j) { oj0[j] = e.optr[sj[j]]; oj1[j] = e.optr[sj[j] + 1]; cj[j] = (int)cur[sj[j]]; stj[j] = e.dst[sj[j]]; }
        uint32_t blo[CB], bhi[CB];
        int endj[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          endj[j] = cj[j] < oj1[j] ? cj[j] + 1 : oj1[j];
          const int w0 = oj0[j] >> 5;
          blo[j] = e.blk[w0]; bhi[j] = e.blk[w0 < wl ? w0 + 1 : w0];
        }
        int nsum = 0;
#pragma unroll
        for (int j = 0; j < CB; ++j) {
          const int i = (bb + j) * WAVE + e.lane;
          const bool valid = i < n_src && !(XE && COLD(xany) && x_isout(e, sj[j]));   // (rows with added edges: spread_x_counts below)
          const int len = endj[j] - oj0[j];
          const bool far = valid && len > 33;
          int n = 0;
          if (valid && !far) {
            const uint64_t bits = ((uint64_t)blo[j] | ((uint64_t)bhi[j] << 32)) >> (oj0[j] & 31);
            n = len - __popcll(len > 0 ? bits & (~0ull >> (64 - len)) : 0ull);
          }
          if (__any(far)) { if (far) n = len - (WIDE ? range_popc_wide(e.blk, oj0[j], endj[j]) : range_popc(e.blk, oj0[j], endj[j])); }
          if (valid && (stj[j] & CG_D_DC) && cj[j] < oj1[j]) cby_or(e, e.ocol[cj[j]], ebit);
          if (i < Mp) cntv[i] = (uint16_t)n;
          nsum += n;
        }
        total_new += wave_sum(nsum);
      }
    } else {
    for (int b0 = 0; b0 < n_src; b0 += WAVE) {
      const int i = b0 + e.lane;
      int n = 0;
      if (i < n_src) {
        const int s = slist[i];
        const int o0 = e.optr[s], o1 = e.optr[s + 1], c = cur[s];
        const uint8_t st = e.dst[s];
        if (COLD(xany && x_isout(e, s))) {   // counted (and attributed) by spread_x_counts below
        } else {
          const int end = c < o1 ? c + 1 : o1;
          n = (end - o0) - (WIDE ? range_popc_wide(e.blk, o0, end) : range_popc(e.blk, o0, end));
          if ((st & CG_D_DC) && c < o1) cby_or(e, e.ocol[c], ebit);
        }
      }
      cntv[i < Mp ? i : 0] = (uint16_t)n;
      total_new += wave_sum(n);
    }
    }
    wsync();
    if constexpr (XE) { if (COLD(xany)) { total_new += spread_x_counts(e, cur, slist, cntv, n_src, ebit); wsync(); } }
    SUBSTAMP(12);
    // ring: only the last CG_LOG_RING entries (global order: source id, then row order) matter; the long history
    // (cygym_buffers.hist, full-feature kernel only) takes the last CG_HIST_RING of them, straight to global memory
    if (total_new > 0) {
      const uint32_t base = (uint32_t)e.log_total;
      const uint32_t end = base + (uint32_t)total_new;
      const uint32_t lo = end > CG_LOG_RING ? end - CG_LOG_RING : 0;
      uint16_t* hist = nullptr;
      if constexpr (XE) { if (COLD(P.b.hist != nullptr)) hist = P.b.hist + (size_t)e.env * CG_HIST_RING * 2; }
      const uint32_t lo_all = hist ? (end > CG_HIST_RING ? end - CG_HIST_RING : 0) : lo;   // lo_all <= lo
      auto put = [&](uint32_t idx, int from, int to) {
        if (idx >= lo) { e.ring[2 * (idx % CG_LOG_RING)] = (uint16_t)from; e.ring[2 * (idx % CG_LOG_RING) + 1] = (uint16_t)to; }
        if constexpr (XE) {
          if (COLD(hist != nullptr) && idx >= lo_all) { hist[2 * (idx % CG_HIST_RING)] = (uint16_t)from; hist[2 * (idx % CG_HIST_RING) + 1] = (uint16_t)to; }
        }
      };
      uint32_t after = end;   // global index just past the current block of sources
      for (int b0 = ((n_src - 1) / WAVE) * WAVE; b0 >= 0 && after > lo_all; b0 -= WAVE) {
        const int i = b0 + e.lane;
        int n = i < n_src ? cntv[i] : 0;
        int incl = wave_incl_scan(n, e.lane);
        int blk_total = __builtin_amdgcn_readlane(incl, 63);
        uint32_t cbase = after - (uint32_t)blk_total;
        uint32_t off = cbase + (uint32_t)(incl - n);
        bool mine = n > 0 && off + (uint32_t)n > lo_all;
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
              put(idx, s, v);
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
            put(idx, s, e.ocol[k]);
            ++idx;
          }
        }
        uint64_t lm = ballot(is_long);
        while (lm) {
          int ll = __builtin_ctzll(lm);
          lm &= lm - 1;
          int ls = __builtin_amdgcn_readlane(s, ll);
          uint32_t idx0 = (uint32_t)__builtin_amdgcn_readlane((int)off, ll);
          int o0 = e.optr[ls], o1 = e.optr[ls + 1];
          int last = cur[ls] < o1 ? (int)cur[ls] : o1 - 1;
          for (int k0 = o0; k0 <= last; k0 += WAVE) {
            int k = k0 + e.lane;
            bool p = k <= last && !e.blocked(k);
            uint64_t m = ballot(p);
            uint32_t idx = idx0 + (uint32_t)below(m);
            if (p) put(idx, ls, e.ocol[k]);
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
    SUBSTAMP(14);
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The spread at a compile-time size without added edges (lean kernels, 64 / 256 devices), written against what the launch
// time of a one-residency-round batch is made of: the slowest env's chain of DEPENDENT LDS round trips (~300 cycles each
// while all sixteen waves of a CU are alive) and its instruction count (profiles/r04_tail_hist.txt: the spread env that
// finishes last lives 39 k cycles; a sweep costs 3-6 k).  Same fix point as attacker_spread below -- any processing order
// reaches it: a sweep without a conflicting take leaves every pick standing -- but:
//   * the sources are compacted ONCE, straight from the ballots of the flag bytes (no source bitmask round trip);
//   * lane i of block b owns source b * 64 + i for the whole action: its row bounds, static byte, pick and entry count live
//     in registers (three VGPRs per block) -- no `cur` / `cntv` arrays, no pass re-reads slist / optr / dst / cur;
//   * sweep 0 scans a row in stages, two blocks at a time: [source id] -> [row bounds + static byte] -> [first four
//     neighbours + the row's blocked words] -> [their T words]; a full row (attacker-owned hub: every other device,
//     ascending) rides in the same stages -- its first candidate comes from the candidate masks (uniform), so its
//     stage-3 / 4 reads are the blocked word and T word of that one slot;
//   * a verification sweep (only after a conflict) is ONE round trip for all blocks -- the T words of the picks -- and
//     only the sources that lost theirs rescan.
template <int MCT, bool WIDE, class KP>
__device__ __forceinline__ void attacker_spread_ct(Env& e, const KP& P, const int32_t* expl, int ex0, int n_expl) {
  const int M = e.M, E1 = P.t.E > 0 ? P.t.E - 1 : 0;
  const int lane = e.lane;
  uint32_t* T = e.scr;                          // [Mp] first-compromise time (source id + 1) << 2 | eligibility bits
  uint16_t* slist = e.lsrc;                     // [Mp] the sources in id order (snapshot :1127)
  uint64_t* cand = (uint64_t*)e.marks;          // [MC] candidate-device bitmask (LDS copy for the per-lane rescans)
  // :1127 snapshot of the sources, once per action: compact list in id order
  int n_src = 0;
  {
    uint32_t fj[MCT];
#pragma unroll
    for (int c = 0; c < MCT; ++c) fj[c] = e.flags[c * WAVE + lane < M ? c * WAVE + lane : 0];
#pragma unroll
    for (int c = 0; c < MCT; ++c) {
      const int d = c * WAVE + lane;
      const uint64_t m = ballot(d < M && (fj[c] & (CG_F_COMP | CG_F_OWNED)));
      if ((m >> lane) & 1ull) slist[n_src + below(m)] = (uint16_t)d;
      n_src += __popcll(m);
    }
  }
  int zocc = 0;
  for (int jx = 0; jx < n_expl; ++jx) {
    int raw = jx == 0 ? ex0 : expl[jx];   // expl[0] was fetched with the tick's header
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
    // ---- T words and candidate masks from the flag / vulnerability bytes (device-major, one round trip) ----
    int u1 = -1, u2 = -1;   // the two lowest candidates (uniform): a full row's first pick is the lowest one that is not itself
    {
      uint32_t fj[MCT], vj[MCT];
#pragma unroll
      for (int c = 0; c < MCT; ++c) { const int d = c * WAVE + lane, dc = d < M ? d : 0; fj[c] = e.flags[dc]; vj[c] = e.vul[dc]; }
#pragma unroll
      for (int c = 0; c < MCT; ++c) {
        const int d = c * WAVE + lane;
        const uint32_t f = d < M ? fj[c] : 0u;
        const uint32_t x = ((f & CG_F_REACH) ? 1u : 0u) | (((f & CG_F_KNOWN) && (vj[c] & ebit)) ? 2u : 0u);
        uint64_t cm = ballot((x & 1u) || ((x & 2u) && !(f & CG_F_COMP)));
        T[d] = (((f & CG_F_COMP) ? 0u : T_TIME_INF) << 2) | x;
        if (lane == 0) cand[c] = cm;
        if (u1 < 0 && cm) { u1 = c * WAVE + __builtin_ctzll(cm); cm &= cm - 1; }
        if (u1 >= 0 && u2 < 0 && cm) u2 = c * WAVE + __builtin_ctzll(cm);
      }
    }
    wsync();
    SUBSTAMP(10);
    // ---- per-source registers: block b, lane i <-> source b * 64 + i ----
    uint32_t sst[MCT];   // source id | static byte << 16
    uint32_t row[MCT];   // o0 | o1 << 16
    uint32_t pkv[MCT];   // pick: slot k (o1 = none) | target device << 16
#pragma unroll
    for (int b = 0; b < MCT; ++b) { sst[b] = 0; row[b] = 0; pkv[b] = 0; }
    bool lost_any = false;     // some take hit a target another source had taken in this tick: a verification sweep is due
    constexpr int NS = 4;      // slots staged per row
    constexpr int BP = (WIDE && MCT >= 2) ? 2 : 1;   // blocks per staged group (two where the registers allow it: the kernels capped at 80 VGPRs spill with two)
#pragma unroll
    for (int b0 = 0; b0 < MCT; b0 += BP) {
      if (b0 * WAVE >= n_src) break;
      int sv[BP], o0[BP], o1[BP];
      uint32_t stv[BP];
      bool act[BP];
#pragma unroll
      for (int q = 0; q < BP; ++q) { const int i = (b0 + q) * WAVE + lane; act[q] = i < n_src; sv[q] = slist[act[q] ? i : 0]; }
#pragma unroll
      for (int q = 0; q < BP; ++q) { o0[q] = e.optr[sv[q]]; o1[q] = e.optr[sv[q] + 1]; stv[q] = e.dst[sv[q]]; }
      uint32_t b_lo[BP], b_hi[BP];
      int vv[BP][NS];
      int kf[BP], v0[BP];
      bool full[BP];
#pragma unroll
      for (int q = 0; q < BP; ++q) {
        const int len = o1[q] - o0[q];
        const bool dc = stv[q] & CG_D_DC;
        full[q] = len > LONG_ROW && (stv[q] & CG_D_FULLROW) && !dc;
        v0[q] = u1 == sv[q] ? u2 : u1;                                  // a full row's first candidate ...
        kf[q] = o0[q] + (v0[q] > 0 ? v0[q] : 0) - (v0[q] > sv[q] ? 1 : 0);   // ... sits at this slot
        const int base = full[q] ? kf[q] : o0[q];
        const int w0 = base >> 5, wl = ((P.t.EW + 3) & ~3) - 1;
        b_lo[q] = e.blk[w0]; b_hi[q] = e.blk[w0 < wl ? w0 + 1 : w0];
#pragma unroll
        for (int j = 0; j < NS; ++j) { const int k = o0[q] + j; vv[q][j] = e.ocol[k < E1 ? k : E1]; }
      }
      uint32_t tt[BP][NS];
#pragma unroll
      for (int q = 0; q < BP; ++q) {
        if (full[q]) vv[q][0] = v0[q] > 0 ? v0[q] : 0;
#pragma unroll
        for (int j = 0; j < NS; ++j) { const int v = vv[q][j] < M ? vv[q][j] : M - 1; tt[q][j] = T[v]; }
      }
#pragma unroll
      for (int q = 0; q < BP; ++q)
#pragma unroll
        for (int j = 0; j < NS; ++j) PIN(tt[q][j]);
#pragma unroll
      for (int q = 0; q < BP; ++q) {
        const int s = sv[q];
        const int len = o1[q] - o0[q];
        const bool dc = stv[q] & CG_D_DC;
        const bool shortrow = len <= LONG_ROW;
        int k = o1[q], v = 0;
        uint32_t low = 0;
        if (act[q] && shortrow) {
          const uint64_t bits = ((uint64_t)b_lo[q] | ((uint64_t)b_hi[q] << 32)) >> (o0[q] & 31);   // bit i <-> slot o0 + i (rows of <= 8 slots)
          bool found = false;
#pragma unroll
          for (int j = 0; j < NS; ++j) {
            const uint32_t t = tt[q][j];
            const bool okv = (t & 1u) || ((t & 2u) && ((t >> 2) >= (uint32_t)(s + 1)));
            if (!found && j < len && !((bits >> j) & 1ull) && (dc || okv)) { found = true; k = o0[q] + j; v = vv[q][j]; low = t & 3u; }
          }
          if (!found && len > NS) {   // slots 4..7 of a short row: one by one (few rows are that long)
            uint64_t b2 = bits >> NS;
            for (int kk = o0[q] + NS; kk < o1[q]; ++kk, b2 >>= 1) {
              if (b2 & 1ull) continue;
              const int vk = e.ocol[kk];
              const uint32_t tk = T[vk];
              if (dc || (tk & 1u) || ((tk & 2u) && ((tk >> 2) >= (uint32_t)(s + 1)))) { k = kk; v = vk; low = tk & 3u; break; }
            }
          }
        } else if (act[q] && full[q] && v0[q] >= 0) {
          const uint32_t t = tt[q][0];
          const bool okv = (t & 1u) || ((t & 2u) && ((t >> 2) >= (uint32_t)(s + 1)));
          if (!((b_lo[q] >> (kf[q] & 31)) & 1u) && okv) { k = kf[q]; v = v0[q]; low = t & 3u; }
          else {   // the lowest candidate is blocked for this hub (or was taken a moment ago): walk on from behind it
            k = spread_scan_full(e, T, cand, s, kf[q] + 1, o0[q], o1[q]);
            if (k < o1[q]) { v = (k - o0[q]) + ((k - o0[q]) >= s ? 1 : 0); low = T[v] & 3u; }
          }
        }
        // long rows that are not "full" (e.g. a domain controller with many neighbours): one row at a time, 64 slots per step
        uint64_t nm = ballot(act[q] && !shortrow && !full[q]);
        while (nm) {
          const int sl = __builtin_ctzll(nm);
          nm &= nm - 1;
          const int rs = __builtin_amdgcn_readlane(s, sl), ro0 = __builtin_amdgcn_readlane(o0[q], sl), ro1 = __builtin_amdgcn_readlane(o1[q], sl);
          const bool rdc = __builtin_amdgcn_readlane((int)stv[q], sl) & CG_D_DC;
          const int kc = spread_scan_coop(e, T, rs, rdc, ro0, ro1);
          if (lane == sl) { k = kc; if (kc < ro1) { v = e.ocol[kc]; low = T[v] & 3u; } }
        }
        if (act[q] && k < o1[q] && spread_take_c(T, v, s, low)) lost_any = true;
        sst[b0 + q] = (uint32_t)s | (stv[q] << 16);
        row[b0 + q] = (uint32_t)o0[q] | ((uint32_t)o1[q] << 16);
        pkv[b0 + q] = (uint32_t)k | ((uint32_t)v << 16);
      }
    }
    wsync();
    [[maybe_unused]] int n_rounds = 1;   // read by the stamps of diagnostic builds
    // ---- sweeps >= 1, only while some take reported a conflict: every pick is verified (one round trip for all blocks);
    // a source whose pick an EARLIER source took rescans from behind it, block after block in ascending order
    while (__any(lost_any)) {
      ++n_rounds;
      lost_any = false;
      uint32_t tv[MCT];
#pragma unroll
      for (int b = 0; b < MCT; ++b) tv[b] = T[pkv[b] >> 16];
#pragma unroll
      for (int b = 0; b < MCT; ++b) {
        if (b * WAVE >= n_src) break;
        const int s = (int)(sst[b] & 0xFFFFu), o0 = (int)(row[b] & 0xFFFFu), o1 = (int)(row[b] >> 16), k0 = (int)(pkv[b] & 0xFFFFu);
        const uint32_t st = sst[b] >> 16;
        const bool dc = st & CG_D_DC;
        // lost: an earlier source holds the target (reachable targets and dc sources never lose)
        const bool lost = b * WAVE + lane < n_src && k0 < o1 && !dc && !(tv[b] & 1u) && (tv[b] >> 2) < (uint32_t)(s + 1);
        if (ballot(lost) == 0ull) continue;
        const bool shortrow = o1 - o0 <= LONG_ROW;
        const bool full = !shortrow && (st & CG_D_FULLROW);
        int k = k0, v = (int)(pkv[b] >> 16);
        if (lost && (shortrow || full)) {
          if (shortrow) { k = spread_scan_lane(e, T, s, false, k0 + 1, o1); v = k < o1 ? (int)e.ocol[k] : 0; }
          else { k = spread_scan_full(e, T, cand, s, k0 + 1, o0, o1); v = k < o1 ? (k - o0) + ((k - o0) >= s ? 1 : 0) : 0; }
          if (k < o1 && spread_take_c(T, v, s, T[v] & 3u)) lost_any = true;
        }
        uint64_t nm = ballot(lost && !shortrow && !full);
        while (nm) {
          const int sl = __builtin_ctzll(nm);
          nm &= nm - 1;
          const int rs = __builtin_amdgcn_readlane(s, sl), ro1 = __builtin_amdgcn_readlane(o1, sl), rk = __builtin_amdgcn_readlane(k0, sl);
          const int kc = spread_scan_coop(e, T, rs, false, rk + 1, ro1);
          if (lane == sl) {
            k = kc; v = 0;
            if (kc < ro1) { v = e.ocol[kc]; if (spread_take_c(T, v, rs, T[v] & 3u)) lost_any = true; }
          }
        }
        pkv[b] = (uint32_t)k | ((uint32_t)v << 16);
        wsync();   // the next block's rescans see these takes
      }
    }
    SUBSTAMP(11);
    SUBVAL(15, n_rounds);
    // ---- apply the compromise flags (:1163-1185) ----
    {
      uint32_t tj[MCT], fj[MCT];
#pragma unroll
      for (int c = 0; c < MCT; ++c) { const int d = c * WAVE + lane, dc = d < M ? d : 0; tj[c] = T[dc]; fj[c] = e.flags[dc]; }
#pragma unroll
      for (int c = 0; c < MCT; ++c) {
        const int d = c * WAVE + lane;
        if (d < M && (tj[c] >> 2) != T_TIME_INF && (tj[c] >> 2) != 0u) e.flags[d] = (uint8_t)(fj[c] | CG_F_COMP);
      }
    }
    // ---- log entries of every source: the unblocked out-entries up to and including its pick (+ the domain-controller
    // attribution of that pick, :1163-1168); the counts stay in registers
    int cnt[MCT];
    int total_new = 0;
#pragma unroll
    for (int b = 0; b < MCT; ++b) {
      cnt[b] = 0;
#ifdef CG_ABL_NO_CNT
      continue;
#endif
      if (b * WAVE >= n_src) continue;   // (uniform)
      const int o0 = (int)(row[b] & 0xFFFFu), o1 = (int)(row[b] >> 16), k = (int)(pkv[b] & 0xFFFFu);
      int n = 0;
      const bool mine = b * WAVE + lane < n_src;
      const int end = k < o1 ? k + 1 : o1;
      // the prefix [o0, end) of nearly every source -- short rows, and hubs whose pick is one of their first neighbours -- spans
      // at most two words of the blocked bitmask: two reads and one 64-bit popcount; the nine-word form only where some lane needs it
      const bool far = mine && end - o0 > 33;
      if (mine && !far) {
        const int w0 = o0 >> 5, wl = ((P.t.EW + 3) & ~3) - 1;
        const uint64_t bits = ((uint64_t)e.blk[w0] | ((uint64_t)e.blk[w0 < wl ? w0 + 1 : w0] << 32)) >> (o0 & 31);
        const int len = end - o0;   // 0..33
        n = len - __popcll(len > 0 ? bits & (~0ull >> (64 - len)) : 0ull);
      }
      if (__any(far)) { if (far) n = (end - o0) - (WIDE ? range_popc_wide(e.blk, o0, end) : range_popc(e.blk, o0, end)); }
      if (mine && ((sst[b] >> 16) & CG_D_DC) && k < o1) cby_or(e, (int)(pkv[b] >> 16), ebit);
      cnt[b] = n;
      total_new += wave_sum(n);
    }
    wsync();
    SUBSTAMP(12);
    // ring: only the last CG_LOG_RING entries (global order: source id, then row order) matter
#ifdef CG_ABL_NO_RING
    e.log_total += total_new;   // (timing ablation: the counts stay alive, the ring is not written)
    if (false) {
#else
    if (total_new > 0) {
#endif
      const uint32_t base = (uint32_t)e.log_total;
      const uint32_t end = base + (uint32_t)total_new;
      const uint32_t lo = end > CG_LOG_RING ? end - CG_LOG_RING : 0;
      uint32_t after = end;   // global index just past the current block of sources
#pragma unroll
      for (int b = MCT - 1; b >= 0; --b) {
        if (b * WAVE >= n_src) continue;   // (uniform)
        if (after <= lo) break;
        const int s = (int)(sst[b] & 0xFFFFu), o0 = (int)(row[b] & 0xFFFFu), o1 = (int)(row[b] >> 16), k = (int)(pkv[b] & 0xFFFFu);
        const int n = cnt[b];
        const int incl = wave_incl_scan(n, lane);
        const int blk_total = __builtin_amdgcn_readlane(incl, 63);
        const uint32_t cbase = after - (uint32_t)blk_total;
        const uint32_t off = cbase + (uint32_t)(incl - n);
        const bool mine = n > 0 && off + (uint32_t)n > lo;
        const int last = k < o1 ? k : o1 - 1;
        // (cooperative only for sources with many entries or a pick far into the row: a hub whose pick is among its first neighbours --
        // the usual case -- goes with the short rows, one step per ENTRY on the bits of its blocked-word pair)
        const bool is_long = mine && (last - o0 >= 33 || n > LONG_ROW);
#ifdef CG_ABL_RING_NOLANE
        if (false) {
#else
        if (mine && !is_long) {
#endif
          // a short row's entries = the clear bits of its blocked-word pair below the pick: one read of the pair, then one step per
          // ENTRY (not per slot, and no blocked-bit read inside the loop)
          const int w0 = o0 >> 5, wl = ((P.t.EW + 3) & ~3) - 1;
          const uint64_t bits = ((uint64_t)e.blk[w0] | ((uint64_t)e.blk[w0 < wl ? w0 + 1 : w0] << 32)) >> (o0 & 31);
          uint64_t z = ~bits & (~0ull >> (63 - (last - o0)));   // slots o0 .. last (rows of <= LONG_ROW slots)
          uint32_t idx = off;
          while (z) {
            const int kk = o0 + __builtin_ctzll(z);
            z &= z - 1;
            if (idx >= lo) { e.ring[2 * (idx % CG_LOG_RING)] = (uint16_t)s; e.ring[2 * (idx % CG_LOG_RING) + 1] = e.ocol[kk]; }
            ++idx;
          }
        }
#ifdef CG_ABL_RING_NOCOOP
        uint64_t lm = 0;
#else
        uint64_t lm = ballot(is_long);
#endif
        while (lm) {
          const int ll = __builtin_ctzll(lm);
          lm &= lm - 1;
          const int ls = __builtin_amdgcn_readlane(s, ll);
          // BACKWARDS from the pick, 64 slots per step, and only until the ring's window is covered: a hub that found no target
          // logs its whole row (255 entries at 256 devices) of which at most the last CG_LOG_RING are kept -- walked forwards that
          // was four steps per such hub, and the spread env with a hub among its last sources is the one an attacker launch waits for
          const uint32_t endidx = (uint32_t)__builtin_amdgcn_readlane((int)off + n, ll);   // just past this source's entries
          const int lo0 = __builtin_amdgcn_readlane(o0, ll), llast = __builtin_amdgcn_readlane(last, ll);
          uint32_t tail = 0;   // entries of the steps already taken (they follow this step's)
          for (int k1 = llast; k1 >= lo0 && endidx - tail > lo; k1 -= WAVE) {
            const int kk = k1 - (WAVE - 1) + lane;
            const bool p = kk >= lo0 && !e.blocked(kk);
            const uint64_t m = ballot(p);
            const uint32_t c = (uint32_t)__popcll(m);
            const uint32_t idx = endidx - tail - c + (uint32_t)below(m);
            if (p && idx >= lo) { e.ring[2 * (idx % CG_LOG_RING)] = (uint16_t)ls; e.ring[2 * (idx % CG_LOG_RING) + 1] = e.ocol[kk]; }
            tail += c;
          }
        }
        after = cbase;
      }
      e.log_total = (int)end;
      e.ring_dirty = true;
    }
    wsync();
    SUBSTAMP(13);
    SUBSTAMP(14);
    SUBVAL(16, n_src); SUBVAL(17, 0); SUBVAL(18, 0);
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

#endif  // CG_ATTACKER_HPP
