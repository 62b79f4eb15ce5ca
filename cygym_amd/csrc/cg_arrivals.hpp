// cg_arrivals.hpp -- Workload arrivals (volt_typhoon_env.py:575-596, CDSimulator.py:244-348).
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_ARRIVALS_HPP
#define CG_ARRIVALS_HPP

// ---------------- arrivals: CDSimulator.generate_workloads :244-348 ----------------
// random.sample(candidates, k) as "the k smallest (philox key, id)": a radix select over the candidates' 32-bit keys, one
// bit per round.  MCT = chunks of 64 devices when known at compile time (<= 4: 64 / 256 devices), 0 = run-time size.
// Every env of a batch is due in the same tick (same step_num, same period), so an arrival tick is as slow as this
// function: with the keys and candidate bits re-read from LDS in every round (384 dependent round trips per call at 256
// devices) such a launch took twice as long as any other (profiles/r04_tail_hist.txt).  Compile-time sizes keep the keys
// in registers and the candidate masks in SGPRs: a round is MCT ballots; run-time sizes keep the "still matches the
// prefix" bits in one register per lane (bit c <-> chunk c) and read the keys in staged groups of eight.
template <int MCT, class KP>
__device__ __forceinline__ void gen_workloads(Env& e, const KP& P, int num, bool server, int n_active, int step_num) {
  const int M = e.M, MC = MCT ? MCT : e.MC;
  if (n_active <= 0) return;
  if (P.c.workload_cap >= 0 && num > P.c.workload_cap) num = P.c.workload_cap;
  if (COLD(P.c.turbo)) {   // turbo throttling: cap + ramp, never zero (volt_typhoon_env.py:219-231), in the reference's own f64 steps
    const double frac = server ? P.c.turbo_fraction_servers : P.c.turbo_fraction_clients;
    int frac_cap = (int)(frac * (double)n_active);
    if (frac_cap < 1) frac_cap = 1;
    const int hard_cap = server ? P.c.turbo_max_servers : P.c.turbo_max_clients;
    double alpha = (double)step_num / (double)(P.c.turbo_ramp_steps > 1 ? P.c.turbo_ramp_steps : 1);
    alpha = alpha < 0.0 ? 0.0 : (alpha > 1.0 ? 1.0 : alpha);
    int cap = (int)rint((double)(frac_cap < hard_cap ? frac_cap : hard_cap) * alpha);   // Python round(): half to even
    if (cap < 1) cap = 1;
    if (num > cap) num = cap;
  }
  if (num > n_active) num = n_active;
  if (num <= 0) return;
  const uint32_t site = server ? CG_SITE_ARR_SERVER : CG_SITE_ARR_CLIENT;
  auto is_cand = [&](int d) -> bool {
    if (d >= M) return false;
    const uint8_t f = e.flags[d];
    return !(f & CG_F_NYA) && e.wl[d] == 0 && e.busy[d] == 0 && (((e.dst[d] & CG_D_SERVER) != 0) == server);
  };
  auto give = [&](int d) {   // the sampled device receives a job: ceil(triangular(0, mode, high)) ticks (CDSimulator.py:308)
    e.wl[d] = (uint8_t)(1 + cdf_lookup(e.draw(CG_SITE_ARR_TIME, d, 0), P.c.tri_thr, CG_TRI_TABLE));
    e.flags[d] &= (uint8_t)~CG_F_WLADV;
  };
  if constexpr (MCT > 0) {
    uint64_t cm[MCT];   // candidate masks: uniform (SGPR pairs)
    int n = 0;
#pragma unroll
    for (int c = 0; c < MCT; ++c) { cm[c] = ballot(is_cand(c * WAVE + e.lane)); n += __popcll(cm[c]); }
    if (n == 0) return;
    const int k = num < n ? num : n;
    if (k == n) {
#pragma unroll
      for (int c = 0; c < MCT; ++c) if ((cm[c] >> e.lane) & 1ull) give(c * WAVE + e.lane);
      wsync();
      return;
    }
    uint32_t key[MCT];
    bool alive[MCT];   // candidate whose key matches the prefix found so far
#pragma unroll
    for (int c = 0; c < MCT; ++c) {
      alive[c] = (cm[c] >> e.lane) & 1ull;
      key[c] = alive[c] ? e.draw(site, c * WAVE + e.lane, 0) : 0u;
    }
    // One bit per round, most significant first.  `alive`: candidates whose key matches the k-th smallest key on the bits
    // seen so far; `sure`: candidates already known to be smaller than it (selected whatever comes).  As soon as exactly
    // as many candidates are alive as are still to be selected they are all in, and the rounds stop -- with random 32-bit
    // keys after about log2(n) + 2 of the 32 rounds.
    bool sure[MCT];
#pragma unroll
    for (int c = 0; c < MCT; ++c) sure[c] = false;
    int kk = k, n_alive = n;   // kk: 1-based rank still to locate among the alive candidates
#pragma nounroll
    for (int bit = 31; bit >= 0 && n_alive > kk; --bit) {
      int cnt0 = 0;
#pragma unroll
      for (int c = 0; c < MCT; ++c) cnt0 += __popcll(ballot(alive[c] && !((key[c] >> bit) & 1u)));
      const bool one = kk > cnt0;   // the k-th smallest has this bit set: the alive keys with a 0 here are smaller
      if (one) { kk -= cnt0; n_alive -= cnt0; } else n_alive = cnt0;
#pragma unroll
      for (int c = 0; c < MCT; ++c) {
        const bool b1 = ((key[c] >> bit) & 1u) != 0;
        sure[c] = sure[c] || (alive[c] && one && !b1);
        alive[c] = alive[c] && b1 == one;
      }
    }
    // select: the sure ones, plus the first kk alive candidates in id order (all of them unless keys collide)
    int seen = 0;
#pragma unroll
    for (int c = 0; c < MCT; ++c) {
      const uint64_t em = ballot(alive[c]);
      const bool take = sure[c] || (alive[c] && (seen + below(em)) < kk);
      seen += __popcll(em);
      if (take) give(c * WAVE + e.lane);
    }
    wsync();
    return;
  } else {
  uint32_t* key = e.scr;                     // [Mp]
  int n = 0;
  uint32_t mine = 0;                         // bit c: device c * 64 + lane is a candidate (MC <= 32)
  for (int c = 0; c < MC; ++c) {
    const bool cand = is_cand(c * WAVE + e.lane);
    mine |= (cand ? 1u : 0u) << c;
    n += __popcll(ballot(cand));
  }
  if (n == 0) return;
  int k = num < n ? num : n;
  const bool all = (k == n);
  uint32_t alive = mine, sure = 0;   // bit c <-> chunk c: see the compile-time path
  int kk = k, n_alive = n;
  if (!all) {
    for (int c = 0; c < MC; ++c)
      if ((mine >> c) & 1u) key[c * WAVE + e.lane] = e.draw(site, c * WAVE + e.lane, 0);
    wsync();
    // the keys of a round are read in groups of eight before the first ballot (one LDS round trip per group, not per chunk)
#pragma nounroll
    for (int bit = 31; bit >= 0 && n_alive > kk; --bit) {
      int cnt0 = 0;
      uint32_t zero = 0;   // bit c: my key of chunk c has a 0 here
#pragma nounroll
      for (int c0 = 0; c0 < MC; c0 += 8) {
        uint32_t kj[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const int c = c0 + j < MC ? c0 + j : MC - 1; kj[j] = key[c * WAVE + e.lane]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) PIN(kj[j]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int c = c0 + j;
          const bool z = c < MC && !((kj[j] >> bit) & 1u);
          zero |= (z ? 1u : 0u) << (c & 31);
          cnt0 += __popcll(ballot(z && ((alive >> (c & 31)) & 1u)));
        }
      }
      const bool one = kk > cnt0;
      if (one) { kk -= cnt0; n_alive -= cnt0; sure |= alive & zero; } else n_alive = cnt0;
      alive &= one ? ~zero : zero;
    }
  }
  // select: the sure ones, plus the first kk alive candidates in id order (all of them unless keys collide)
  int seen = 0;
  for (int c = 0; c < MC; ++c) {
    const bool al = (alive >> c) & 1u;
    const uint64_t em = ballot(al);
    const bool take = ((sure >> c) & 1u) || (al && (seen + below(em)) < kk);
    seen += __popcll(em);
    if (take) give(c * WAVE + e.lane);
  }
  wsync();
  }
}

// volt_typhoon_env.py:575-596; the three counts come from the fused pass of the tick
template <int MCT, class KP>
__device__ __forceinline__ void arrivals(Env& e, const KP& P, int step_num, int n_active, int idle, int free_s) {
  int free_c = idle - free_s;
  int n1 = n_active > 1 ? n_active : 1;
  const int half = half_isqrt(n1);   // int(0.5*sqrt(n)) (:141-145)
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
  gen_workloads<MCT>(e, P, nC, false, n_active, step_num);
  gen_workloads<MCT>(e, P, nS, true, n_active, step_num);
}

#endif  // CG_ARRIVALS_HPP
