// cg_arrivals.hpp -- Workload arrivals (volt_typhoon_env.py:575-596, CDSimulator.py:244-348).
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_ARRIVALS_HPP
#define CG_ARRIVALS_HPP

// ---------------- arrivals: CDSimulator.generate_workloads :244-348 ----------------
template <class KP>
__device__ __forceinline__ void gen_workloads(Env& e, const KP& P, int num, bool server, int n_active, int step_num) {
  const int M = e.M, MC = e.MC;
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
      e.wl[d] = (uint8_t)(1 + cdf_lookup(e.draw(CG_SITE_ARR_TIME, d, 0), P.c.tri_thr, CG_TRI_TABLE));
      e.flags[d] &= (uint8_t)~CG_F_WLADV;
    }
  }
  wsync();
}

// volt_typhoon_env.py:575-596; the three counts come from the fused pass of the tick
template <class KP>
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
  gen_workloads(e, P, nC, false, n_active, step_num);
  gen_workloads(e, P, nS, true, n_active, step_num);
}

#endif  // CG_ARRIVALS_HPP
