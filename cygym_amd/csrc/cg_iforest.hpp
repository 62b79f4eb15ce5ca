// cg_iforest.hpp -- host side of Detector.train (CDSimulator.py:681-695): the reference's estimator,
//   IsolationForest(n_estimators=2, max_samples=256, n_jobs=1).fit(X),   X = [[from_device, to_device], ...]
// restated natively so that a batch of trainings is a multi-threaded C++ call instead of one scikit-learn fit (about
// 4 ms of Python) per env.  Not part of the reference repository: scikit-learn is its third-party dependency (the
// reference pins no version; this image holds 1.7.2).  What is restated is the published algorithm of
//   sklearn/ensemble/_iforest.py (fit), _bagging.py (_fit, _parallel_build_estimators, _generate_bagging_indices),
//   _base.py (_set_random_states), utils/_random.pyx (sample_without_replacement), tree/_classes.py (ExtraTreeRegressor),
//   tree/_tree.pyx (DepthFirstTreeBuilder), tree/_splitter.pyx (random splitter), tree/_partitioner.pyx (dense),
//   tree/_criterion.pyx (MSE) and tree/_utils.pyx / utils/_random.pxd (rand_int, rand_uniform, xorshift rand_r),
// together with the numpy legacy generator they draw from (MT19937 seeded from an int, 53-bit doubles, masked
// rejection for bounded integers, Fisher-Yates permutation) -- in the order those calls consume random numbers, so that
// with the same 32-bit seed the same forest comes out, node for node.  Pinned by tests/test_detector_cpu.py against
// scikit-learn itself (thousands of random training sets) and against the forests the reference fitted in the golden
// fixtures; scikit-learn stays the checker and the fallback (cygym_amd/detector.py) when its version differs.
//
// Exactness: sample selection, feature draws, thresholds, partitions, node order, depths and counts are integer /
// float32 / float64 operations restated one for one.  Two tests of the tree builder compare a float64 variance with
// machine epsilon (`impurity <= EPSILON`, `improvement + EPSILON < 0`); they are evaluated with the criterion's own
// formulas in the criterion's own summation order.
#ifndef CG_IFOREST_HPP
#define CG_IFOREST_HPP
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <thread>
#include <vector>
#include "cygym_spec.h"

namespace cg_iforest {

// ---- numpy legacy RandomState on MT19937 ----
struct MT {
  uint32_t key[624];
  int pos;
  void seed(uint32_t s) {   // mt19937_seed (init_genrand)
    for (int i = 0; i < 624; ++i) {
      key[i] = s;
      s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)i + 1u;
    }
    pos = 624;
  }
  void gen() {
    const uint32_t N = 624, Mm = 397;
    int kk;
    uint32_t y;
    for (kk = 0; kk < (int)(N - Mm); ++kk) {
      y = (key[kk] & 0x80000000u) | (key[kk + 1] & 0x7fffffffu);
      key[kk] = key[kk + Mm] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; kk < (int)N - 1; ++kk) {
      y = (key[kk] & 0x80000000u) | (key[kk + 1] & 0x7fffffffu);
      key[kk] = key[kk + (int)Mm - (int)N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (key[N - 1] & 0x80000000u) | (key[0] & 0x7fffffffu);
    key[N - 1] = key[Mm - 1] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    pos = 0;
  }
  uint32_t u32() {
    if (pos == 624) gen();
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  double dbl() {   // mt19937_next_double
    const int32_t a = (int32_t)(u32() >> 5), b = (int32_t)(u32() >> 6);
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
  // a value in [0, rng] by masked rejection on 32-bit draws (random_bounded_uint64_fill with use_masked, rng < 2^32 - 1;
  // also random_interval of the shuffle)
  uint32_t bounded(uint32_t rng) {
    if (rng == 0) return 0;
    uint32_t mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = u32() & mask; } while (v > rng);
    return v;
  }
  uint32_t randint(uint32_t high) { return bounded(high - 1u); }   // RandomState.randint(high): [0, high)
};

// ---- sklearn's rand_r replacement (utils/_random.pxd) ----
static inline uint32_t our_rand_r(uint32_t* seed) {
  if (*seed == 0) *seed = 1;
  *seed ^= (uint32_t)(*seed << 13);
  *seed ^= (uint32_t)(*seed >> 17);
  *seed ^= (uint32_t)(*seed << 5);
  return *seed % 2147483648u;
}
static inline long rand_int(long low, long high, uint32_t* st) { return low + (long)(our_rand_r(st) % (uint32_t)(high - low)); }
static inline double rand_uniform(double low, double high, uint32_t* st) {
  return ((high - low) * (double)our_rand_r(st) / 2147483647.0) + low;
}

struct Node { int left, right, feature, depth, n; double thr; };

// One ExtraTreeRegressor(max_features=1, splitter="random", max_depth) on the selected rows (ascending row order, like
// the zero / one sample weights of _parallel_build_estimators make it) with targets y.
struct TreeFit {
  const float* X;        // [n][2]
  const double* y;       // [n]
  std::vector<long> samples;
  std::vector<float> fv;
  long features[2], constant_features[2];
  uint32_t rr;
  double weighted_n_samples;
  std::vector<Node> nodes;

  struct Rec { long start, end; int depth, parent; bool is_left; double impurity; long n_const; };

  void build(int max_depth) {
    const double EPS = 2.220446049250313e-16;
    features[0] = 0; features[1] = 1;
    constant_features[0] = constant_features[1] = 0;
    fv.assign(samples.size() ? samples.size() : 1, 0.f);
    std::vector<Rec> stack;
    stack.push_back({0, (long)samples.size(), 0, -1, false, INFINITY, 0});
    bool first = true;
    while (!stack.empty()) {
      const Rec r = stack.back();
      stack.pop_back();
      const long start = r.start, end = r.end, n_node = end - start;
      // criterion.init: sums over samples[start:end] in their current order (weights are 1)
      double sum_total = 0.0, sq_sum_total = 0.0, wn = 0.0;
      for (long p = start; p < end; ++p) {
        const double yi = y[samples[p]];
        sum_total += yi;
        sq_sum_total += yi * yi;
        wn += 1.0;
      }
      bool is_leaf = r.depth >= max_depth || n_node < 2;
      double impurity = r.impurity;
      if (first) {
        impurity = sq_sum_total / wn;
        impurity -= (sum_total / wn) * (sum_total / wn);
        first = false;
      }
      is_leaf = is_leaf || impurity <= EPS;
      long n_const = r.n_const;
      long pos = end;
      int s_feature = -2;
      double s_thr = -2.0, imp_l = 0.0, imp_r = 0.0, improvement = 0.0;
      if (!is_leaf) {
        // node_split_random with max_features = 1, min_samples_leaf = 1, min_weight_leaf = 0, no missing values
        long f_i = 2, f_j, n_found = 0, n_drawn = 0, n_known = n_const, n_total = n_known, n_visited = 0;
        long best_pos = end;
        int best_feature = -2, cur_feature = -2;
        double best_thr = 0.0;
        bool have = false;
        while (f_i > n_total && (n_visited < 1 || n_visited <= n_found + n_drawn)) {
          ++n_visited;
          f_j = rand_int(n_drawn, f_i - n_found, &rr);
          if (f_j < n_known) {
            const long t = features[n_drawn]; features[n_drawn] = features[f_j]; features[f_j] = t;
            ++n_drawn;
            continue;
          }
          f_j += n_found;
          cur_feature = (int)features[f_j];
          float mn = X[2 * samples[start] + cur_feature], mx = mn;
          fv[start] = mn;
          for (long p = start + 1; p < end; ++p) {
            const float v = X[2 * samples[p] + cur_feature];
            fv[p] = v;
            if (v < mn) mn = v; else if (v > mx) mx = v;
          }
          if (mx <= (float)(mn + 1e-7f)) {   // constant on this node (FEATURE_THRESHOLD)
            features[f_j] = features[n_total];
            features[n_total] = cur_feature;
            ++n_found; ++n_total;
            continue;
          }
          --f_i;
          { const long t = features[f_i]; features[f_i] = features[f_j]; features[f_j] = t; }
          double thr = rand_uniform((double)mn, (double)mx, &rr);
          if (thr == (double)mx) thr = (double)mn;
          long p = start, pe = end;
          while (p < pe) {
            if ((double)fv[p] <= thr) ++p;
            else {
              --pe;
              const float tf = fv[p]; fv[p] = fv[pe]; fv[pe] = tf;
              const long ts = samples[p]; samples[p] = samples[pe]; samples[pe] = ts;
            }
          }
          if (pe - start < 1 || end - pe < 1) continue;
          // (one candidate: its proxy improvement beats -inf)
          have = true; best_pos = pe; best_feature = cur_feature; best_thr = thr;
        }
        if (have && best_pos < end) {
          // criterion.reset(); update(best_pos); children_impurity(); impurity_improvement()
          double sum_left = 0.0, wl = 0.0;
          if ((best_pos - start) <= (end - best_pos)) {
            for (long p = start; p < best_pos; ++p) { sum_left += y[samples[p]]; wl += 1.0; }
          } else {
            sum_left = sum_total; wl = wn;
            for (long p = end - 1; p > best_pos - 1; --p) { sum_left -= y[samples[p]]; wl -= 1.0; }
          }
          const double sum_right = sum_total - sum_left, wr = wn - wl;
          double sq_left = 0.0;
          for (long p = start; p < best_pos; ++p) { const double yi = y[samples[p]]; sq_left += yi * yi; }
          const double sq_right = sq_sum_total - sq_left;
          imp_l = sq_left / wl; imp_r = sq_right / wr;
          imp_l -= (sum_left / wl) * (sum_left / wl);
          imp_r -= (sum_right / wr) * (sum_right / wr);
          improvement = (wn / weighted_n_samples) * (impurity - (wr / wn * imp_r) - (wl / wn * imp_l));
          pos = best_pos; s_feature = best_feature; s_thr = best_thr;
        }
        // constant-feature bookkeeping for siblings and children
        memcpy(&features[0], &constant_features[0], sizeof(long) * (size_t)n_known);
        memcpy(&constant_features[n_known], &features[n_known], sizeof(long) * (size_t)n_found);
        n_const = n_total;
        is_leaf = is_leaf || pos >= end || (improvement + EPS < 0.0);
      }
      const int id = (int)nodes.size();
      Node nd;
      nd.left = nd.right = -1; nd.feature = is_leaf ? -2 : s_feature; nd.thr = is_leaf ? -2.0 : s_thr;
      nd.depth = r.depth; nd.n = (int)n_node;
      nodes.push_back(nd);
      if (r.parent >= 0) { if (r.is_left) nodes[r.parent].left = id; else nodes[r.parent].right = id; }
      if (!is_leaf) {
        stack.push_back({pos, end, r.depth + 1, id, false, imp_r, n_const});
        stack.push_back({start, pos, r.depth + 1, id, true, imp_l, n_const});
      }
    }
  }
};

// One Detector.train: `n_fits` consecutive IsolationForest.fit calls on the same rows from one numpy stream seeded with
// `seed32` (cygym_amd/detector.fit_forest); the last forest, flattened (cygym_spec.h), into `out` [CG_FOREST_WORDS].
// sstar[m] = decision threshold S* for max_samples_ == m (host table, detector.score_threshold).  Returns 0, or -1 when a
// tree does not fit the flat layout (more than 511 nodes, ids >= 4096): the caller then falls back to scikit-learn.
static int fit_one(const uint16_t* rows, long n, uint32_t seed32, int n_fits, const double* sstar, uint32_t* out) {
  if (n <= 0) return -1;
  std::vector<float> X((size_t)n * 2);
  for (long i = 0; i < 2 * n; ++i) X[(size_t)i] = (float)rows[i];
  MT rnd;
  rnd.seed(seed32);
  std::vector<double> y((size_t)n);
  const long ms = n < 256 ? n : 256;
  int max_depth = 0;
  { long v = ms < 2 ? 2 : ms; while ((1L << max_depth) < v) ++max_depth; }   // ceil(log2(max(max_samples, 2)))
  TreeFit tf[CG_FOREST_TREES];
  for (int f = 0; f < (n_fits < 1 ? 1 : n_fits); ++f) {
    for (long i = 0; i < n; ++i) y[(size_t)i] = rnd.dbl();                    // y = rnd.uniform(size=n)
    uint32_t seeds[CG_FOREST_TREES];
    for (int t = 0; t < CG_FOREST_TREES; ++t) seeds[t] = rnd.randint(2147483647u);   // random_state.randint(MAX_INT, size=2)
    for (int t = 0; t < CG_FOREST_TREES; ++t) {
      MT r1; r1.seed(seeds[t]);
      const uint32_t tree_seed = r1.randint(2147483647u);                    // _set_random_states
      MT r2; r2.seed(seeds[t]);                                              // _generate_bagging_indices
      // (features: sample_without_replacement(2, 2) -> reservoir with nothing to replace: no draw)
      std::vector<char> sel((size_t)n, 0);
      const double ratio = (double)ms / (double)n;
      if (ratio > 0.01 && ratio < 0.99) {                                    // rng.permutation(n)[:ms]
        std::vector<long> perm((size_t)n);
        for (long i = 0; i < n; ++i) perm[(size_t)i] = i;
        for (long i = n - 1; i >= 1; --i) {
          const long j = (long)r2.bounded((uint32_t)i);
          const long tmp = perm[(size_t)i]; perm[(size_t)i] = perm[(size_t)j]; perm[(size_t)j] = tmp;
        }
        for (long i = 0; i < ms; ++i) sel[(size_t)perm[(size_t)i]] = 1;
      } else if (ratio < 0.2) {                                              // tracking selection (not reachable: ms = min(256, n))
        long got = 0;
        while (got < ms) { const long j = (long)r2.randint((uint32_t)n); if (!sel[(size_t)j]) { sel[(size_t)j] = 1; ++got; } }
      } else {                                                               // reservoir sampling
        std::vector<long> res((size_t)ms);
        for (long i = 0; i < ms; ++i) res[(size_t)i] = i;
        for (long i = ms; i < n; ++i) { const long j = (long)r2.randint((uint32_t)(i + 1)); if (j < ms) res[(size_t)j] = i; }
        for (long i = 0; i < ms; ++i) sel[(size_t)res[(size_t)i]] = 1;
      }
      TreeFit& T = tf[t];
      T.X = X.data(); T.y = y.data();
      T.samples.clear(); T.nodes.clear();
      for (long i = 0; i < n; ++i) if (sel[(size_t)i]) T.samples.push_back(i);
      T.weighted_n_samples = (double)ms;
      MT r3; r3.seed(tree_seed);
      T.rr = r3.randint(2147483647u);                                        // splitter.rand_r_state = random_state.randint(0, RAND_R_MAX)
      T.build(max_depth);
    }
  }
  memset(out, 0, sizeof(uint32_t) * CG_FOREST_WORDS);
  memcpy(out, &sstar[ms], 8);
  uint32_t counts[CG_FOREST_TREES];
  for (int t = 0; t < CG_FOREST_TREES; ++t) {
    const std::vector<Node>& nd = tf[t].nodes;
    if (nd.size() > CG_FOREST_NODES - 1) return -1;
    uint32_t* w = out + CG_FOREST_HDR + t * CG_FOREST_NODES;
    for (size_t i = 0; i < nd.size(); ++i) {
      if (nd[i].left < 0) {
        const int depth = nd[i].depth + 1;                                   // compute_node_depths: root = 1
        if (nd[i].n < 0 || nd[i].n >= 512 || depth < 1 || depth > 15) return -1;
        w[i] = (1u << 31) | ((uint32_t)depth << 9) | (uint32_t)nd[i].n;
      } else {
        const double fl = floor(nd[i].thr);
        if (nd[i].feature < 0 || nd[i].feature > 1 || fl < 0.0 || fl >= 4096.0) return -1;
        w[i] = ((uint32_t)nd[i].feature << 30) | ((uint32_t)fl << 18) | ((uint32_t)nd[i].left << 9) | (uint32_t)nd[i].right;
      }
    }
    counts[t] = (uint32_t)nd.size();
  }
  out[2] = counts[0] | (counts[1] << 16);
  out[7] = (uint32_t)ms;
  return 0;
}

}  // namespace cg_iforest
#endif  // CG_IFOREST_HPP
