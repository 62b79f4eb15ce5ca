/* cygym_oracle.c -- CPU restatement of one CyGym tick.  TEST INFRASTRUCTURE.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  It restates, sequentially and in
 * the reference's own loop order, what the reference environment does per tick,
 * over the flat struct-of-arrays layout of include/cygym_abi.h (all pointers are
 * HOST pointers here).  It deliberately shares no algorithm with the HIP kernels
 * beyond include/cygym_spec.h (bit layout + draw convention): the kernels use
 * parallel reformulations (fix-point frontier expansion, ballot ranking, radix
 * selection) which this scalar code checks.
 *
 * Pinning: the reference's own tests hold no golden vectors for this path
 * (SURVEY.md section 4), so this oracle is pinned against OUTPUTS OF THE REFERENCE
 * ITSELF, run in the build container with its RNG call sites fed from the same
 * Philox stream (oracle/harness/ref_harness.py -> tests/golden/*.npz;
 * tests/test_oracle_golden.py).  Detector "trained" mode: Detector.train is a host
 * callback (scikit-learn's IsolationForest, the reference's own dependency; the
 * harness exports the forests the reference fitted, flattened as in cygym_spec.h),
 * Detector.batch_predict is restated below as a walk over those flat trees and is
 * pinned by the fixtures whose names end in "_trained".
 *
 * Reference (paths relative to the reference checkout):
 *   step            volt_typhoon_env.py:818-1333
 *   step_grouped    volt_typhoon_env.py:612-779
 *   arrivals        volt_typhoon_env.py:141-145, 184-293, 575-596; CDSimulator.py:244-348
 *   stash           volt_typhoon_env.py:419-453
 *   edge picks      volt_typhoon_env.py:485-511
 *   randomize       volt_typhoon_env.py:330-383
 *   observations    CyberDefenseEnv.py:146-257
 *   done            CyberDefenseEnv.py:547-552
 *   evolve_network  CyberDefenseEnv.py:583-875
 *   logger/detector CDSimulator.py:663-742
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "cygym_abi.h"

typedef struct {
  const cygym_topology* t;
  const cygym_config* c;
  int M, X, E, EW;
  uint8_t *flags, *busy, *wl, *comp_by, *st_flags, *st_busy, *st_wl, *st_comp_by;
  uint32_t* blocked;
  uint32_t* extra; /* [CG_X_WORDS(K)] edges added by evolve_network, or NULL */
  int K;
  uint16_t* ring;
  uint16_t* hist;   /* [CG_HIST_RING][2] long comm-log history, or NULL */
  uint32_t* forest; /* [CG_FOREST_WORDS] fitted isolation forest, or NULL */
  float* anomaly;   /* [M] Device.anomaly_score of this env (-1 = None), or NULL: the topology's static column */
  int32_t* ienv;
  double* fenv;
  uint32_t env_id, tick;
  uint8_t* occ; /* [3][M] per-tick occurrence counters: clean, patch, block/unblock */
  int isolate_iter;
  int scan_ord;
  int zeroday_occ;
  int baseline; /* env.base_line of the tick in progress (config, or the override in the mode word) */
} env_t;

static uint32_t drw(const env_t* e, uint32_t site, uint32_t a, uint32_t b) {
  return cg_draw(e->c->seed, e->env_id, e->tick, site, a, b);
}
static int is_blocked(const env_t* e, int slot) { return (e->blocked[slot >> 5] >> (slot & 31)) & 1u; }
static void set_blocked_pair(env_t* e, int u, int v, int val) {
  /* (u,v) in env._blocked: every duplicate (u,v) out-entry shares the state */
  for (int k = e->t->out_ptr[u]; k < e->t->out_ptr[u + 1]; ++k)
    if (e->t->out_col[k] == v) {
      if (val) e->blocked[k >> 5] |= (1u << (k & 31));
      else     e->blocked[k >> 5] &= ~(1u << (k & 31));
    }
}
/* ---- edges added by evolve_network (cygym_spec.h: extra-edge list) ---- */
static int x_n(const env_t* e) { return e->extra ? (int)CG_E_NX(e->ienv[CG_I_FLAGS]) : 0; }
static void x_set_n(env_t* e, int n) {
  e->ienv[CG_I_FLAGS] = (int32_t)(((uint32_t)e->ienv[CG_I_FLAGS] & 0xFFFFu) | ((uint32_t)n << CG_E_NX_SHIFT));
}
static int x_blocked(const env_t* e, int j) { return (e->extra[e->K + (j >> 5)] >> (j & 31)) & 1u; }
/* an entry of a merged adjacency row: the neighbour and where its blocked bit lives */
typedef struct { int other; int ref; } ment_t; /* ref >= 0: base out-slot; ref < 0: extra entry -(ref+1) */
static int ref_blocked(const env_t* e, int ref) { return ref >= 0 ? is_blocked(e, ref) : x_blocked(e, -(ref + 1)); }
/* _outnbrs[u] after _rebuild_graph_cache: base row and u's added edges, by ascending neighbour id */
static int merged_out(const env_t* e, int u, ment_t* m) {
  const int xn = x_n(e);
  int n = 0, j = 0;
  while (j < xn && (int)(e->extra[j] >> 16) < u) ++j;
  for (int k = e->t->out_ptr[u]; k < e->t->out_ptr[u + 1]; ++k) {
    int v = e->t->out_col[k];
    while (j < xn && (int)(e->extra[j] >> 16) == u && (int)(e->extra[j] & 0xFFFFu) < v) {
      m[n].other = (int)(e->extra[j] & 0xFFFFu); m[n].ref = -(j + 1); ++n; ++j;
    }
    m[n].other = v; m[n].ref = k; ++n;
  }
  while (j < xn && (int)(e->extra[j] >> 16) == u) { m[n].other = (int)(e->extra[j] & 0xFFFFu); m[n].ref = -(j + 1); ++n; ++j; }
  return n;
}
/* _innbrs[d]: base in-row and the added edges ending at d, by ascending source id */
static int merged_in(const env_t* e, int d, ment_t* m) {
  const int xn = x_n(e);
  int n = 0, j = 0;
  for (int k = e->t->in_ptr[d]; k <= e->t->in_ptr[d + 1]; ++k) {
    int u = k < e->t->in_ptr[d + 1] ? e->t->in_col[k] : 0x10000;
    for (; j < xn; ++j) {
      if ((int)(e->extra[j] & 0xFFFFu) != d) continue;
      if ((int)(e->extra[j] >> 16) >= u) break;
      m[n].other = (int)(e->extra[j] >> 16); m[n].ref = -(j + 1); ++n;
    }
    if (k < e->t->in_ptr[d + 1]) { m[n].other = u; m[n].ref = e->t->in_eid[k]; ++n; }
  }
  return n;
}
static void set_blocked_ref(env_t* e, int u, int v, int ref, int val) {
  if (ref >= 0) { set_blocked_pair(e, u, v, val); return; }
  int j = -(ref + 1);
  if (val) e->extra[e->K + (j >> 5)] |= (1u << (j & 31));
  else     e->extra[e->K + (j >> 5)] &= ~(1u << (j & 31));
}
static void log_comm(env_t* e, int from, int to) { /* CDSimulator.py:120, :667 */
  uint32_t n = (uint32_t)e->ienv[CG_I_LOG_TOTAL];
  uint16_t* r = e->ring + 2 * (n % CG_LOG_RING);
  r[0] = (uint16_t)from; r[1] = (uint16_t)to;
  if (e->hist) { uint16_t* q = e->hist + 2 * (n % CG_HIST_RING); q[0] = (uint16_t)from; q[1] = (uint16_t)to; }
  e->ienv[CG_I_LOG_TOTAL] = (int32_t)(n + 1);
}
static void set_busy(env_t* e, int d, int v) {
  if (v > 255) { v = 255; e->ienv[CG_I_FLAGS] |= CG_E_BUSY_SAT; }
  e->busy[d] = (uint8_t)v;
}
static void clear_wl(env_t* e, int d) { e->wl[d] = 0; e->flags[d] &= (uint8_t)~CG_F_WLADV; }
static int stall(env_t* e, uint32_t site, int d, int b, int lo, int hi) {
  return cg_randint(drw(e, site, (uint32_t)d, (uint32_t)b), lo, hi);
}

/* IsolationForest.predict(point) == -1 for one (from, to) log point, over the flattened forest of cygym_spec.h:
 * depths = sum over trees of (node depth of the leaf + apl[n_node_samples of the leaf] - 1.0)
 * (sklearn _parallel_compute_tree_depths), anomaly iff depths < S* (the header's threshold). */
static int forest_anomaly_s(const uint32_t* fo, const double* apl, unsigned from, unsigned to, double* sum);
static int forest_anomaly(const uint32_t* fo, const double* apl, unsigned from, unsigned to) {
  double s;
  return forest_anomaly_s(fo, apl, from, to, &s);
}
static int forest_anomaly_s(const uint32_t* fo, const double* apl, unsigned from, unsigned to, double* sum) {
  uint64_t sb = (uint64_t)fo[0] | ((uint64_t)fo[1] << 32);
  double sstar, depths = 0.0;
  memcpy(&sstar, &sb, 8);
  for (int t = 0; t < CG_FOREST_TREES; ++t) {
    const uint32_t* tr = fo + CG_FOREST_HDR + t * CG_FOREST_NODES;
    uint32_t w = tr[0];
    for (int it = 0; it < 16 && !CG_FN_LEAF(w); ++it) {
      unsigned x = CG_FN_FEAT(w) ? to : from;
      w = tr[x <= CG_FN_THR(w) ? CG_FN_LEFT(w) : CG_FN_RIGHT(w)];
    }
    unsigned n = CG_FN_NSAMP(w);
    if (n >= CG_DET_APL_N) n = CG_DET_APL_N - 1;
    depths += ((double)CG_FN_DEPTH(w) + apl[n]) - 1.0;
  }
  *sum = depths;
  return depths < sstar;
}

/* ---- defender global actions, shared by step (:918-976) and _step_apply_only (:627-668) */
static void def_global(env_t* e, int at, const int16_t* dev, int L, double* cost, int* dirty, int grouped) {
  const double ds = e->c->def_scale;
  const int M = e->M;
  if (at == 2) {
    e->ienv[CG_I_CKPT_CNT] += 1;
    e->ienv[CG_I_FLAGS] |= CG_E_HAS_CKPT;
    *cost += -0.5 * L * ds;
    e->fenv[CG_D_DEF_COST] += 0.5 * L * ds;
    for (int d = 0; d < M; ++d) if (e->busy[d] > 0) set_busy(e, d, e->busy[d] + 1);
  } else if (at == 3) {
    e->ienv[CG_I_REVERT_CNT] += 1;
    if (e->ienv[CG_I_FLAGS] & CG_E_HAS_CKPT) {
      for (int d = 0; d < M; ++d) {
        set_busy(e, d, stall(e, CG_SITE_STALL_REVERT, d, 0, 0, e->c->default_high));
        clear_wl(e, d);
      }
      *cost += -1.0 * L * ds;
      *dirty = 1;
    }
  } else if (at == 10) {
    if (!grouped) {
      if (L > 0) {
        int d = dev[0];
        if (d >= 0 && d < M) set_busy(e, d, e->busy[d] + 1);
      } else {
        for (int d = 0; d < M; ++d) if (e->busy[d] > 0) set_busy(e, d, e->busy[d] + 1);
      }
    }
    *cost += -1.0 * ds;
    if (e->ienv[CG_I_LOG_TOTAL] > 0) { /* Detector.train(non-empty) CDSimulator.py:692-695: the fit itself is
                                          the host's job (cygym_spec.h); the tick records the request */
      e->ienv[CG_I_FLAGS] |= CG_E_DET_TRAIN | CG_E_DET_PENDING;
      e->ienv[CG_I_FLAGS] &= ~CG_E_DET_RANDOM;
      if (e->forest) {
        e->forest[6] = (e->forest[3] == e->tick && e->forest[6] > 0) ? e->forest[6] + 1 : 1;
        e->forest[3] = e->tick; e->forest[4] = (uint32_t)e->ienv[CG_I_LOG_TOTAL];
      }
    }
  } else if (at == 11) {
    if (L > 0) {
      int d = dev[0];
      if (d >= 0 && d < M) { /* _device_state :419-428 */
        uint8_t f = e->flags[d];
        e->st_flags[d] = (uint8_t)(CG_S_VALID | (f & CG_S_KEEP));
        e->st_busy[d] = e->busy[d];
        e->st_wl[d] = e->wl[d];
        e->st_comp_by[d] = e->comp_by[d];
      }
    }
    e->ienv[CG_I_CKPT_CNT] += 1;
    *cost += -0.1 * ds;
    e->fenv[CG_D_DEF_COST] += 0.1 * ds;
  }
}

static void do_clean(env_t* e, int d, double* cost) { /* :996-1011 / :676-690 */
  const double ds = e->c->def_scale;
  if (e->flags[d] & CG_F_OWNED) return;
  int comp = e->flags[d] & CG_F_COMP;
  *cost += (comp ? 0.3 : -0.01) * ds;
  e->fenv[CG_D_CLEAN_COST] += (comp ? 0.3 : 0.01) * ds;
  e->fenv[CG_D_DEF_COST] += (comp ? 0.3 : 0.01) * ds;
  e->ienv[CG_I_DISCOVERED] |= e->comp_by[d];
  e->comp_by[d] = 0;
  e->flags[d] &= (uint8_t)~CG_F_COMP;
  int b = e->occ[d]++;
  set_busy(e, d, stall(e, CG_SITE_STALL_CLEAN, d, b, 0, e->c->default_high));
  clear_wl(e, d);
}

/* pool of incident edges with the given blocked state (:502-511); returns count, writes (u,v,ref) */
static int incident_pool(const env_t* e, int d, int want_blocked, int* pu, int* pv, int* pr, ment_t* row) {
  int n = 0;
  int len = merged_out(e, d, row);
  for (int i = 0; i < len; ++i)
    if (ref_blocked(e, row[i].ref) == want_blocked) { pu[n] = d; pv[n] = row[i].other; pr[n] = row[i].ref; ++n; }
  len = merged_in(e, d, row);
  for (int i = 0; i < len; ++i)
    if (ref_blocked(e, row[i].ref) == want_blocked) { pu[n] = row[i].other; pv[n] = d; pr[n] = row[i].ref; ++n; }
  return n;
}

static void def_per_device(env_t* e, int at, const int16_t* dev, int L, int app, double* cost, int* dirty,
                           int* pu, int* pv, int* pr, ment_t* row) {
  const double ds = e->c->def_scale;
  const int M = e->M;
  for (int p = 0; p < L; ++p) {
    int d = dev[p];
    if (d < 0 || d >= M) continue;          /* host rejects these (KeyError) */
    if (e->flags[d] & CG_F_NYA) continue;   /* :992 */
    switch (at) {
      case 1: do_clean(e, d, cost); break;
      case 4: /* :1013-1018 */
        *cost += -1.0 * ds;
        if (app >= 0 && app < e->t->napps[d]) {
          int b = e->occ[M + d]++;
          set_busy(e, d, stall(e, CG_SITE_STALL_PATCH, d, b, 0, e->c->default_high));
        }
        break;
      case 5: { /* scan :1020-1069 */
        e->ienv[CG_I_SCAN_CNT] += 1;
        int ord = e->scan_ord++;
        uint32_t total = (uint32_t)e->ienv[CG_I_LOG_TOTAL];
        if (!e->c->fast_scan) { /* per-log path :1030-1050: predict the last <= 256 entries one by one */
          int w = total < CG_SLOW_SCAN_WINDOW ? (int)total : CG_SLOW_SCAN_WINDOW;
          int fl = e->ienv[CG_I_FLAGS];
          int coin = (fl & CG_E_DET_RANDOM) != 0;
          int trained = !coin && (fl & CG_E_DET_TRAIN) && !e->c->turbo;
          if (trained && w > 0 && (!e->forest || !e->t->det_apl || (fl & CG_E_DET_PENDING) || e->forest[2] == 0)) {
            e->ienv[CG_I_FLAGS] |= CG_E_UNPINNED;
            trained = 0;
          }
          for (int j = 0; j < w; ++j) {
            uint32_t idx = total - (uint32_t)w + (uint32_t)j;
            const uint16_t* pt = e->hist + 2 * (idx % CG_HIST_RING);
            int a = 0;
            float score = -1.f; /* Detector.predict(..., return_score=True): None unless trained */
            if (e->c->turbo) score = 0.f; /* :1036-1038 */
            else if (coin) a = (cg_index(drw(e, CG_SITE_DET_COIN, (uint32_t)j, (uint32_t)ord), 2) == 0);
            else if (trained) {
              double sum;
              a = forest_anomaly_s(e->forest, e->t->det_apl, pt[0], pt[1], &sum);
              unsigned ms = e->forest[7] < CG_DET_APL_N ? e->forest[7] : CG_DET_APL_N - 1;
              double den = (double)CG_FOREST_TREES * e->t->det_apl[ms];
              score = (float)(0.5 - pow(2.0, -(den != 0.0 ? sum / den : 1.0)));
            }
            if (e->anomaly) e->anomaly[d] = score;
            *cost += -0.5 * ds;
            e->fenv[CG_D_DEF_COST] += 0.5 * ds;
            if (a && pt[0] < e->M) {
              int snd = pt[0];
              e->ienv[CG_I_DISCOVERED] |= e->comp_by[snd];
              e->comp_by[snd] = 0;
              e->flags[snd] &= (uint8_t)~CG_F_COMP;
              set_busy(e, snd, stall(e, CG_SITE_STALL_SCAN, snd, ord, 0, e->c->default_high));
            }
          }
          break;
        }
        int w = total < CG_SCAN_WINDOW ? (int)total : CG_SCAN_WINDOW;
        if (w > 0) {
          int anom[CG_SCAN_WINDOW];
          int n_anom = 0;
          int fl = e->ienv[CG_I_FLAGS];
          int trained = (fl & CG_E_DET_TRAIN) && !(fl & CG_E_DET_RANDOM) && !e->c->turbo;
          if (trained && (!e->forest || !e->t->det_apl || (fl & CG_E_DET_PENDING) || e->forest[2] == 0)) { /* [2] == 0: no trees were ever installed */
            e->ienv[CG_I_FLAGS] |= CG_E_UNPINNED; /* no current forest: all "D", flagged (cygym_spec.h) */
            trained = 0;
          }
          for (int j = 0; j < w; ++j) { /* Detector.batch_predict CDSimulator.py:714-723 */
            int a = 0;
            if (e->c->turbo) a = 0; /* predictions = [] in turbo mode (volt_typhoon_env.py:1055) */
            else if (fl & CG_E_DET_RANDOM) a = (cg_index(drw(e, CG_SITE_DET_COIN, (uint32_t)j, (uint32_t)ord), 2) == 0);
            else if (trained) { /* IsolationForest.predict == -1 (:721-723) over the flat trees */
              uint32_t idx = total - (uint32_t)w + (uint32_t)j;
              const uint16_t* pt = e->ring + 2 * (idx % CG_LOG_RING);
              a = forest_anomaly(e->forest, e->t->det_apl, pt[0], pt[1]);
            }
            /* untrained -> all "D" (:718-719) */
            anom[j] = a; n_anom += a;
          }
          int majority = w / 2 + 1;
          *cost += -0.5 * ds;
          e->fenv[CG_D_DEF_COST] += 0.5 * ds;
          if (n_anom >= majority) {
            for (int j = 0; j < w; ++j) if (anom[j]) {
              uint32_t idx = total - (uint32_t)w + (uint32_t)j;
              int snd = e->ring[2 * (idx % CG_LOG_RING)];
              if (snd >= e->M) continue; /* a ring loaded with foreign ids: the reference would raise KeyError */
              e->flags[snd] &= (uint8_t)~CG_F_COMP;
              set_busy(e, snd, stall(e, CG_SITE_STALL_SCAN, snd, ord, 0, e->c->default_high));
            }
          }
        }
        break;
      }
      case 6: { /* :1071-1080 */
        *cost += -0.5 * ds;
        e->fenv[CG_D_DEF_COST] += 0.5 * ds;
        int n = incident_pool(e, d, 0, pu, pv, pr, row);
        if (n > 0) {
          int b = e->occ[2 * M + d]++;
          int r = (int)cg_index(drw(e, CG_SITE_PICK_BLOCK, (uint32_t)d, (uint32_t)b), (uint32_t)n);
          set_blocked_ref(e, pu[r], pv[r], pr[r], 1);
          e->ienv[CG_I_EDGES_BLOCKED] += 1;
          *dirty = 1;
        }
        break;
      }
      case 7: /* :1082-1089 */
        *cost += -0.5 * ds;
        e->flags[d] |= CG_F_NYA;
        e->flags[d] &= (uint8_t)~CG_F_COMP;
        e->comp_by[d] = 0;
        clear_wl(e, d);
        *dirty = 1;
        break;
      case 9: { /* :1091-1100 */
        *cost += -0.5 * ds;
        e->fenv[CG_D_DEF_COST] += 0.5 * ds;
        int n = incident_pool(e, d, 1, pu, pv, pr, row);
        if (n > 0) {
          int b = e->occ[2 * M + d]++;
          int r = (int)cg_index(drw(e, CG_SITE_PICK_UNBLOCK, (uint32_t)d, (uint32_t)b), (uint32_t)n);
          set_blocked_ref(e, pu[r], pv[r], pr[r], 0);
          e->ienv[CG_I_EDGES_ADDED] += 1;
          *dirty = 1;
        }
        break;
      }
      case 12: { /* :1102-1109, _apply_device_state :430-437 */
        int d0 = dev[0];
        if (d0 >= 0 && d0 < M && (e->st_flags[d0] & CG_S_VALID)) {
          uint8_t sf = e->st_flags[d0];
          e->flags[d0] = (uint8_t)((e->flags[d0] & ~CG_S_KEEP) | (sf & CG_S_KEEP));
          e->busy[d0] = e->st_busy[d0];
          e->wl[d0] = e->st_wl[d0];
          e->comp_by[d0] = e->st_comp_by[d0];
          *cost += -1.0 * ds;
          e->fenv[CG_D_DEF_COST] += 1.0 * ds;
        }
        break;
      }
      case 13: { /* :1111-1123 */
        int d0 = dev[0];
        if (d0 >= 0 && d0 < M) {
          e->flags[d0] &= (uint8_t)~CG_F_COMP;
          e->comp_by[d0] = 0;
          clear_wl(e, d0);
          int b = e->isolate_iter++;
          set_busy(e, d0, stall(e, CG_SITE_STALL_ISOLATE, d0, b, 3, e->c->default_high + 3));
        }
        *cost += -3.0 * ds;
        e->fenv[CG_D_CLEAN_COST] += 3.0 * ds;
        e->fenv[CG_D_DEF_COST] += 3.0 * ds;
        break;
      }
      default: break;
    }
  }
}

/* ---- attacker (:1126-1202) ---- */
static void attacker(env_t* e, int at, const int32_t* expl, int n_expl, double* cost, int* srcs, ment_t* row) {
  const int M = e->M;
  int n_src = 0;
  for (int d = 0; d < M; ++d)
    if (e->flags[d] & (CG_F_COMP | CG_F_OWNED)) srcs[n_src++] = d; /* :1127 snapshot */
  if (e->baseline == 3) return; /* "No Attack" */
  if (at == 1) {
    for (int j = 0; j < n_expl; ++j) {
      int raw = expl[j];
      if (e->c->zero_day) {
        uint32_t mask = (uint32_t)e->c->zero_day_owned_mask;
        int in = (raw >= 0 && raw < 32 && ((mask >> raw) & 1u));
        if (!in) { /* random.choice(owned_indices) :1136 (sorted canonical order) */
          int cnt = __builtin_popcount(mask);
          if (cnt == 0) continue;
          int r = (int)cg_index(drw(e, CG_SITE_ZERODAY, (uint32_t)e->zeroday_occ++, 0), (uint32_t)cnt);
          for (int i = 0; i < 32; ++i) if ((mask >> i) & 1u) { if (r-- == 0) { raw = i; break; } }
        }
      }
      if (raw < 0 || raw >= e->X) continue; /* :1138-1144 (id lookup never matches an int) */
      uint8_t ebit = (uint8_t)(1u << raw);
      for (int si = 0; si < n_src; ++si) {
        int s = srcs[si];
        int s_dc = e->t->dstatic[s] & CG_D_DC;
        const int len = merged_out(e, s, row);
        for (int k = 0; k < len; ++k) {
          if (ref_blocked(e, row[k].ref)) continue;
          int v = row[k].other;
          log_comm(e, s, v);
          if (s_dc) { e->flags[v] |= CG_F_COMP; e->comp_by[v] |= ebit; break; }
          if (e->flags[v] & CG_F_REACH) { e->flags[v] |= CG_F_COMP; break; }
          if (!(e->flags[v] & CG_F_COMP) && (e->flags[v] & CG_F_KNOWN)) {
            if (e->t->vuln[v] & ebit) { e->flags[v] |= CG_F_COMP; break; }
          }
        }
      }
    }
  } else if (at == 2) {
    if (n_src > 0) {
      int s = srcs[cg_index(drw(e, CG_SITE_PROBE_SRC, 0, 0), (uint32_t)n_src)];
      const int len = merged_out(e, s, row);
      for (int k = 0; k < len; ++k) {
        if (ref_blocked(e, row[k].ref)) continue;
        int v = row[k].other;
        if (!(e->flags[v] & CG_F_KNOWN)) { e->flags[v] |= CG_F_KNOWN; *cost += 0.1; break; }
      }
    }
  }
}

/* ---- workload advance (:1242-1261 / :705-725) ---- */
static int advance_work(env_t* e) {
  int current = 0;
  for (int d = 0; d < e->M; ++d) {
    if (e->busy[d] != 0 || (e->flags[d] & CG_F_NYA)) continue;
    if (e->wl[d] > 0 && !(e->flags[d] & CG_F_WLADV)) {
      if (--e->wl[d] == 0) { e->ienv[CG_I_WORK_DONE] += 1; current += 1; }
    }
    if (e->wl[d] > 0 && (e->flags[d] & CG_F_WLADV)) {
      if (--e->wl[d] == 0) e->flags[d] &= (uint8_t)~CG_F_WLADV;
    }
  }
  return current;
}

typedef struct { uint32_t key; int id; } kid_t;
static int kid_cmp(const void* a, const void* b) {
  const kid_t* x = (const kid_t*)a; const kid_t* y = (const kid_t*)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->id - y->id;
}

/* CDSimulator.generate_workloads :244-348 behind _generate_workloads_timed :193-245 */
static void gen_workloads(env_t* e, int num, int server, kid_t* tmp) {
  const int M = e->M;
  int n_active = 0;
  for (int d = 0; d < M; ++d) n_active += !(e->flags[d] & CG_F_NYA);
  if (n_active <= 0) return;
  if (e->c->workload_cap >= 0 && num > e->c->workload_cap) num = e->c->workload_cap;
  if (e->c->turbo) { /* turbo throttling volt_typhoon_env.py:219-231 (bootstrap is never set on the step path) */
    double frac = server ? e->c->turbo_fraction_servers : e->c->turbo_fraction_clients;
    int frac_cap = (int)(frac * (double)n_active);
    if (frac_cap < 1) frac_cap = 1;
    int hard_cap = server ? e->c->turbo_max_servers : e->c->turbo_max_clients;
    double alpha = (double)e->ienv[CG_I_STEP_NUM] / (double)(e->c->turbo_ramp_steps > 1 ? e->c->turbo_ramp_steps : 1);
    alpha = alpha < 0.0 ? 0.0 : (alpha > 1.0 ? 1.0 : alpha);
    int cap = (int)rint((double)(frac_cap < hard_cap ? frac_cap : hard_cap) * alpha); /* Python round(): half to even */
    if (cap < 1) cap = 1;
    if (num > cap) num = cap;
  }
  if (num > n_active) num = n_active;
  if (num <= 0) return;
  uint32_t site = server ? CG_SITE_ARR_SERVER : CG_SITE_ARR_CLIENT;
  int n = 0;
  for (int d = 0; d < M; ++d) {
    if (e->flags[d] & CG_F_NYA) continue;
    if (e->wl[d] != 0) continue;
    if (e->busy[d] > 0) continue;
    int is_srv = (e->t->dstatic[d] & CG_D_SERVER) != 0;
    if (is_srv != server) continue;
    tmp[n].key = drw(e, site, (uint32_t)d, 0);
    tmp[n].id = d;
    ++n;
  }
  if (n == 0) return;
  int k = num < n ? num : n;
  qsort(tmp, (size_t)n, sizeof(kid_t), kid_cmp);
  for (int i = 0; i < k; ++i) {
    int d = tmp[i].id;
    e->wl[d] = (uint8_t)(1 + cg_cdf_lookup(drw(e, CG_SITE_ARR_TIME, (uint32_t)d, 0), e->c->tri_thr, CG_TRI_TABLE));
    e->flags[d] &= (uint8_t)~CG_F_WLADV;
  }
}

static int round_half_even_ratio(int num, int den) { /* round(num/den) for num,den > 0 */
  int q = num / den, r = num % den;
  if (2 * r > den) return q + 1;
  if (2 * r < den) return q;
  return (q & 1) ? q + 1 : q;
}

static void arrivals(env_t* e, kid_t* tmp) { /* :575-596 */
  const int M = e->M;
  int n_active = 0, idle = 0, free_c = 0, free_s = 0;
  for (int d = 0; d < M; ++d) {
    if (e->flags[d] & CG_F_NYA) continue;
    ++n_active;
    if (e->busy[d] == 0 && e->wl[d] == 0) {
      ++idle;
      if (e->t->dstatic[d] & CG_D_SERVER) ++free_s; else ++free_c;
    }
  }
  /* _arrival_period :141-145: min(max, max(10, int(base + 0.5*sqrt(max(1,n))))) */
  int n1 = n_active > 1 ? n_active : 1;
  int half = 0;
  while (4 * (half + 1) * (half + 1) <= n1) ++half;
  int period = e->c->workload_period_base + half;
  if (period < 10) period = 10;
  if (period > e->c->workload_period_max) period = e->c->workload_period_max;
  if (e->ienv[CG_I_STEP_NUM] % period != 0) return;
  if (n_active == 0 || 10 * idle < n_active) return; /* _idle_fraction() < 0.10 */
  int nC, nS;
  if (e->c->scaling_vulnerability) { /* _scaled_numloads(100, 10) :266-293, anchor 50 */
    int req_c = round_half_even_ratio(100 * n_active, 50);
    int req_s = round_half_even_ratio(10 * n_active, 50);
    if (req_c < 1) req_c = 1;
    if (req_s < 1) req_s = 1;
    int cap_c = free_c > 1 ? free_c : 1, cap_s = free_s > 1 ? free_s : 1;
    nC = req_c < cap_c ? req_c : cap_c;
    nS = req_s < cap_s ? req_s : cap_s;
  } else { nC = 100; nS = 10; }
  if (e->c->workload_cap > 0) {
    int total = nC + nS;
    if (total > e->c->workload_cap) {
      double ratio = (double)e->c->workload_cap / (double)total;
      nC = (int)(nC * ratio); if (nC < 0) nC = 0;
      nS = (int)(nS * ratio); if (nS < 0) nS = 0;
    }
  }
  gen_workloads(e, nC, 0, tmp);
  gen_workloads(e, nS, 1, tmp);
}

static int rank_select(const env_t* e, uint8_t mask, uint8_t want, int r) {
  for (int d = 0; d < e->M; ++d)
    if ((e->flags[d] & mask) == want) { if (r-- == 0) return d; }
  return -1;
}
static int has_edge(const env_t* e, int u, int v) { /* g.get_eid(u, v, directed=True, error=False) != -1 */
  for (int k = e->t->out_ptr[u]; k < e->t->out_ptr[u + 1]; ++k) if (e->t->out_col[k] == v) return 1;
  const uint32_t key = ((uint32_t)u << 16) | (uint32_t)v;
  for (int j = 0; j < x_n(e); ++j) if (e->extra[j] == key) return 1;
  return 0;
}
static int degree_all(const env_t* e, int d) { /* g.degree(d): in + out, a loop counts twice */
  int deg = (e->t->out_ptr[d + 1] - e->t->out_ptr[d]) + (e->t->in_ptr[d + 1] - e->t->in_ptr[d]);
  for (int j = 0; j < x_n(e); ++j)
    deg += ((int)(e->extra[j] >> 16) == d) + ((int)(e->extra[j] & 0xFFFFu) == d);
  return deg;
}
/* g.add_edges([(u, v)]): sorted insert into the env's list; 1 when the edge went in */
static int add_edge(env_t* e, int u, int v) {
  const int n = x_n(e);
  if (!e->extra || n >= e->K) { e->ienv[CG_I_FLAGS] |= CG_E_TOPO_OVF; return 0; }
  const uint32_t key = ((uint32_t)u << 16) | (uint32_t)v;
  int j = n;
  while (j > 0 && e->extra[j - 1] > key) { e->extra[j] = e->extra[j - 1]; --j; }
  e->extra[j] = key;
  x_set_n(e, n + 1);
  return 1;
}

static void evolve(env_t* e, uint8_t* newly, int* cdf) { /* CyberDefenseEnv.py:583-875 */
  const int M = e->M;
  if (!(e->ienv[CG_I_FLAGS] & CG_E_EVO_INIT)) { /* :654-659 */
    for (int d = 0; d < M; ++d) {
      if (e->flags[d] & CG_F_NYA) e->flags[d] &= (uint8_t)~CG_F_EVOACT; else e->flags[d] |= CG_F_EVOACT;
    }
    e->ienv[CG_I_FLAGS] |= CG_E_EVO_INIT;
  }
  memset(newly, 0, (size_t)M);
  int n_ev = cg_cdf_lookup(drw(e, CG_SITE_EVO_POISSON, 0, 0), e->c->poisson_thr, CG_POISSON_TABLE);
  int floor_n = e->c->num_of_device > e->c->min_network_size ? e->c->num_of_device : e->c->min_network_size;
  for (int ev = 0; ev < n_ev; ++ev) {
    if (cg_bernoulli(drw(e, CG_SITE_EVO_COIN, (uint32_t)ev, 0), e->c->p_add_thr)) {
      int n_in = 0;
      for (int d = 0; d < M; ++d) n_in += !(e->flags[d] & CG_F_EVOACT);
      if (n_in > 0) {
        int d = rank_select(e, CG_F_EVOACT, 0, (int)cg_index(drw(e, CG_SITE_EVO_PICK_IN, (uint32_t)ev, 0), (uint32_t)n_in));
        e->flags[d] &= (uint8_t)~CG_F_NYA;
        e->flags[d] |= CG_F_EVOACT;
        newly[d] = 1;
        if (cg_bernoulli(drw(e, CG_SITE_EVO_ATT, (uint32_t)ev, 0), e->c->p_attacker_thr))
          e->flags[d] |= (CG_F_COMP | CG_F_OWNED | CG_F_KNOWN);
      }
    } else {
      int n_act = 0;
      for (int d = 0; d < M; ++d) n_act += (e->flags[d] & CG_F_EVOACT) != 0;
      if (n_act > floor_n) {
        int d = rank_select(e, CG_F_EVOACT, CG_F_EVOACT, (int)cg_index(drw(e, CG_SITE_EVO_PICK_ACT, (uint32_t)ev, 0), (uint32_t)n_act));
        e->flags[d] |= CG_F_NYA;
        clear_wl(e, d);
        e->busy[d] = 0;
        e->flags[d] &= (uint8_t)~CG_F_EVOACT;
        newly[d] = 0; /* removal after activation in the same call leaves it in newly_activated,
                         but it is then skipped as Not_yet_added (:783) */
      }
    }
  }
  int changed = 0;
  /* star reconnection :738-774: hub = first active attacker-owned device in dict order */
  int hub = -1;
  for (int d = 0; d < M; ++d) {
    if ((e->flags[d] & (CG_F_OWNED | CG_F_EVOACT)) != (CG_F_OWNED | CG_F_EVOACT)) continue;
    if (hub < 0) { hub = d; continue; }
    if (!has_edge(e, hub, d)) changed |= add_edge(e, hub, d);
    if (!has_edge(e, d, hub)) changed |= add_edge(e, d, hub);
  }
  /* PA attachment of isolated newcomers :776-843 (one degree snapshot, taken after the star edges) */
  int need_pa = 0;
  for (int d = 0; d < M && !need_pa; ++d)
    if (newly[d] && !(e->flags[d] & (CG_F_NYA | CG_F_OWNED)) && degree_all(e, d) < 1) need_pa = 1;
  if (need_pa) {
    int total = 0;
    for (int d = 0; d < M; ++d) { /* candidates: self._active_ids (ascending ids), weight degree + 1 */
      if (e->flags[d] & CG_F_EVOACT) total += degree_all(e, d) + 1;
      cdf[d] = total;
    }
    for (int d = 0; d < M && total > 0; ++d) {
      if (!newly[d] || (e->flags[d] & (CG_F_NYA | CG_F_OWNED))) continue;
      if (degree_all(e, d) >= 1) continue;
      /* r = random.uniform(0, total); j = bisect_left(cdf, r): first candidate with cdf >= total * u / 2^32 */
      const uint64_t r = (uint64_t)total * (uint64_t)drw(e, CG_SITE_EVO_PA, (uint32_t)d, 0);
      int tgt = -1;
      for (int a = 0; a < M; ++a)
        if ((e->flags[a] & CG_F_EVOACT) && ((uint64_t)cdf[a] << 32) >= r) { tgt = a; break; }
      if (tgt >= 0 && !has_edge(e, d, tgt)) changed |= add_edge(e, d, tgt);
    }
  }
  if (changed) { /* _rebuild_graph_cache (volt_typhoon_env.py:456-481) starts from an empty _blocked set */
    memset(e->blocked, 0, (size_t)e->EW * 4);
    memset(e->extra + e->K, 0, (size_t)((e->K + 31) / 32) * 4);
  }
}

static void write_obs(const env_t* e, float* obs) { /* _get_state CyberDefenseEnv.py:146-191 */
  for (int d = 0; d < e->M; ++d) {
    float* r = obs + 6 * d;
    uint8_t f = e->flags[d];
    r[0] = e->t->os_val[d];
    r[1] = e->t->version[d];
    r[2] = (f & CG_F_COMP) ? 1.f : 0.f;
    r[3] = e->anomaly ? e->anomaly[d] : e->t->anomaly[d];
    r[4] = (f & CG_F_KNOWN) ? 1.f : 0.f;
    r[5] = (f & CG_F_NYA) ? 1.f : 0.f;
  }
}

static void count_comp(const env_t* e, int* n_comp, int* n_comp_dc) { /* _count_comp :563-572 */
  int a = 0, b = 0;
  for (int d = 0; d < e->M; ++d) {
    uint8_t f = e->flags[d];
    if ((f & CG_F_COMP) && !(f & CG_F_NYA) && !(f & CG_F_OWNED)) {
      ++a;
      if (e->t->dstatic[d] & CG_D_DC) ++b;
    }
  }
  *n_comp = a; *n_comp_dc = b;
}

static void rewards(env_t* e, int mode, double cost, int current_work, double* raw, double* shaped) {
  int n_comp, n_comp_dc;
  count_comp(e, &n_comp, &n_comp_dc);
  e->ienv[CG_I_LAST_NCOMP] = n_comp;
  double def_work = e->c->work_scale * current_work;
  if (mode == CG_MODE_DEFENDER) {
    *raw = cost + def_work - n_comp * e->c->comp_scale; /* :1291 */
    *shaped = *raw;
  } else {
    double r = cost + e->c->comp_scale * (n_comp + 10 * n_comp_dc); /* :1294 */
    double phi = e->M > 0 ? (double)n_comp / (double)e->M : 0.0;
    if (!(e->ienv[CG_I_FLAGS] & CG_E_PREV_SET)) {
      e->fenv[CG_D_PREV_ATT_POT] = phi;
      e->ienv[CG_I_FLAGS] |= CG_E_PREV_SET;
    }
    double inc = e->c->gamma * phi - e->fenv[CG_D_PREV_ATT_POT];
    double bonus = 0.1 * inc + 0.0;
    e->fenv[CG_D_PREV_ATT_POT] = e->c->gamma * phi;
    *raw = r;
    *shaped = r + bonus;
  }
}

static void snapshot_restore(env_t* e, const cygym_buffers* snap, int idx);

typedef struct { uint8_t* newly; uint8_t* occ; int* srcs; int* pu; int* pv; int* pr; int* cdf; ment_t* row; kid_t* tmp; } scratch_t;

/* role 0: _get_state, 1: _get_defender_state, 2: _get_attacker_state (CyberDefenseEnv.py:146-257) */
static void observe_one(env_t* e, int role, float* o) {
  const int M = e->M;
  const cygym_topology* t = e->t;
  if (role == 0 || role == 1) {
    write_obs(e, o);
    if (role == 1) {
      for (int d = 0; d < M; ++d) {
        uint8_t f = e->flags[d];
        if ((f & CG_F_NYA) || !(f & CG_F_OWNED)) for (int k = 0; k < 6; ++k) o[6 * d + k] = -1.f;
        o[6 * d + 2] = -1.f;
      }
    }
  } else {
    for (int d = 0; d < M; ++d) {
      uint8_t f = e->flags[d];
      int vis = (f & CG_F_KNOWN) && !(f & CG_F_NYA) && (f & CG_F_OWNED);
      o[4 * d + 0] = vis ? t->os_val[d] : -1.f;
      o[4 * d + 1] = vis ? t->version[d] : -1.f;
      o[4 * d + 2] = vis ? ((f & CG_F_COMP) ? 1.f : 0.f) : -1.f;
      o[4 * d + 3] = vis ? ((f & CG_F_KNOWN) ? 1.f : 0.f) : -1.f;
    }
    for (int k = 0; k < e->c->max_exploits; ++k) o[4 * M + k] = (k < t->n_exploits) ? 1.f : 0.f;
  }
}

static void step_one(env_t* e, const cygym_actions* a, const cygym_outputs* o, int idx, const scratch_t* sc,
                     const cygym_buffers* snap) {
  const int M = e->M;
  const int G = a->max_groups, L = a->max_devs;
  uint8_t* newly = sc->newly;
  uint8_t* occ = sc->occ;
  int* srcs = sc->srcs;
  int* pu = sc->pu;
  int* pv = sc->pv;
  int* pr = sc->pr;
  ment_t* row = sc->row;
  kid_t* tmp = sc->tmp;
  memset(occ, 0, (size_t)(3 * M));
  e->occ = occ;
  e->isolate_iter = 0; e->scan_ord = 0; e->zeroday_occ = 0;
  e->tick = (uint32_t)e->ienv[CG_I_RNG_TICK];
  const int partial = (a->mode[idx] & CG_MODE_PARTIAL) != 0 && a->n_groups[idx] == 0; /* agent_cnt mismatch :1207 */
  const int mode = a->mode[idx] & 0xFF;
  const int baseline = CG_MODE_BASELINE_OF(a->mode[idx], e->c->baseline); /* env.base_line of this env for this tick */
  e->baseline = baseline;
  const int ng = a->n_groups[idx];
  if (ng < 0) return; /* this env does not tick */
  const int16_t* devs = a->dev_idx + (size_t)idx * L;
  double cost = 0.0;
  int dirty = 0;
  int current_work;
  double raw, shaped;
  if (ng == 0) {
    int at = a->atype[(size_t)idx * G];
    int Ld = a->dev_cnt[(size_t)idx * G];
    if (Ld > L) Ld = L;
    if (mode == CG_MODE_DEFENDER) { if (!(at >= 0 && at < e->c->n_def_actions)) at = 8; }
    else                          { if (!(at >= 0 && at < e->c->n_att_actions)) at = 3; }
    for (int d = 0; d < M; ++d) { /* :904-908 */
      if (e->flags[d] & CG_F_BUSYC) {
        if (e->busy[d] > 0) e->busy[d]--;
      }
    }
    if (mode == CG_MODE_DEFENDER) {
      if (baseline != 0) at = 8;
      def_global(e, at, devs, Ld, &cost, &dirty, 0);
      if (at == 1 || at == 4 || at == 5 || at == 6 || at == 7 || at == 9 || at == 12 || at == 13)
        def_per_device(e, at, devs, Ld, a->app[(size_t)idx * G], &cost, &dirty, pu, pv, pr, row);
    } else {
      attacker(e, at, a->exploit + (size_t)idx * G * CG_MAX_EXPLOITS, a->n_exploit[(size_t)idx * G], &cost, srcs, row);
    }
    current_work = 0;
    if (!partial) {
      current_work = advance_work(e);
      arrivals(e, tmp);
    }
    {
      int n_comp, n_dc;
      count_comp(e, &n_comp, &n_dc);
      e->ienv[CG_I_COMP_CNT] += n_comp; /* :1267-1270 */
    }
    rewards(e, mode, cost, current_work, &raw, &shaped);
    e->ienv[CG_I_LAST_ATYPE] = at;
  } else {
    const int16_t* dp = devs;
    int used = 0;
    for (int g = 0; g < ng && g < G; ++g) { /* _step_apply_only :612-692 */
      int at = a->atype[(size_t)idx * G + g];
      int Ld = a->dev_cnt[(size_t)idx * G + g];
      if (used + Ld > L) Ld = L - used;
      if (mode == CG_MODE_DEFENDER && at == 0) at = 8;
      else if (mode == CG_MODE_ATTACKER && at == 0) at = 3;
      if (mode == CG_MODE_DEFENDER) {
        if (baseline != 0) at = 8;
        def_global(e, at, dp, Ld, &cost, &dirty, 1);
        if (at == 1) def_per_device(e, 1, dp, Ld, -1, &cost, &dirty, pu, pv, pr, row);
      }
      dp += Ld; used += Ld;
    }
    for (int d = 0; d < M; ++d) if (e->busy[d] > 0) e->busy[d]--; /* _tick_busy_time_once :607 */
    current_work = advance_work(e);
    arrivals(e, tmp);
    rewards(e, mode, cost, current_work, &raw, &shaped);
    e->ienv[CG_I_LAST_ATYPE] = -1;
  }
  if (o->obs) write_obs(e, o->obs + (size_t)idx * M * 6);
  if (!partial) { /* :1307-1312 */
    e->ienv[CG_I_STEP_NUM] += 1;
    if (mode == CG_MODE_ATTACKER) e->ienv[CG_I_ATT_STEP] += 1; else e->ienv[CG_I_DEF_STEP] += 1;
  }
  int done = e->ienv[CG_I_STEP_NUM] > e->c->episode_limit;
  if (dirty || (e->ienv[CG_I_STEP_NUM] % e->c->evolve_period) == 0) evolve(e, newly, sc->cdf);
  if (ng == 0) { /* :1330 */
    for (int d = 0; d < M; ++d) {
      if (e->busy[d] > 0) e->flags[d] |= CG_F_BUSYC; else e->flags[d] &= (uint8_t)~CG_F_BUSYC;
    }
  }
  e->ienv[CG_I_RNG_TICK] += 1;
  o->raw[idx] = raw;
  o->shaped[idx] = shaped;
  o->done[idx] = (uint8_t)done;
  if (o->ret && o->alive && o->alive[idx]) { /* episode returns of a rollout loop (do_agent.py:266-274) */
    o->ret[(size_t)idx * 2 + (mode & 1)] += raw;
    if (done) o->alive[idx] = 0;
  }
  if (done && e->c->auto_reset && snap) snapshot_restore(e, snap, idx);
  /* optional role views of the state the tick leaves behind (cygym_outputs.obs_def / obs_att) + the status word */
  if (o->obs_def) observe_one(e, 1, o->obs_def + (size_t)idx * 6 * M);
  if (o->obs_att) observe_one(e, 2, o->obs_att + (size_t)idx * (4 * M + e->c->max_exploits));
  if (o->status) *o->status |= (uint32_t)e->ienv[CG_I_FLAGS] & (CG_E_TOPO_OVF | CG_E_BUSY_SAT | CG_E_DET_PENDING | CG_E_UNPINNED);
}

static void bind_env(env_t* e, const cygym_topology* t, const cygym_config* c, const cygym_buffers* b, int idx) {
  e->t = t; e->c = c;
  e->M = t->n_devices; e->X = t->n_exploits; e->E = t->n_edges; e->EW = (t->n_edges + 31) / 32 > 0 ? (t->n_edges + 31) / 32 : 1;
  size_t o = (size_t)idx * e->M * CG_PLANES;
  e->flags = b->live + o; e->busy = e->flags + e->M; e->wl = e->busy + e->M; e->comp_by = e->wl + e->M;
  e->st_flags = b->stash + o; e->st_busy = e->st_flags + e->M; e->st_wl = e->st_busy + e->M; e->st_comp_by = e->st_wl + e->M;
  e->blocked = b->blocked + (size_t)idx * e->EW;
  e->K = b->extra ? t->max_extra_edges : 0;
  e->extra = e->K > 0 ? b->extra + (size_t)idx * CG_X_WORDS(e->K) : NULL;
  e->ring = b->ring + (size_t)idx * CG_LOG_RING * 2;
  e->hist = b->hist ? b->hist + (size_t)idx * CG_HIST_RING * 2 : NULL;
  e->forest = b->forest ? b->forest + (size_t)idx * CG_FOREST_WORDS : NULL;
  e->anomaly = b->anomaly ? b->anomaly + (size_t)idx * e->M : NULL;
  e->ienv = b->ienv + (size_t)idx * CG_I_COUNT;
  e->fenv = b->fenv + (size_t)idx * CG_D_COUNT;
  e->env_id = (uint32_t)(c->env_id_base + idx);
}

static void snapshot_restore(env_t* e, const cygym_buffers* s, int idx) {
  int si = s->n_envs == 1 ? 0 : idx;
  size_t o = (size_t)si * e->M * CG_PLANES;
  int32_t tick = e->ienv[CG_I_RNG_TICK];
  memcpy(e->flags, s->live + o, (size_t)e->M * CG_PLANES);
  memcpy(e->st_flags, s->stash + o, (size_t)e->M * CG_PLANES);
  memcpy(e->blocked, s->blocked + (size_t)si * e->EW, (size_t)e->EW * 4);
  if (e->extra && s->extra) memcpy(e->extra, s->extra + (size_t)si * CG_X_WORDS(e->K), (size_t)CG_X_WORDS(e->K) * 4);
  memcpy(e->ring, s->ring + (size_t)si * CG_LOG_RING * 2, CG_LOG_RING * 2 * 2);
  if (e->hist && s->hist) memcpy(e->hist, s->hist + (size_t)si * CG_HIST_RING * 2, CG_HIST_RING * 2 * 2);
  if (e->forest && s->forest) memcpy(e->forest, s->forest + (size_t)si * CG_FOREST_WORDS, CG_FOREST_WORDS * 4);
  if (e->anomaly && s->anomaly) memcpy(e->anomaly, s->anomaly + (size_t)si * e->M, (size_t)e->M * 4);
  memcpy(e->ienv, s->ienv + (size_t)si * CG_I_COUNT, CG_I_COUNT * 4);
  memcpy(e->fenv, s->fenv + (size_t)si * CG_D_COUNT, CG_D_COUNT * 8);
  e->ienv[CG_I_RNG_TICK] = tick; /* the draw counter is monotone across episodes */
}

int cgo_step(const cygym_topology* t, const cygym_config* c, const cygym_buffers* b,
             const cygym_actions* a, const cygym_outputs* o, const cygym_buffers* snapshot,
             int32_t env_begin, int32_t env_end) {
  size_t M = (size_t)t->n_devices, E = (size_t)t->n_edges;
  scratch_t sc;
  if (!c->fast_scan && (!b->hist || !b->anomaly)) return CYGYM_EINVAL; /* the per-log scan path reads the long history */
  const size_t K = (size_t)(t->max_extra_edges > 0 ? t->max_extra_edges : 0);
  sc.newly = (uint8_t*)malloc(M + 1);
  sc.occ = (uint8_t*)malloc(3 * M + 1);
  sc.srcs = (int*)malloc(sizeof(int) * (M + 1));
  sc.pu = (int*)malloc(sizeof(int) * (2 * E + 2 * K + 2));
  sc.pv = (int*)malloc(sizeof(int) * (2 * E + 2 * K + 2));
  sc.pr = (int*)malloc(sizeof(int) * (2 * E + 2 * K + 2));
  sc.cdf = (int*)malloc(sizeof(int) * (M + 1));
  sc.row = (ment_t*)malloc(sizeof(ment_t) * (E + K + 2));
  sc.tmp = (kid_t*)malloc(sizeof(kid_t) * (M + 1));
  if (!sc.newly || !sc.occ || !sc.srcs || !sc.pu || !sc.pv || !sc.pr || !sc.cdf || !sc.row || !sc.tmp) return CYGYM_EINVAL;
  for (int i = env_begin; i < env_end; ++i) {
    env_t e;
    bind_env(&e, t, c, b, i);
    step_one(&e, a, o, i, &sc, snapshot);
  }
  free(sc.newly); free(sc.occ); free(sc.srcs); free(sc.pu); free(sc.pv); free(sc.pr); free(sc.cdf); free(sc.row); free(sc.tmp);
  return CYGYM_OK;
}

int cgo_reset(const cygym_topology* t, const cygym_config* c, const cygym_buffers* b,
              const cygym_buffers* snapshot, const int32_t* env_ids, int32_t n) {
  for (int j = 0; j < n; ++j) {
    int i = env_ids ? env_ids[j] : j;
    env_t e;
    bind_env(&e, t, c, b, i);
    snapshot_restore(&e, snapshot, i);
  }
  return CYGYM_OK;
}

int cgo_randomize(const cygym_topology* t, const cygym_config* c, const cygym_buffers* b,
                  const int32_t* env_ids, int32_t n) { /* volt_typhoon_env.py:330-383 */
  kid_t* tmp = (kid_t*)malloc(sizeof(kid_t) * (size_t)(t->n_devices + 1));
  for (int j = 0; j < n; ++j) {
    int i = env_ids ? env_ids[j] : j;
    env_t e;
    bind_env(&e, t, c, b, i);
    e.tick = (uint32_t)e.ienv[CG_I_RNG_TICK];
    int cnt = 0, k_owned = 0, k_comp = 0;
    for (int d = 0; d < e.M; ++d) {
      if ((e.flags[d] & CG_F_NYA) || (t->dstatic[d] & CG_D_DC)) continue;
      tmp[cnt].key = drw(&e, CG_SITE_SHUFFLE, (uint32_t)d, 0);
      tmp[cnt].id = d; ++cnt;
      k_owned += (e.flags[d] & CG_F_OWNED) != 0;
      k_comp += (e.flags[d] & CG_F_COMP) != 0;
    }
    e.ienv[CG_I_RNG_TICK] += 1;
    if (cnt == 0 || (k_owned == 0 && k_comp == 0)) continue;
    qsort(tmp, (size_t)cnt, sizeof(kid_t), kid_cmp);
    int extra = k_comp - k_owned; if (extra < 0) extra = 0;
    for (int r = 0; r < cnt; ++r) {
      int d = tmp[r].id;
      e.flags[d] &= (uint8_t)~(CG_F_OWNED | CG_F_COMP | CG_F_KNOWN);
      if (r < k_owned) e.flags[d] |= (CG_F_OWNED | CG_F_COMP | CG_F_KNOWN);
      else if (r < k_owned + extra) e.flags[d] |= (CG_F_COMP | CG_F_KNOWN);
    }
  }
  free(tmp);
  return CYGYM_OK;
}

/* role 0: _get_state, 1: _get_defender_state, 2: _get_attacker_state (CyberDefenseEnv.py:146-257) */
int cgo_observe(const cygym_topology* t, const cygym_config* c, const cygym_buffers* b, int32_t role,
                float* out, int32_t n_envs) {
  const int M = t->n_devices;
  for (int i = 0; i < n_envs; ++i) {
    env_t e;
    bind_env(&e, t, c, b, i);
    observe_one(&e, role, out + (size_t)i * (role == 2 ? 4 * M + c->max_exploits : 6 * M));
  }
  return CYGYM_OK;
}

int cgo_abi_version(void) { return CYGYM_ABI_VERSION; }
