/* cygym_spec.h -- bit layouts, RNG sites and draw conventions of the batched
 * CyGym tick.  Shared verbatim by the HIP kernels (cygym_amd/csrc), the C oracle
 * (oracle/cygym_oracle.c) and mirrored in Python (cygym_amd/spec.py).
 *
 * Nothing here is copied from the reference; it is the flat (struct-of-arrays)
 * restatement of the object graph the reference mutates:
 *   Device flags ........ CDSimulatorComponents.py:217-243
 *   Workload ............ CDSimulatorComponents.py:18-26
 *   device stash ........ volt_typhoon_env.py:419-453
 *   busy-set cache ...... volt_typhoon_env.py:117, 904-908, 1330
 *   _active_ids set ..... CyberDefenseEnv.py:654-659
 */
#ifndef CYGYM_SPEC_H
#define CYGYM_SPEC_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define CYGYM_HD __host__ __device__ __forceinline__
#else
#define CYGYM_HD static inline
#endif

/* ---- per-device dynamic flag byte (plane `flags`, u8 [N][M]) ---- */
#define CG_F_COMP       0x01u /* Device.isCompromised                      */
#define CG_F_OWNED      0x02u /* Device.attacker_owned                     */
#define CG_F_KNOWN      0x04u /* Device.Known_to_attacker                  */
#define CG_F_REACH      0x08u /* Device.reachable_by_attacker              */
#define CG_F_NYA        0x10u /* Device.Not_yet_added                      */
#define CG_F_EVOACT     0x20u /* member of env._active_ids (evolve_network)*/
#define CG_F_BUSYC      0x40u /* member of env._busy_devices (cached set)  */
#define CG_F_WLADV      0x80u /* Device.workload.adversarial               */

/* ---- per-device stash flag byte (plane `st_flags`) -- action 11/12 ---- */
#define CG_S_VALID      0x40u /* device id present in env._device_ckpts    */
/* bits COMP/KNOWN/REACH/NYA/WLADV reuse the CG_F_* positions              */
#define CG_S_KEEP (CG_F_COMP | CG_F_KNOWN | CG_F_REACH | CG_F_NYA | CG_F_WLADV)

/* ---- per-device static byte (plane `dstatic`, u8 [M], per topology) ---- */
#define CG_D_DC         0x01u /* device_type == "DomainController"         */
#define CG_D_SERVER     0x02u /* wtype == 'server'                         */
/* vuln mask (which exploit indices can compromise this device) is its own
 * plane `vuln` u8 [M]; number of apps is plane `napps` u8 [M].             */

/* ---- per-env scalar flags (i32 column ENV_FLAGS) ---- */
#define CG_E_HAS_CKPT   0x01 /* env.checkpoint is not None                 */
#define CG_E_EVO_INIT   0x02 /* env._active_ids exists                     */
#define CG_E_DET_TRAIN  0x04 /* simulator.detector.trained                 */
#define CG_E_DET_RANDOM 0x08 /* simulator.detector.random_detection        */
#define CG_E_PREV_SET   0x10 /* env._prev_att_potential is not None        */
#define CG_E_TOPO_OVF   0x20 /* evolve wanted to add an edge that did not fit
                                the per-env extra-edge list (max_extra_edges) */
#define CG_E_BUSY_SAT   0x40 /* a busy counter saturated at 255            */
/* 0x80 is kernel-private (star edges verified for the current owned set)   */
#define CG_E_DET_PENDING 0x100 /* action 10 asked for Detector.train(non-empty logs) (volt_typhoon_env.py:945-962):
                                  the host must fit the env's isolation forest and install it (cygym_abi.h,
                                  cygym_buffers.forest) before the next scan                            */
#define CG_E_UNPINNED   0x200 /* sticky: a scan ran in TRAINED mode without a current forest (no forest buffer
                                  bound, or CG_E_DET_PENDING still set); its predictions were taken as all "D",
                                  which the reference does not promise -- results of this env are unpinned */
#define CG_E_NX_SHIFT   16   /* bits 16..31: number of live entries of the env's
                                extra-edge list (edges evolve_network added,
                                CyberDefenseEnv.py:738-843)                   */
#define CG_E_NX(f)      (((uint32_t)(f)) >> CG_E_NX_SHIFT)
/* extra-edge list of one env, u32 [CG_X_WORDS(K)], K = max_extra_edges:
 *   words [0, K)      keys (u << 16 | v) of the added directed edges, ascending; the first NX are live
 *   words [K, K+KW)   blocked bit per list entry, KW = ceil(K/32)
 * A row of the cached adjacency is then the base CSR row merged with the env's extra edges in
 * ascending neighbour id (igraph returns neighbour lists sorted by vertex id; an added edge never
 * duplicates an existing one, CyberDefenseEnv.py:756-759, :824).                                  */
#define CG_X_WORDS(K)   ((K) + ((K) + 31) / 32)

/* ---- per-env integer counters: column indices of `ienv` i32 [N][CG_I_COUNT] */
enum {
  CG_I_STEP_NUM = 0,     /* env.step_num                                   */
  CG_I_DEF_STEP,         /* env.defender_step                              */
  CG_I_ATT_STEP,         /* env.attacker_step                              */
  CG_I_WORK_DONE,        /* env.work_done                                  */
  CG_I_CKPT_CNT,         /* env.checkpoint_count                           */
  CG_I_REVERT_CNT,       /* env.revert_count                               */
  CG_I_SCAN_CNT,         /* env.scan_cnt                                   */
  CG_I_COMP_CNT,         /* env.compromised_devices_cnt                    */
  CG_I_EDGES_BLOCKED,    /* env.edges_blocked                              */
  CG_I_EDGES_ADDED,      /* env.edges_added                                */
  CG_I_FLAGS,            /* CG_E_* bits                                    */
  CG_I_RNG_TICK,         /* monotone tick counter feeding the Philox ctr   */
  CG_I_LOG_TOTAL,        /* len(simulator.logger.logs)                     */
  CG_I_DISCOVERED,       /* bitmask over exploit index: Exploit.discovered */
  CG_I_LAST_NCOMP,       /* n_comp of the last tick (for host-side info)   */
  CG_I_LAST_ATYPE,       /* info['executed_atype'] of the last tick        */
  CG_I_COUNT
};

/* ---- per-env f64 accumulators: `fenv` f64 [N][CG_D_COUNT] ---- */
enum {
  CG_D_DEF_COST = 0,     /* env.defensive_cost                             */
  CG_D_CLEAN_COST,       /* env.clearning_cost (sic)                       */
  CG_D_PREV_ATT_POT,     /* env._prev_att_potential                        */
  CG_D_COUNT
};

/* ---- actions ---- */
#define CG_MODE_DEFENDER 0
#define CG_MODE_ATTACKER 1
#define CG_MODE_PARTIAL  0x100 /* OR-ed into mode: step(action, agent_cnt != len(net)) -- no workload advance,
                                  no arrivals, no step counters (volt_typhoon_env.py:1207, :1307) */
/* env.base_line of THIS env for THIS tick, OR-ed into mode: bits 16..18 hold (baseline code + 1), 0 = use
 * cygym_config.baseline.  The reference's rollout loops assign env.base_line per env and per turn (a baseline
 * strategy sets it before its step, do_agent.py:218-221, and it stays set for the other role's turns:
 * volt_typhoon_env.py:913-914 then turns every defender action into a no-op, :1126 skips the attacker). */
#define CG_MODE_BASELINE_SHIFT 16
#define CG_MODE_BASELINE(code) ((((code) & 3) + 1) << CG_MODE_BASELINE_SHIFT)
#define CG_MODE_BASELINE_OF(mode, cfg_baseline) \
  ((((mode) >> CG_MODE_BASELINE_SHIFT) & 7) ? ((((mode) >> CG_MODE_BASELINE_SHIFT) & 7) - 1) : (cfg_baseline))

/* ---- comm-log ring ---- */
#define CG_LOG_RING 32       /* entries kept per env (fast scan reads 30)  */
#define CG_SCAN_WINDOW 30    /* volt_typhoon_env.py:1052                   */
#define CG_SLOW_SCAN_WINDOW 256 /* fast_scan=False: logs[-512:][-256:] are predicted one by one (:1022-1031); needs the
                                   long history (cygym_buffers.hist) and the per-env anomaly plane (cygym_buffers.anomaly) */
#define CG_HIST_RING 2048    /* optional long history (cygym_buffers.hist): what Detector.train fits on    */
#define CG_TRAIN_WINDOW 2000 /* action 10 trains on the last <= 2000 logs  volt_typhoon_env.py:958        */

/* ---- trained detector: IsolationForest(n_estimators=2, max_samples=256) CDSimulator.py:683 ----
 * One env's fitted forest, u32 [CG_FOREST_WORDS] (cygym_buffers.forest):
 *   header  [0],[1]  S*: f64 bits (lo, hi).  A log point is an anomaly ("A") iff  v0 + v1 < S*, where
 *                    v_t = (depth_t + apl[n_t]) - 1.0 is the value of the leaf it reaches in tree t
 *                    (sklearn _parallel_compute_tree_depths) -- S* is the smallest f64 sum for which
 *                    sklearn's own  -(2 ** -(s / (2 * apl[max_samples_]))) - offset_ < 0  is false; the host
 *                    finds it by bisection with the same numpy expressions, so the kernel needs no pow.
 *           [2]      node count of tree 0 | tree 1 << 16; 0 = no trees were ever installed (a scan in trained mode then
 *                    answers all "D" and raises CG_E_UNPINNED, like a pending request)
 *           [3]      rng tick of the action-10 tick that asked for this training  (written by the tick)
 *           [4]      len(logger.logs) at that moment                              (written by the tick)
 *           [5]      rng tick whose request the installed forest answers          (written by the host)
 *           [6]      how many times that tick asked (a step_grouped tick may carry action 10 in several groups:
 *                    the reference then fits several times on the same logs, each fit continuing the numpy
 *                    stream -- the host does the same and keeps the last forest)   (written by the tick)
 *           [7]      max_samples_ of the fit (min(256, training rows)): the slow scan path turns a point's summed depth s into
 *                    sklearn's decision_function value  0.5 - 2^(-s / (2 * apl[max_samples_]))  (device.anomaly_score,
 *                    volt_typhoon_env.py:1033-1035)                                    (written by the host)
 *   tree t: words [CG_FOREST_HDR + t * CG_FOREST_NODES, +node count), node 0 = root
 *     internal node: bit 31 = 0 | feature << 30 (0: from_device, 1: to_device) | floor(threshold) << 18 (12 bits)
 *                    | left child << 9 | right child;  go left iff x[feature] <= threshold (ids are integers,
 *                    so comparing with floor(threshold) is exact)
 *     leaf:          bit 31 = 1 | depth << 9 (root = 1, sklearn compute_node_depths) | n_node_samples (9 bits)
 * apl[n] = sklearn _average_path_length(n), n = 0..CG_DET_APL_N-1: a per-handle f64 table handed over by the host
 * (cygym_topology.det_apl) so that no log() is evaluated on the device or in the oracle.                      */
#define CG_FOREST_TREES 2
#define CG_FOREST_NODES 512
#define CG_FOREST_HDR 8
#define CG_FOREST_WORDS (CG_FOREST_HDR + CG_FOREST_TREES * CG_FOREST_NODES)
#define CG_DET_APL_N 257
#define CG_FN_LEAF(w)   ((w) >> 31)
#define CG_FN_FEAT(w)   (((w) >> 30) & 1u)
#define CG_FN_THR(w)    (((w) >> 18) & 0xFFFu)
#define CG_FN_LEFT(w)   (((w) >> 9) & 0x1FFu)
#define CG_FN_RIGHT(w)  ((w) & 0x1FFu)
#define CG_FN_DEPTH(w)  (((w) >> 9) & 0xFu)
#define CG_FN_NSAMP(w)  ((w) & 0x1FFu)

/* ---- RNG sites: one id per random call site on the step path ---- */
enum {
  CG_SITE_STALL_REVERT = 1,  /* volt_typhoon_env.py:936 / :643   a=device            */
  CG_SITE_STALL_CLEAN,       /* :1009 / :689                     a=device b=occurrence*/
  CG_SITE_STALL_PATCH,       /* :1018                            a=device b=occurrence*/
  CG_SITE_STALL_SCAN,        /* :1069                            a=sender b=scan ord. */
  CG_SITE_STALL_ISOLATE,     /* :1120                            a=device b=iteration */
  CG_SITE_PICK_BLOCK,        /* :505                             a=device b=occurrence*/
  CG_SITE_PICK_UNBLOCK,      /* :511                             a=device b=occurrence*/
  CG_SITE_PROBE_SRC,         /* :1189                                                 */
  CG_SITE_ZERODAY,           /* :1136                            a=position in list   */
  CG_SITE_ARR_CLIENT,        /* CDSimulator.py:298 (wtype client) a=device (sort key) */
  CG_SITE_ARR_SERVER,        /* CDSimulator.py:298 (wtype server) a=device (sort key) */
  CG_SITE_ARR_TIME,          /* CDSimulator.py:308               a=device             */
  CG_SITE_EVO_POISSON,       /* CyberDefenseEnv.py:668                                */
  CG_SITE_EVO_COIN,          /* :679                             a=event index        */
  CG_SITE_EVO_PICK_IN,       /* :681 -> :675                     a=event index        */
  CG_SITE_EVO_PICK_ACT,      /* :703 -> :675                     a=event index        */
  CG_SITE_EVO_ATT,           /* :690                             a=event index        */
  CG_SITE_EVO_PA,            /* :817                             a=device             */
  CG_SITE_SHUFFLE,           /* volt_typhoon_env.py:359          a=device (sort key)  */
  CG_SITE_DET_COIN,          /* CDSimulator.py:716               a=window pos b=scan  */
  CG_SITE_LAZY,              /* CDSimulator.py:328 (no observable effect)             */
  CG_SITE_DET_FIT,           /* CDSimulator.py:694 IsolationForest.fit: seed of the numpy stream it draws from */
  CG_SITE_ACTGEN = 64,       /* synthetic action script of bench.py (not reference)   */
  CG_SITE_EPS_TYPE = 65,     /* do_agent.py:972-973 epsilon-greedy action type of decode_action: word 0 is the coin
                                (u < ceil(eps * 2^32)), word 1 the uniform type index.  Addressed by the env's own rng
                                tick, read when the action is decoded (the tick that will execute it)            */
  CG_SITE_GROUP_PICK = 66,   /* IPPO.py:566-567 / MAPPO.py: random.choice(devs) of a single-device action type when
                                per-device types are grouped (cygym_group_actions): a = action type; addressed by the
                                env's rng tick like CG_SITE_EPS_TYPE                                              */
  CG_SITE_SAMPLE = 67        /* IPPO.py:524-555 Categorical(logits).sample() of cygym_sample_group_actions: a = device id,
                                b = 0 (per-device type) / a = 0, b = 1 (exploit) / b = 2 (app); u = word 0 / 2^32 walks
                                the inverse CDF of softmax(logits); same addressing                               */
};

/* ---- Philox4x32-10 (Salmon et al., SC'11), counter-based ----
 * key  = (seed_lo, seed_hi)
 * ctr  = (global_env_id, rng_tick, site, a | (b << 16))
 * A draw is word 0 of the output; words 1..3 are used only where stated.  */
#define CG_PHILOX_M0 0xD2511F53u
#define CG_PHILOX_M1 0xCD9E8D57u
#define CG_PHILOX_W0 0x9E3779B9u
#define CG_PHILOX_W1 0xBB67AE85u

typedef struct { uint32_t v[4]; } cg_u32x4;

CYGYM_HD cg_u32x4 cg_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                   uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)CG_PHILOX_M0 * c0;
    uint64_t p1 = (uint64_t)CG_PHILOX_M1 * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += CG_PHILOX_W0; k1 += CG_PHILOX_W1;
  }
  cg_u32x4 o; o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

/* one 32-bit draw for (env, tick, site, a, b) */
CYGYM_HD uint32_t cg_draw(uint64_t seed, uint32_t env, uint32_t tick, uint32_t site,
                          uint32_t a, uint32_t b) {
  return cg_philox4x32_10(env, tick, site, (a & 0xFFFFu) | (b << 16),
                          (uint32_t)seed, (uint32_t)(seed >> 32)).v[0];
}

/* index in [0, n): multiply-high (n < 2^32, n > 0) */
CYGYM_HD uint32_t cg_index(uint32_t u, uint32_t n) {
  return (uint32_t)(((uint64_t)u * (uint64_t)n) >> 32);
}
/* random.randint(lo, hi) inclusive */
CYGYM_HD int cg_randint(uint32_t u, int lo, int hi) {
  return lo + (int)cg_index(u, (uint32_t)(hi - lo + 1));
}
/* `random.random() < p`  <=>  u < thr  with thr = ceil(p * 2^32) (u64) */
CYGYM_HD int cg_bernoulli(uint32_t u, uint64_t thr) { return (uint64_t)u < thr; }
/* inverse-CDF lookup on an ascending table of u64 thresholds ceil(cdf_k * 2^32):
 * returns the number of thresholds <= u, i.e. the smallest k with u < thr[k]. */
CYGYM_HD int cg_cdf_lookup(uint32_t u, const uint64_t* thr, int n) {
  int k = 0;
  for (int j = 0; j < n; ++j) k += ((uint64_t)u >= thr[j]) ? 1 : 0;
  return k;
}

#define CG_POISSON_TABLE 16   /* thresholds kept for np.random.poisson       */
#define CG_TRI_TABLE 8        /* thresholds kept for ceil(triangular(0,m,h)) */
#define CG_MAX_EXPLOITS 6     /* CyberDefenseEnv.py:48 (MaxExploits)         */
#define CG_MAX_EVENTS 16      /* evolve events per call (Poisson table cap)  */

#endif /* CYGYM_SPEC_H */
