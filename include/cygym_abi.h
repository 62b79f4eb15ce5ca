/* cygym_abi.h -- C ABI of libcygym_hip.so: the batched, MI355X-native tick of
 * CyGym's Volt_Typhoon_CyberDefenseEnv.
 *
 * Plain C across the boundary: pointers, sizes, PODs.  No torch / C++ types.
 * Every device buffer is CALLER-OWNED (the Python host allocates torch tensors
 * and passes tensor.data_ptr()); the library never allocates or frees HBM except
 * for its private copy of the (<= ~100 KB) shared topology and static tables made in
 * cygym_create (launch parameters travel as the kernel argument only; the scratch of
 * cygym_randomize is caller-owned too, see there).
 *
 * Each entry point cites the reference interface it replaces (paths relative to
 * the reference checkout).  The reference has no FFI of its own -- it is 100 %
 * Python -- so "what the reference's FFI would bind" is the method surface of the
 * environment object; INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions
 *   * every function returns 0 on success, a negative CYGYM_E* code otherwise;
 *     cygym_last_error() gives the message; nothing throws or aborts.
 *   * calls on one handle are stream-ordered and asynchronous w.r.t. the host;
 *     they are not thread-safe per handle.  Distinct handles (one per GPU /
 *     process) are independent.
 *   * N = number of envs of this handle (one shard), M = devices per env,
 *     X = exploits, E = directed edges of the shared cached adjacency.
 */
#ifndef CYGYM_ABI_H
#define CYGYM_ABI_H

#include <stdint.h>
#include "cygym_spec.h"

#ifdef __cplusplus
extern "C" {
#endif

#define CYGYM_ABI_VERSION 4

#define CYGYM_OK            0
#define CYGYM_EINVAL       -1  /* bad argument / shape                       */
#define CYGYM_EHIP         -2  /* a HIP runtime call failed                  */
#define CYGYM_EUNSUPPORTED -3  /* configuration outside the implemented path */
#define CYGYM_ENOTBOUND    -4  /* cygym_bind not called                      */

/* Shared topology + static per-device columns (HOST pointers; copied at create).
 * Flattening of Subnet.net / Subnet.graph as cached by
 * volt_typhoon_env.py:456-473 (_outnbrs/_innbrs, neighbour order preserved) and
 * of the static Device/App/Vulnerability/Exploit attributes the tick reads
 * (CDSimulatorComponents.py:120-127, 217-243, 491-531). */
typedef struct cygym_topology {
  int32_t n_devices;        /* M  (= Max_network_size = len(subnet.net))       */
  int32_t n_exploits;       /* X  (= len(simulator.exploits)) <= 6             */
  int32_t n_edges;          /* E                                                */
  int32_t max_extra_edges;  /* K: capacity of the per-env list of edges evolve_network may add (0: additions only
                               raise CG_E_TOPO_OVF); rows must then be sorted by neighbour id      */
  const uint8_t* dstatic;   /* [M] CG_D_DC | CG_D_SERVER                        */
  const uint8_t* vuln;      /* [M] bit e: an app vuln id is in exploits[e].target */
  const uint8_t* napps;     /* [M] len(device.apps)                             */
  const float*   os_val;    /* [M] os_to_float(device.OS)  CyberDefenseEnv.py:125 */
  const float*   version;   /* [M] float(device.version) or -1                  */
  const float*   anomaly;   /* [M] device.anomaly_score at export time, -1 = None (see cygym_buffers.anomaly) */
  const int32_t* out_ptr;   /* [M+1] CSR of _outnbrs                            */
  const int32_t* out_col;   /* [E]                                              */
  const int32_t* in_ptr;    /* [M+1] CSR of _innbrs                             */
  const int32_t* in_col;    /* [E]                                              */
  const int32_t* in_eid;    /* [E] out-CSR slot of each in-entry (blocked bit)  */
  const double*  det_apl;   /* [CG_DET_APL_N] sklearn.ensemble._iforest._average_path_length(n), n = 0..256: the leaf
                               term of IsolationForest's score (CDSimulator.py:683, :721-723), evaluated by the host
                               with numpy so that device and oracle add the very same f64 values.  NULL: a scan in
                               trained-detector mode raises CG_E_UNPINNED                                          */
} cygym_topology;

/* Scalar knobs: plain attributes of the reference env object
 * (volt_typhoon_env.py:32-117, CyberDefenseEnv.py:19-62). */
typedef struct cygym_config {
  uint64_t seed;                 /* Philox key                                  */
  int64_t  env_id_base;          /* global id of env 0 of this shard            */
  int32_t  num_of_device;        /* env.numOfDevice                             */
  int32_t  min_network_size;     /* env.Min_network_size                        */
  int32_t  max_exploits;         /* env.MaxExploits                             */
  int32_t  evolve_period;        /* env._evolve_period                          */
  int32_t  workload_cap;         /* env.workload_cap, -1 = None                 */
  int32_t  workload_period_base; /* env.workload_period_base                    */
  int32_t  workload_period_max;  /* env.workload_period_max                     */
  int32_t  scaling_vulnerability;/* env.scaling_vulnerability                   */
  int32_t  fast_scan;            /* env.fast_scan; 0 = the per-log scan path (volt_typhoon_env.py:1030-1050): needs
                                    cygym_buffers.hist and cygym_buffers.anomaly bound                       */
  int32_t  n_att_actions;        /* env.attacker_action_space.n                 */
  int32_t  n_def_actions;        /* env.defender_action_space.n                 */
  int32_t  zero_day;             /* env.zero_day                                */
  int32_t  zero_day_owned_mask;  /* bit i: i in common|private exploit indices  */
  int32_t  default_high;         /* env.default_high                            */
  int32_t  baseline;             /* 0 Nash, 1 No Defense, 2 Preset, 3 No Attack */
  int32_t  auto_reset;           /* 1: reload snapshot when done (batched only) */
  int32_t  episode_limit;        /* done iff step_num > limit (1000) CyberDefenseEnv.py:549 */
  int32_t  turbo;                /* env.turbo (volt_typhoon_env.py:92): scans skip the detector (:1055), arrivals are
                                    capped and ramped (:219-231), trainings see a clipped, strided log (:165-169) */
  double   work_scale, comp_scale, def_scale, gamma;
  uint64_t p_add_thr;            /* ceil(p_add * 2^32)      CyberDefenseEnv.py:679 */
  uint64_t p_attacker_thr;       /* ceil(p_attacker * 2^32) CyberDefenseEnv.py:690 */
  uint64_t poisson_thr[CG_POISSON_TABLE]; /* np.random.poisson(lambda_events) :668 */
  uint64_t tri_thr[CG_TRI_TABLE];         /* ceil(triangular(0,2,5)) CDSimulator.py:308 */
  /* turbo throttling of _generate_workloads_timed (volt_typhoon_env.py:97-101, :219-231); used only when turbo != 0 */
  double   turbo_fraction_clients, turbo_fraction_servers;
  int32_t  turbo_max_clients, turbo_max_servers, turbo_ramp_steps;
  int32_t  turbo_train_max_logs, turbo_train_stride;   /* host side of Detector.train in turbo mode (:108-109) */
  int32_t  reserved2;
} cygym_config;

/* Per-env mutable state, struct-of-arrays, DEVICE pointers (caller-owned).
 * The four live byte planes of one env are contiguous ([N][4][M]) so that a wave
 * stages a whole env with 16-byte-per-lane loads; plane order:
 *   0 flags   CG_F_*                                   (CDSimulatorComponents.py:217-243)
 *   1 busy    Device.busy_time (saturates at 255)
 *   2 wl      Workload.processing_time, 0 = no workload (CDSimulatorComponents.py:18-26)
 *   3 comp_by bitmask over exploit index               (Device.compromised_by)
 * `stash` has the same shape: the per-device in-memory checkpoint of actions 11/12
 * (volt_typhoon_env.py:419-453), plane 0 = CG_S_VALID | kept flag bits. */
#define CG_P_FLAGS 0
#define CG_P_BUSY 1
#define CG_P_WL 2
#define CG_P_COMPBY 3
#define CG_PLANES 4
typedef struct cygym_buffers {
  uint8_t*  live;       /* [N][4][M]                                            */
  uint8_t*  stash;      /* [N][4][M]                                            */
  uint32_t* blocked;    /* [N][EW] bit per out-CSR slot, EW = ceil(E/32)        */
  uint32_t* blocked_in; /* [N][EW] DERIVED mirror of `blocked` in in-CSR entry order (bit j = blocked[in_eid[j]]),
                           maintained by the library so that every incident-edge pool is two contiguous bit
                           ranges; fill it with cygym_derive() after writing `blocked` from the host */
  uint16_t* ring;       /* [N][CG_LOG_RING][2] last comm-log (from,to) pairs    */
  int32_t*  ienv;       /* [N][CG_I_COUNT]                                      */
  double*   fenv;       /* [N][CG_D_COUNT]                                      */
  uint32_t* extra;      /* [N][CG_X_WORDS(K)] edges added by evolve_network (cygym_spec.h); NULL iff K == 0 */
  uint32_t* forest;     /* [N][CG_FOREST_WORDS] the env's fitted isolation forest (cygym_spec.h), or NULL.  Detector.train
                           (CDSimulator.py:688-695) is a HOST callback: the tick of action 10 records the request in the
                           forest header and sets CG_E_DET_PENDING; the host fits scikit-learn's IsolationForest on the
                           last <= CG_TRAIN_WINDOW entries of `hist`, writes the flattened trees here and clears the bit
                           (cygym_amd/detector.py).  Detector.batch_predict (:721-723) then runs in the tick kernel.   */
  uint16_t* hist;       /* [N][CG_HIST_RING][2] the last 2048 comm-log (from,to) pairs -- what action 10 trains on
                           (volt_typhoon_env.py:955-961) -- or NULL (then only `ring` is kept)                          */
  float*    anomaly;    /* [N][M] Device.anomaly_score per env (-1 = None), or NULL: the topology's static column is
                           the score of every env.  Written only by the per-log scan path (fast_scan = 0), which sets
                           the score of every scanned device to the detector's decision_function of the last log it
                           looked at (volt_typhoon_env.py:1033-1035); read by the observation builders (column 3).   */
  int32_t   n_envs;     /* leading dimension (N, or 1 for a broadcast snapshot) */
  int32_t   reserved;
} cygym_buffers;

/* One tick's actions for every env, DEVICE pointers.
 * The reference's action is (action_type, exploit_indices, device_indices,
 * app_index) or a list of such tuples (volt_typhoon_env.py:818, 842-876). */
typedef struct cygym_actions {
  const int32_t* mode;      /* [N] CG_MODE_*  (env.mode)                        */
  const int32_t* n_groups;  /* [N] 0: step(action); g>0: step_grouped(g groups);
                               <0: this env does not tick (its state and outputs stay) */
  const int32_t* atype;     /* [N][G]                                           */
  const int32_t* n_exploit; /* [N][G]                                           */
  const int32_t* exploit;   /* [N][G][CG_MAX_EXPLOITS]                          */
  const int32_t* app;       /* [N][G] app_index, -1 when not a Python int       */
  const int32_t* dev_cnt;   /* [N][G] len(device_indices)                       */
  const int16_t* dev_idx;   /* [N][L] the groups' device lists, concatenated    */
  int32_t max_groups;       /* G                                                */
  int32_t max_devs;         /* L                                                */
} cygym_actions;

/* What step() returns, for every env, DEVICE pointers. */
typedef struct cygym_outputs {
  float*   obs;     /* [N][M][6] env.state  CyberDefenseEnv.py:146-191, or NULL: not written (a closed-loop
                       consumer that only reads a role view saves the 24 B/device)          */
  double*  raw;     /* [N] raw_reward                                           */
  double*  shaped;  /* [N] shaped_reward                                        */
  uint8_t* done;    /* [N]                                                      */
  /* Optional role views of the state the tick LEAVES BEHIND (after evolve_network) -- what the reference's rollout
   * loops read before the next action: `env._get_defender_state()` / `env._get_attacker_state()`
   * (do_agent.py:212-262, IPPO.py:503-620; CyberDefenseEnv.py:194-257).  NULL: not written.  A closed loop that
   * alternates roles passes obs_att on defender ticks and obs_def on attacker ticks and needs no cygym_observe
   * launch in between.  An env that auto-resets in this tick reports the view of its reloaded state.       */
  float*   obs_def; /* [N][6M]                 _get_defender_state()            */
  float*   obs_att; /* [N][4M + MaxExploits]   _get_attacker_state()            */
  /* Optional episode-return accumulators of a rollout loop (do_agent.py:266-274: `def_total += r` on defender turns,
   * `att_total += r` on attacker turns, `if done: break`): while alive[env] != 0 the tick adds its raw reward to
   * ret[env][mode & 1], and a tick that reports done clears alive[env].  Both NULL: nothing is accumulated.   */
  double*  ret;     /* [N][2] (defender, attacker) reward sums                   */
  uint8_t* alive;   /* [N] 1 until the env's first done                           */
  uint32_t* status; /* optional, ONE word: OR of (CG_E_TOPO_OVF | CG_E_BUSY_SAT | CG_E_DET_PENDING | CG_E_UNPINNED)
                       over the envs this launch ticked, as they stand at write-back (atomically OR-ed in: clear
                       it before the launch).  Lets a driver learn with one 4-byte read whether any env asked for
                       Detector.train in this tick or ran a scan without a current forest.               */
} cygym_outputs;

typedef struct cygym_handle cygym_handle;

int cygym_version(void);
/* sizeof of the ABI structs as this library was compiled (which: 0 cygym_topology, 1 cygym_config, 2 cygym_buffers,
 * 3 cygym_actions, 4 cygym_outputs, 5 cygym_action_rows, 6 cygym_action_vectors, 7 cygym_actor_head, 8 cygym_actor_mlp, 9 cygym_device_types, 10 cygym_device_logits; -1 for anything else): lets a
 * binding check its own struct layouts at load time. */
int cygym_sizeof(int32_t which);
const char* cygym_last_error(const cygym_handle* h);  /* h may be NULL */

/* Replaces: building the env object's caches after initialize_environment()
 * (volt_typhoon_env.py:1485, :456) -- ingests the RESULT, flattened. */
int cygym_create(const cygym_topology* topo, const cygym_config* cfg, int32_t n_envs,
                 int32_t device_id, cygym_handle** out);
void cygym_destroy(cygym_handle* h);
/* attribute writes on the env object (env.base_line = ..., env.comp_scale = ...) */
int cygym_set_config(cygym_handle* h, const cygym_config* cfg);
int cygym_bind(cygym_handle* h, const cygym_buffers* state);

/* Recompute the derived members of `bufs` (blocked_in) from the canonical ones, for all
 * bufs->n_envs envs.  Call after loading state / snapshots from the host. */
int cygym_derive(cygym_handle* h, const cygym_buffers* bufs, void* stream);

/* Replaces: reset(from_init=True) volt_typhoon_env.py:1904-1936 (restore the
 * pickled initial env).  `snapshot` has n_envs == 1 (broadcast) or N.
 * env_ids: DEVICE int32[n] or NULL for all envs. */
int cygym_reset(cygym_handle* h, const cygym_buffers* snapshot, const int32_t* env_ids,
                int32_t n, void* stream);

/* Registers the initial-state snapshot used by cygym_reset(snapshot == NULL) and by
 * config.auto_reset (episode end inside cygym_step).  Pass NULL to clear. */
int cygym_set_snapshot(cygym_handle* h, const cygym_buffers* snapshot);

/* Replaces: randomize_compromise_and_ownership() volt_typhoon_env.py:330-383.
 * scratch: DEVICE uint32 [n][ceil(M/64)*64] owned by the caller (the shuffle keys of the n envs). */
int cygym_randomize(cygym_handle* h, const int32_t* env_ids, int32_t n, uint32_t* scratch, void* stream);

/* Replaces: step(action) volt_typhoon_env.py:818-1333 and
 * step_grouped(groups) :694-779, incl. evolve_network CyberDefenseEnv.py:583-875,
 * arrivals :575-596 / CDSimulator.py:244-348, logger/detector CDSimulator.py:663-742. */
int cygym_step(cygym_handle* h, const cygym_actions* a, const cygym_outputs* o, void* stream);

/* cygym_step over the envs [env_begin, env_begin + n) only: the arrays of `a` and `o` are still indexed by env id
 * (full [N] leading dimension).  Lets a closed-loop driver pipeline sub-batches on several streams -- one
 * sub-batch's policy evaluation and the tail of its slowest env overlap the other sub-batches' ticks -- the
 * batched counterpart of the reference's process-per-rollout fan-out (do_agent.py:1928-1942).  Calls on one
 * handle that run concurrently on different streams must cover disjoint env ranges. */
int cygym_step_range(cygym_handle* h, int32_t env_begin, int32_t n, const cygym_actions* a, const cygym_outputs* o,
                     void* stream);

/* n_ticks consecutive ticks in ONE launch: every array of `a` and `o` has a leading tick
 * dimension ([n_ticks][N]...), the actions of all ticks are staged beforehand (open-loop
 * policies: baselines, fixed action sequences `strat.actions[t % len]` do_agent.py:2052, or
 * any pre-computed script).  Same results as n_ticks calls of cygym_step, but an env's state
 * stays on chip between its ticks and envs do not wait for each other between ticks. */
int cygym_rollout(cygym_handle* h, int32_t n_ticks, const cygym_actions* a, const cygym_outputs* o,
                  void* stream);

/* Replaces: _get_state / _get_defender_state / _get_attacker_state
 * CyberDefenseEnv.py:146-257.  role 0: full [N][6M]; 1: defender [N][6M];
 * 2: attacker [N][4M + MaxExploits].  out: DEVICE float32. */
int cygym_observe(cygym_handle* h, int32_t role, float* out, void* stream);

/* One strategy's chosen actions for n envs (single-action form), DEVICE pointers: what the reference's rollout loop
 * builds per env as the tuple (action_type, [exploit], device_indices, app_index) from its actor's output
 * (do_agent.decode_action, do_agent.py:253-262). */
typedef struct cygym_action_rows {
  const int32_t* rows;     /* [n] env ids (rows of the action tensors) to write; NULL = rows 0..n-1             */
  const int32_t* atype;    /* [n]                                                                              */
  const int32_t* exploit;  /* [n] one exploit index, -1 = none                                                 */
  const int32_t* app;      /* [n] app_index                                                                    */
  const uint8_t* dev_mask; /* [n][M] non-zero = device chosen: compacted to the ascending id list (first max_devs
                              of them), or NULL: take dev_idx / dev_cnt                                        */
  const int16_t* dev_idx;  /* [n][max_devs] device lists (when dev_mask == NULL)                               */
  const int32_t* dev_cnt;  /* [n]                                                                              */
  int32_t n;
  int32_t reserved;
} cygym_action_rows;

/* Scatter `src` into group 0 of the rows src->rows of the action tensors `dst` (atype, n_exploit, exploit[.][0][0],
 * app, dev_cnt, dev_idx -- list entries past the count are zeroed; mode and n_groups are not touched): ONE launch
 * per strategy of a closed loop instead of one tensor op per field.  Replaces the per-env action-tuple assembly of
 * the reference's loop (do_agent.py:206-265) for a batch. */
int cygym_write_actions(cygym_handle* h, const cygym_action_rows* src, const cygym_actions* dst, void* stream);

/* Actor outputs of n envs, DEVICE pointers: one row per env laid out as the reference's DDPG / actor-critic policies
 * emit it (do_agent.py:1016-1020): [n_types action-type logits | n_devices device values | n_exploits exploit values |
 * n_apps app values]. */
typedef struct cygym_action_vectors {
  const int32_t* rows;     /* [n] env ids (rows of the action tensors) to write; NULL = rows 0..n-1             */
  const float*   vec;      /* [n][stride]                                                                      */
  const int32_t* type_map; /* optional [n_types]: the action type each logit stands for (NULL: its index)      */
  int32_t stride;          /* floats per row, >= n_types + n_devices + n_exploits + n_apps                     */
  int32_t n_types, n_devices, n_exploits, n_apps;
  int32_t n;
  uint64_t epsilon_thr;    /* ceil(epsilon * 2^32): with probability epsilon the action type is a uniformly random one
                              instead of the argmax (the epsilon-greedy of decode_action, do_agent.py:972-973), drawn
                              with the Philox draw addressed (env, the env's current rng tick, CG_SITE_EPS_TYPE); 0 = greedy.
                              Needs a bound handle (the rng ticks are read from cygym_buffers.ienv)                */
  uint32_t* status;        /* optional, ONE word: CG_DECODE_TRUNCATED is OR-ed in when a row chose more devices than
                              the action tensors' max_devs holds (the list is cut to the first max_devs ids)    */
} cygym_action_vectors;
#define CG_DECODE_TRUNCATED 0x10000u

/* Replaces: DoubleOracle.decode_action (do_agent.py:935-998, the plain branch :970-998) for a batch, fused with the
 * scatter into the action tensors: action_type = argmax of the type logits (first maximum, like np.argmax),
 * device_indices = ascending ids whose value is > 0, exploit_indices = [argmax of the exploit values] ([0] when
 * n_exploits == 0), app_index = argmax of the app values (0 when n_apps == 0).  Values must be finite.  Writes group 0
 * of the rows like cygym_write_actions; n_devices must equal the handle's device count. */
int cygym_decode_actions(cygym_handle* h, const cygym_action_vectors* src, const cygym_actions* dst, void* stream);

/* The LAST layer of an actor network fused with cygym_decode_actions: `vec` of cygym_action_vectors is not read;
 * row r of the action vectors is  act(hidden[r] x weight_t + bias)  with weight_t [H][pitch] (k-major: torch's
 * nn.Linear.weight transposed), n_out = n_types + n_devices + n_exploits + n_apps, act = tanh when `tanh_out` (the reference's
 * actor ends in tanh, do_agent.py:370) else identity -- computed in fp32 in the kernel (k ascending per output) and
 * decoded from registers, so the [n][n_out] action vectors never touch HBM.  Limits: H <= 256, n_out <= 512
 * (CYGYM_EUNSUPPORTED otherwise: run the layer yourself and call cygym_decode_actions). */
typedef struct cygym_actor_head {
  const float* hidden;     /* [n][hidden_stride] activations of the actor's last hidden layer                   */
  const float* weight_t;   /* [H][weight_pitch], the first n_out entries of a row are used                       */
  const float* bias;       /* [n_out] or NULL                                                                  */
  int32_t H, hidden_stride;
  int32_t tanh_out;
  int32_t weight_pitch;    /* floats per row of weight_t: n_out rounded up to a multiple of 64 (rows 16-byte aligned)  */
  /* Several actors of ONE architecture in one launch (a population of strategies): source row r belongs to actor
   * r / rows_per_group and is multiplied with that actor's matrix, weight_t + (r / rows_per_group) * H * weight_pitch, and
   * bias + (r / rows_per_group) * n_out.  rows_per_group must be a multiple of 16; n_groups <= 1: one actor for all rows. */
  int32_t n_groups, rows_per_group;
} cygym_actor_head;
int cygym_actor_head_decode(cygym_handle* h, const cygym_actor_head* head, const cygym_action_vectors* layout,
                            const cygym_actions* dst, void* stream);

/* Per-device action types of n envs, DEVICE pointers: what the reference's IPPO / MAPPO / HMARL policies sample per decision
 * (IPPO.py:526-557: one Categorical per device over the role's action types, one exploit index, one app index). */
typedef struct cygym_device_types {
  const int32_t* rows;     /* [n] env ids (rows of the action tensors) to write; NULL = rows 0..n-1             */
  const uint8_t* types;    /* [n][M] sampled action type of every device, 0 .. n_types - 1                      */
  const uint8_t* visible;  /* [n][M] non-zero = the device takes part, or NULL: the role's visibility mask computed
                              from the handle's bound flag plane (build_visibility_mask, IPPO.py:74-96: defender =
                              attacker_owned and not Not_yet_added; attacker = that and Known_to_attacker)      */
  const int32_t* exploit;  /* [n] exploit index, or NULL = 0                                                    */
  const int32_t* app;      /* [n] app index, or NULL = 0                                                        */
  int32_t n, n_types;
  int32_t noop;            /* the role's no-op type (DEFENDER_NOOP = 8, ATTACKER_NOOP = 3): never a group        */
  int32_t role;            /* 1 defender, 2 attacker (only read when visible == NULL)                           */
  uint32_t single_mask;    /* bit t set: type t acts on ONE device -- a uniformly random one of its devices
                              (SINGLE_DEVICE_TYPES = {11, 12}, IPPO.py:27, :566-567), drawn with the Philox draw
                              addressed (env, the env's current rng tick, CG_SITE_GROUP_PICK, t)                 */
  int32_t reserved;
  uint32_t* status;        /* optional, ONE word: CG_DECODE_TRUNCATED is OR-ed in when a row needs more groups than
                              max_groups or more list entries than max_devs (the groups are cut there)          */
} cygym_device_types;

/* Replaces: the grouping of per-device decisions into env.step(groups) (IPPO.py:560-572, MAPPO.py the same) for a
 * batch: for every action type t in ascending order except the no-op, the visible devices that sampled t form the group
 * (t, [exploit], ascending device ids, app) -- one device for a single-device type -- and a row without any group steps
 * [(noop, [0], [], 0)].  Writes n_groups and the groups' atype / n_exploit (= 1) / exploit[.][0] / app / dev_cnt and the
 * concatenated device lists of the rows (mode is not touched).  n_types <= 32; needs a bound handle. */
int cygym_group_actions(cygym_handle* h, const cygym_device_types* src, const cygym_actions* dst, void* stream);

/* Per-device action-type LOGITS of n envs, DEVICE pointers: the outputs of the reference's per-device actor-critic networks
 * (IPPO.py:517-519: out["per_dev_type_logits"] [1, D, K], out["exp_logits"], out["app_logits"]) for a batch. */
typedef struct cygym_device_logits {
  const int32_t* rows;       /* [n] env ids (rows of the action tensors) to write; NULL = rows 0..n-1           */
  const float* logits;       /* [n][M][n_types]                                                                 */
  const float* exp_logits;   /* [n][n_exp] or NULL (exploit index 0)                                            */
  const float* app_logits;   /* [n][n_app] or NULL (app index 0)                                                */
  uint8_t* types_out;        /* [n][M] the sampled type of every device, 0 where invisible (Step.per_dev_types) */
  int32_t* exp_out;          /* [n] or NULL                                                                     */
  int32_t* app_out;          /* [n] or NULL                                                                     */
  float*   logp_out;         /* [n] sum of the log-probabilities of the samples: visible devices + exploit + app */
  int32_t n, n_types, n_exp, n_app;   /* n_types, n_exp, n_app <= 32                                             */
  int32_t noop, role;        /* as in cygym_device_types; the visibility mask is the role's (flag plane)         */
  uint32_t single_mask;
  int32_t greedy;            /* non-zero: arg-max (first maximum) instead of a sample                           */
  uint32_t* status;          /* optional, ONE word: CG_DECODE_TRUNCATED as in cygym_device_types                 */
} cygym_device_logits;

/* Replaces: the sampling of one Categorical per visible device, of the exploit and of the app, their summed log-probability
 * (IPPO.py:524-557) AND the grouping into env.step(groups) (:560-572) for a batch, in ONE launch: sample k of a head with
 * logits l is the first k with  sum_{j <= k} exp(l_j - max l) > u * sum_j exp(l_j - max l),  u from the Philox draw addressed
 * (env, the env's current rng tick, CG_SITE_SAMPLE, device / head); log-probability l_k - max l - log(sum).  Then exactly
 * cygym_group_actions on the sampled types.  Needs a bound handle. */
int cygym_sample_group_actions(cygym_handle* h, const cygym_device_logits* src, const cygym_actions* dst, void* stream);

/* The WHOLE actor network of the reference's policies (do_agent.py:357-370: Linear-ReLU stacks ending in a Linear layer,
 * optionally tanh) fused with cygym_decode_actions -- ONE launch per acting role and tick of a closed loop: a workgroup owns
 * 16 observation rows, stages them through LDS, runs every layer on the matrix cores (fp32 in, fp32 accumulate; hidden
 * activations stay in LDS) and decodes the rows from registers like cygym_actor_head_decode.  Neither the hidden
 * activations nor the action vectors touch HBM.
 *   layer l (0 <= l < n_hidden):  x <- relu(x W_l^T + b_l),  width[l] outputs (a multiple of 16, <= 256)
 *   head:                         v  = act(x W_head^T + b_head),  n_out = n_types + n_devices + n_exploits + n_apps <= 8192
 *                                 (vectors wider than 512 -- more than ~490 devices -- are produced and decoded in chunks of 512)
 * Weights are PACKED in the order the matrix-core fragments read them (cygym_amd.batched_env.pack_linear): for a layer
 * with K inputs and N outputs, [ceil(N / 16)][ceil(K / 16)][64][4] floats with
 *   packed[t][g][lane][i] = W[16 t + lane % 16][16 g + 4 (lane / 16) + i]      (W = nn.Linear.weight [N][K]; 0 outside)
 * -- the head's N rounded up to a multiple of 64.  K of layer 0 is the observation width; K of layer l is width[l-1]. */
#define CG_MLP_MAX_HIDDEN 3
typedef struct cygym_actor_mlp {
  const float* obs;        /* [n][obs_stride] role observations (or, with obs_by_env, [n_envs][obs_stride] indexed by env id) */
  const float* w[CG_MLP_MAX_HIDDEN];   /* packed weights of the hidden layers                                            */
  const float* b[CG_MLP_MAX_HIDDEN];   /* [width[l]] or NULL                                                             */
  const float* w_head;     /* packed weights of the last layer                                                           */
  const float* b_head;     /* [n_out] or NULL                                                                            */
  int32_t obs_stride, K;   /* floats per observation row; observation width (K <= obs_stride)                            */
  int32_t n_hidden;        /* 1 .. CG_MLP_MAX_HIDDEN                                                                     */
  int32_t width[CG_MLP_MAX_HIDDEN];
  int32_t tanh_out;
  int32_t obs_by_env;      /* 0: observation of source row r is obs[r];  1: obs[rows[r]] (rows of cygym_action_vectors:
                              the policy reads the batch's role-view tensor in place, no gather)                         */
  int32_t obs_role;        /* 0: read `obs`.  1 / 2: `obs` is not read -- the defender / attacker view of env rows[r] (or r) is built
                              on chip from the handle's BOUND state (flag plane + static columns; CyberDefenseEnv.py:194-257), 256
                              bytes per env instead of a 6 KB view, and the tick need not write role views at all.  K must be
                              6 M resp. 4 M + MaxExploits, M even.                                                        */
  int32_t reserved;
  /* A population of same-shaped actors in one launch, as in cygym_actor_head: source row r belongs to actor
   * (r / rows_per_group) % n_groups; every packed matrix / bias of actor a follows that of actor a - 1 contiguously. */
  int32_t n_groups, rows_per_group;
} cygym_actor_mlp;
int cygym_actor_mlp_decode(cygym_handle* h, const cygym_actor_mlp* mlp, const cygym_action_vectors* layout,
                           const cygym_actions* dst, void* stream);

/* cygym_step and the NEXT acting role's cygym_actor_mlp_decode as ONE launch -- a whole turn of a closed loop
 * (do_agent.py:206-272: act on the observation, step) per launch instead of two.  Tick the whole batch with the actions `a`
 * (outputs `o`, as cygym_step), then, in the same workgroups, build the role view mlp->obs_role of the state the tick left
 * behind, run the actor on it and write every env's next action into `next` (which may alias `a`: a workgroup reads its
 * envs' actions before it writes them).  `layout`: rows == NULL, n == n_envs; with n_groups > 1 env e belongs to actor
 * (e / rows_per_group) % n_groups (the grid layouts of cygym_amd/rollout_grid: defender strategies vary slowest, attacker
 * strategies next, Monte-Carlo repeats fastest).  Only where both kernels share their launch shape: 256 devices, a fixed
 * topology (no extra-edge list, no detector buffers), a multiple of 16 envs and at most 16 envs per CU, action vectors of
 * 257..384 entries -- CYGYM_EUNSUPPORTED otherwise (use the two calls). */
int cygym_step_actor(cygym_handle* h, const cygym_actions* a, const cygym_outputs* o, const cygym_actor_mlp* mlp,
                     const cygym_action_vectors* layout, const cygym_actions* next, void* stream);

/* Replaces: Detector.train(logs) for a batch (CDSimulator.py:688-695: IsolationForest(n_estimators=2, max_samples=256).fit
 * on the [from_device, to_device] pairs of the last <= 2000 log entries, volt_typhoon_env.py:955-961) -- HOST memory in,
 * HOST memory out, no GPU work: the natively restated estimator (csrc/cg_iforest.hpp) on `n_threads` host threads.
 *   rows     [row_ptr[n]][2] uint16: the training rows of the n requests, concatenated (request i: rows row_ptr[i] ..
 *            row_ptr[i+1]); every request needs at least one row
 *   seeds    [n] the 32-bit seed of the numpy stream each fit draws from (cygym_amd/detector.fit_seed)
 *   n_fits   [n] consecutive fits on the same rows from that one stream (several action-10 groups in one step_grouped
 *            tick refit on the same logs: the last forest stays), or NULL = 1 each
 *   sstar    [257] f64: decision threshold S* by max_samples_ (cygym_spec.h forest header; detector.score_threshold)
 *   out      [n][CG_FOREST_WORDS] flattened forests (header words 0-2 and 7 filled, 3-6 zero)
 *   failed   [n] set to 1 where a forest does not fit the flat layout (the caller falls back to scikit-learn), or NULL
 * Returns the number of failed requests (>= 0), or a negative CYGYM_E* code. */
int cygym_fit_forests(const uint16_t* rows, const int64_t* row_ptr, const uint32_t* seeds, const int32_t* n_fits,
                      const double* sstar, int32_t n, int32_t n_threads, uint32_t* out, uint8_t* failed);

/* Synthetic action script of bench.py -- not a reference
 * interface: fills one tick's cygym_actions from Philox on device. */
int cygym_gen_actions(cygym_handle* h, int32_t tick, int32_t* mode, int32_t* n_groups,
                      int32_t* atype, int32_t* n_exploit, int32_t* exploit, int32_t* app,
                      int32_t* dev_cnt, int16_t* dev_idx, int32_t max_devs, void* stream);

/* Timing aid: run `fn`-less HIP-event bracket on a stream.  Returns milliseconds
 * between two events recorded around the work enqueued by the caller in between:
 *   cygym_timer_start(h, stream); ...launches...; cygym_timer_stop(h, stream, &ms) */
int cygym_timer_start(cygym_handle* h, void* stream);
int cygym_timer_stop(cygym_handle* h, void* stream, float* ms);

/* Introspection: how this handle's tick kernels are launched (what cygym_create / a longer device list planned from the
 * LDS and register budgets; no reference counterpart -- the reference has no launch).  out[8] =
 *   {waves per workgroup of cygym_step, ... of cygym_rollout, LDS bytes per wave, LDS bytes of the shared topology section,
 *    comp_by plane in global memory (0/1), device / extra-edge lists + in-row bounds in global memory (0/1),
 *    reserved (0), one 16-wave workgroup per CU with the in-CSR maps in LDS (0/1)} */
int cygym_launch_plan(const cygym_handle* h, int32_t* out);

/* Diagnostic builds only (-DCG_STAMPS, tools/stamps.py): `stamps` = DEVICE int64 [N][16] receiving per-phase
 * s_memtime stamps of every env's last tick, or NULL to switch them off.  Ignored by the product build. */
int cygym_set_debug(cygym_handle* h, void* stamps);

#ifdef __cplusplus
}
#endif
#endif /* CYGYM_ABI_H */
