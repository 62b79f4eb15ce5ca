// cygym_hip.hip -- MI355X (gfx950, wave64) kernels + C ABI of the batched CyGym tick.
//
// Execution model: ONE WAVEFRONT PER ENVIRONMENT.  A workgroup of WPB waves stages
// the shared topology (out-CSR row pointers + columns, static per-device columns)
// into LDS once, then every wave stages its own env's struct-of-arrays state
// (4 live byte planes + blocked-edge bitmask + log ring) into its private LDS
// region, runs the whole tick there, streams the observation out with 16-byte
// coalesced stores and writes the dirty planes back.  No MFMA: the path is
// integer / bit / index / RNG work bound by HBM traffic.
//
// Parallel restatements of the reference's sequential loops (each checked
// bit-for-bit against oracle/cygym_oracle.c, which keeps the reference's order):
//   * attacker spread (volt_typhoon_env.py:1126-1185): sources are processed
//     "in ascending id order, each seeing what earlier sources compromised".
//     Here: every source picks its target in parallel against a per-device
//     first-compromise time T[v] (atomicMin in LDS); iterate to the unique fix
//     point (== the sequential result; proof sketch in DESIGN.md).  Long rows
//     (attacker-owned hubs) are scanned 64 entries per step with ballot/ffs.
//   * "r-th element of a list in dict order" (random.choice over devices):
//     ballot + popcount ranking.
//   * random.sample(candidates, k) (CDSimulator.py:298): k smallest
//     (philox key, id) by a wave-wide radix select.
//   * per-device defender actions: multiplicity counts via LDS atomics, so
//     duplicate / unsorted device lists give the sequential result.
//
// Reference citations are relative to the reference checkout.
//
// Layout of the sources: this file holds the C ABI (host side) and the small auxiliary kernels; the device code lives
// in the cg_*.hpp files next to it (one file per phase of the tick), gathered by cg_device.hpp in namespace cygym_k.
// The step_kernel variants are compiled in the instantiation units (cg_inst.hip, one object per CG_INST_GROUP) and
// only declared here.
#define CG_MAIN_UNIT 1
#include "cg_device.hpp"
#include "cg_iforest.hpp"
#include <mutex>
#include <set>
#include <utility>
using namespace cygym_k;

// every step_kernel variant lives in one of the instantiation units (cg_inst.hip): declared, not instantiated, here
namespace cygym_k {
#define CG_DECL(W, M, F, X, WD, G) extern template __global__ void step_kernel<W, M, F, X, WD>(const KParams);
CG_STEP_KERNELS(CG_DECL)
#undef CG_DECL
}  // namespace cygym_k

// the tick + actor kernels live in their own unit (cg_inst_actor.hip): declared, not instantiated, here
namespace cygym_k {
#include "cg_tick_actor.hpp"
extern template __global__ void tick_actor_kernel<5>(const KParams, cygym_actor_mlp, cygym_action_vectors, cygym_actions, MlpView);
extern template __global__ void tick_actor_kernel<6>(const KParams, cygym_actor_mlp, cygym_action_vectors, cygym_actions, MlpView);
}  // namespace cygym_k

// =====================================================================
// C ABI
// =====================================================================
struct cygym_handle {
  int device_id;
  int n_envs;
  DevTopo t;
  cygym_config c;
  cygym_buffers b;
  cygym_buffers snap;
  bool bound, has_snap;
  void* dev_blob;       // one allocation holding the topology copies (+ the detector's leaf-term table)
  int wpb, max_devs;
  int wpb_fused;        // shape of the rollout kernels (register-capped at 16 waves per CU: a single 16-wave workgroup where it fits)
  bool few_waves;       // n_envs <= 16 per CU: one wave per env cannot use more than 4 waves per SIMD
  bool wide;            // the WIDE per-tick kernel runs: one 16-wave workgroup per CU with the WHOLE blob (in-CSR maps too) in LDS
  int o_maps_end;       // blob offset just past the in-CSR maps
  int max_row;          // longest out- or in-row of the shared CSR, in slots
  int wave_lds, shared_lds;
  hipEvent_t ev0, ev1;
  unsigned long long* dbg;
  char err[256];
};

static char g_err[256] = "";

static int fail(cygym_handle* h, int code, const char* fmt, const char* detail) {
  char* dst = h ? h->err : g_err;
  snprintf(dst, 256, fmt, detail ? detail : "");
  if (h) snprintf(g_err, 256, "%s", dst);
  return code;
}
#define HIPCHK(h, call)                                                         \
  do {                                                                          \
    hipError_t _e = (call);                                                     \
    if (_e != hipSuccess) return fail(h, CYGYM_EHIP, #call ": %s", hipGetErrorString(_e)); \
  } while (0)

template <int MT, bool FUSED, bool XE, bool WIDE>
static const void* kernel_for(int wpb) {
  if constexpr (!CG_HAS_MT(MT)) return nullptr;
  else if constexpr (WIDE) {   // the WIDE kernel only ever runs as one 16-wave workgroup per CU (choose_launch)
    return wpb == 16 ? (const void*)step_kernel<16, MT, FUSED, XE, WIDE> : nullptr;
  } else
  switch (wpb) {
    case 16: return (const void*)step_kernel<16, MT, FUSED, XE, WIDE>;
    case 8: return (const void*)step_kernel<8, MT, FUSED, XE, WIDE>;
    case 4: return (const void*)step_kernel<4, MT, FUSED, XE, WIDE>;
    case 2: return (const void*)step_kernel<2, MT, FUSED, XE, WIDE>;
    case 1: return (const void*)step_kernel<1, MT, FUSED, XE, WIDE>;
    // run-time sizes only: the shapes in between, for networks whose LDS footprint leaves room for 3, 5, 6 or 12 waves
    // per CU but not the next power of two (2048 devices with an extra-edge list: 3 instead of 2)
    case 12: if constexpr (MT == 0) return (const void*)step_kernel<12, MT, FUSED, XE, WIDE>; else return nullptr;
    case 6: if constexpr (MT == 0) return (const void*)step_kernel<6, MT, FUSED, XE, WIDE>; else return nullptr;
    case 5: if constexpr (MT == 0) return (const void*)step_kernel<5, MT, FUSED, XE, WIDE>; else return nullptr;
    case 3: if constexpr (MT == 0) return (const void*)step_kernel<3, MT, FUSED, XE, WIDE>; else return nullptr;
    default: return nullptr;
  }
}
template <bool FUSED, bool XE, bool WIDE>
static const void* kernel_for_m(const cygym_handle* h) {
  const int wpb = FUSED ? h->wpb_fused : h->wpb;
  if (h->t.ct && h->t.M == 256) return kernel_for<256, FUSED, XE, WIDE>(wpb);
  if (h->t.ct && h->t.M == 64) return kernel_for<64, FUSED, XE, false>(wpb);   // rows of <= 3 words: nothing to gain (measured: -9 %)
  return kernel_for<0, FUSED, XE, false>(wpb);   // run-time M: the wide variant would spill
}
// XE: the kernel that follows the edges evolve_network adds (max_extra_edges > 0).  With no extra-edge list
// the lean instantiation runs: none of that code is in it.
// WIDE (lean per-tick kernel, 256 devices, envs <= 16 per CU: one 16-wave workgroup per CU): the in-CSR columns and slot maps
// ride in LDS as well, the observation leaves through an LDS stage as full 1 KB stores, the spread's log counts read nine words
// at once.  (Its nine-word pool picks are every compile-time-size kernel's since they are arithmetic: cg_env.hpp.)
static bool full_feature(const cygym_handle* h) { return h->t.K > 0 || h->b.forest || h->b.hist || h->b.anomaly; }
static const void* pick_kernel(const cygym_handle* h, bool fused, int full = -1) {
  const bool xe = full < 0 ? full_feature(h) : full != 0;
  if (fused) return xe ? kernel_for_m<true, true, false>(h) : kernel_for_m<true, false, false>(h);
  if (xe) return kernel_for_m<false, true, false>(h);
  return h->wide ? kernel_for_m<false, false, true>(h) : kernel_for_m<false, false, false>(h);
}
static hipError_t set_lds_attr(cygym_handle* h) {   // every instantiation this handle may launch (lean and full-feature)
  for (int full = (h->t.K > 0 ? 1 : 0); full < 2; ++full)
    for (int fused = 0; fused < 2; ++fused) {
      const int lds = h->shared_lds + h->wave_lds * (fused ? h->wpb_fused : h->wpb);
      const void* k = pick_kernel(h, fused != 0, full);
      if (!k) return hipErrorInvalidDeviceFunction;   // development subset build (CG_DEV_MT)
      hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
    }
  return hipSuccess;
}

extern "C" {

int cygym_version(void) { return CYGYM_ABI_VERSION; }
int cygym_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int)sizeof(cygym_topology);
    case 1: return (int)sizeof(cygym_config);
    case 2: return (int)sizeof(cygym_buffers);
    case 3: return (int)sizeof(cygym_actions);
    case 4: return (int)sizeof(cygym_outputs);
    case 5: return (int)sizeof(cygym_action_rows);
    case 6: return (int)sizeof(cygym_action_vectors);
    case 7: return (int)sizeof(cygym_actor_head);
    case 8: return (int)sizeof(cygym_actor_mlp);
    case 9: return (int)sizeof(cygym_device_types);
    case 10: return (int)sizeof(cygym_device_logits);
    default: return -1;
  }
}
const char* cygym_last_error(const cygym_handle* h) { return h ? h->err : g_err; }

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// LDS budget: shared blob prefix + WPB per-wave regions.  Prefers staging the in-CSR too.
// Bytes of the extra-edge section of a wave's LDS block: keys + blocked bits + the per-chunk in / out masks, or -- lists_global --
// the masks alone (the list itself is then read and edited in its global row)
static int x_section_bytes(const DevTopo& t, bool lists_global) {
  if (t.K <= 0) return 0;
  return (int)align_up((lists_global ? 0 : (size_t)4 * (t.K + ((t.KW + 1) & ~1))) + (size_t)16 * t.MC, 16);
}
static size_t wave_lds_bytes(const DevTopo& t, int max_devs) {
  const bool rt = !t.ct;   // run-time size: 4 bytes of scratch per device (16-bit T table), else 6 (env_setup)
  size_t w = align_up((size_t)(t.cby_global ? 3 : 4) * ((t.M + 3) & ~3), 16) + (size_t)t.Mp * (rt ? 4 : 6) + (size_t)((t.EW + 3) & ~3) * 4 * 2 + CG_LOG_RING * 4 +
             (size_t)((t.Mp / 32 + 2) & ~1) * 4 + (size_t)t.MC * 8 + (size_t)t.Mp * 2 +
             (t.lists_global ? 0 : align_up((size_t)max_devs * 2, 16)) + (size_t)x_section_bytes(t, t.lists_global) + 128 /* scalar parking of the fused kernel */ +
             (t.ct && t.M == 64 ? CG_OBS_STAGE_BYTES / 2 : 0) /* the observation's LDS stage at 64 devices (write_obs_staged) */;
  return align_up(w, 16);
}
// The in-CSR columns and slot maps (icol/ieid/oeid, ~2/3 of the blob) are read by block/unblock only (~9 % of
// env-ticks): they stay in global memory (L2-resident); the staged prefix ends before them (o_icol), or already
// before the float columns (o_os).
#ifndef CG_RT_REG_CAP
#define CG_RT_REG_CAP 20
#endif
static int choose_launch_with(cygym_handle* h, int max_devs, int* waves_out);
static int choose_launch(cygym_handle* h, int max_devs) {
  // comp_by in LDS (as ever), or -- run-time sizes with M % 4 == 0 -- in global memory when that frees enough LDS for
  // another resident wave per CU (2048 devices without an extra-edge list: 4 -> 5)
  DevTopo& t = h->t;
  int w_lds = 0, w_glob = 0;
  t.cby_global = 0; t.lists_global = 0;
  const int rc = choose_launch_with(h, max_devs, &w_lds);
  const bool can = !t.ct && (t.M & 3) == 0 && !getenv("CYGYM_CBY_LDS");
  if (can) {
    t.cby_global = 1;
    if (choose_launch_with(h, max_devs, &w_glob) == 0 && (rc != 0 || w_glob > w_lds || getenv("CYGYM_CBY_GLOBAL"))) {   // (env: test aid)
      // ... and, if THAT buys yet another one, the tick's device list, the extra-edge list and the in-row bounds too: they are read
      // where they lie in global memory (2048 devices with a 416-entry extra-edge list: 24.2 -> 22.0 KB per env and 4 KB less of
      // shared topology: 5 -> 6 waves per CU, i.e. 4096 envs in three residency rounds instead of four)
      int w_lists = 0;
      // (the rollout kernels share the plan: 4096 x 2048, 20 ticks per launch: roofline fraction 0.311 -> 0.332 on one box)
      t.lists_global = 1;
      if (!getenv("CYGYM_LISTS_LDS") && choose_launch_with(h, max_devs, &w_lists) == 0 && (w_lists > w_glob || getenv("CYGYM_LISTS_GLOBAL"))) return 0;
      t.lists_global = 0;
      return choose_launch_with(h, max_devs, &w_glob);
    }
    t.cby_global = 0;
    return choose_launch_with(h, max_devs, &w_lds);
  }
  return rc;
}
static int choose_launch_with(cygym_handle* h, int max_devs, int* waves_out) {
  DevTopo& t = h->t;
  const size_t lds_cap = 160 * 1024;
  const size_t wave = wave_lds_bytes(t, max_devs);
  const char* force = getenv("CYGYM_WPB");   // tuning aid: force the waves-per-workgroup choice
  const int forced = force ? atoi(force) : 0;
  int best = 0, best_waves = 0, best_floats = 1;
  // The three static float columns (os / version / anomaly, 12 bytes per device) feed only the observation
  // writer: they ride in LDS unless leaving them in the L2-resident blob buys more resident waves (M >= 1024).
  for (int floats = t.lists_global ? 0 : 1; floats >= 0; --floats) {
    const size_t shared = (size_t)(t.lists_global ? t.o_iptr : floats ? t.o_icol : t.o_os);   // (lists_global: the staged prefix ends before the in-row bounds)
    static const int shapes[] = {16, 12, 8, 6, 5, 4, 3, 2, 1};
    for (int wpb : shapes) {
      if (forced && wpb != forced) continue;
      if ((wpb & (wpb - 1)) != 0 && t.ct) continue;   // the compile-time sizes come in powers of two only
      const size_t per_wg = shared + wave * wpb;
      if (per_wg > lds_cap) continue;
      int waves = (int)(lds_cap / per_wg) * wpb;
      if (waves > 32) waves = 32;
      // ... of which the register file keeps this many resident (whole workgroups): the lean per-tick kernel at a
      // compile-time size is built for 6 waves per SIMD in workgroups of 2-8 waves and 5 otherwise, everything else
      // for 4 (launch bounds of step_kernel).  Without this a 16-wave shape that LDS would hold twice won over three
      // 8-wave workgroups although only one of the two ever runs (16384 x 256: -11 %).
      const bool ct = t.ct != 0, ct_lean = ct && !full_feature(h);
      // (the full-feature per-tick kernels at a compile-time size need <= 102 VGPRs: 5 waves per SIMD;
      // tests/test_host_cpu.py holds them to that)
      // (run-time sizes, per-tick kernels: <= 81 VGPRs since the topology blob is staged by LDS-DMA instead of through registers:
      // five waves per SIMD; tests/test_host_cpu.py holds them to that)
      const int reg_cap = ct_lean ? ((wpb > 1 && wpb <= 8) ? 24 : 20) : (ct ? 20 : CG_RT_REG_CAP);
      if (waves > reg_cap / wpb * wpb) waves = reg_cap / wpb * wpb;
      // ties: two 8-wave workgroups per CU beat one 16-wave workgroup (their phases interleave)
      const bool better = waves > best_waves || (waves == best_waves && floats == best_floats && wpb == 8);
      if (better) { best_waves = waves; best = wpb; best_floats = floats; }
    }
  }
  if (!best) return -1;
  *waves_out = best_waves;
  const size_t shared = (size_t)(t.lists_global ? t.o_iptr : best_floats ? t.o_icol : t.o_os);
  t.x_bytes = x_section_bytes(t, t.lists_global);
  h->wpb = best; h->wave_lds = (int)wave; h->shared_lds = (int)shared;
  // The rollout kernels are built for 4 waves per SIMD whatever the size: 16 resident waves per CU at most, and one
  // 16-wave workgroup measured 4 % faster than two of 8 (16384 x 256).  Otherwise they share the per-tick shape.
  h->wpb_fused = best;
  if (!forced && t.ct && best < 16 && shared + wave * 16 <= lds_cap) h->wpb_fused = 16;
  t.lds_bytes = (int)shared; t.in_lds = best_floats;   // in_lds: the float columns are staged too
  h->max_devs = max_devs;
  // Few envs per CU (<= 16: every env has its own resident wave and a launch lasts as long as its slowest env, which
  // on defender ticks is a block / unblock list): one 16-wave workgroup per CU leaves room for the in-CSR columns and
  // slot maps in LDS as well, so a speculation pass no longer waits on global memory.  Compile-time size 256, lean only.
  h->wide = false;
  // (its nine-word pool reads cover rows of at most 256 slots: max_row is checked here, there is no fallback in the kernel)
  if (h->few_waves && t.ct && t.M == 256 && !full_feature(h) && !forced && h->max_row <= 256 && (size_t)h->o_maps_end + (wave + CG_OBS_STAGE_BYTES) * 16 <= lds_cap) {
    h->wide = true;
    h->wave_lds = (int)wave + CG_OBS_STAGE_BYTES;   // + the observation's LDS stage (write_obs_staged)
    h->wpb = 16; h->wpb_fused = 16; h->shared_lds = h->o_maps_end;
    t.lds_bytes = h->o_maps_end; t.in_lds = 1;
  }
  return 0;
}

int cygym_create(const cygym_topology* topo, const cygym_config* cfg, int32_t n_envs, int32_t device_id,
                 cygym_handle** out) {
  if (!topo || !cfg || !out || n_envs <= 0) return fail(nullptr, CYGYM_EINVAL, "cygym_create: bad argument%s", "");
  const int M = topo->n_devices, E = topo->n_edges, X = topo->n_exploits;
  if (M < 1 || M > 2048) return fail(nullptr, CYGYM_EUNSUPPORTED, "n_devices must be in [1, 2048]%s", "");
  if (E < 0 || E > 65535) return fail(nullptr, CYGYM_EUNSUPPORTED, "n_edges must be <= 65535%s", "");
  if (X < 0 || X > CG_MAX_EXPLOITS) return fail(nullptr, CYGYM_EINVAL, "n_exploits out of range%s", "");
  if (cfg->num_of_device > 5000) return fail(nullptr, CYGYM_EUNSUPPORTED, "numOfDevice > 5000 (ready-set path) is not implemented%s", "");
  // host-side validation of the CSR: a malformed topology must never reach a kernel
  for (int i = 0; i <= M; ++i) {
    if (topo->out_ptr[i] < 0 || topo->out_ptr[i] > E || topo->in_ptr[i] < 0 || topo->in_ptr[i] > E ||
        (i && (topo->out_ptr[i] < topo->out_ptr[i - 1] || topo->in_ptr[i] < topo->in_ptr[i - 1])))
      return fail(nullptr, CYGYM_EINVAL, "malformed CSR row pointers%s", "");
  }
  if (topo->out_ptr[0] != 0 || topo->out_ptr[M] != E || topo->in_ptr[0] != 0 || topo->in_ptr[M] != E)
    return fail(nullptr, CYGYM_EINVAL, "CSR row pointers do not span the edge array%s", "");
  for (int k = 0; k < E; ++k) {
    if (topo->out_col[k] < 0 || topo->out_col[k] >= M || topo->in_col[k] < 0 || topo->in_col[k] >= M ||
        topo->in_eid[k] < 0 || topo->in_eid[k] >= E)
      return fail(nullptr, CYGYM_EINVAL, "CSR column / edge id out of range%s", "");
  }
  {   // in_eid must be a bijection in-entry -> out-slot that agrees with both CSRs
    unsigned char* seen = (unsigned char*)calloc((size_t)(E > 0 ? E : 1), 1);
    if (!seen) return fail(nullptr, CYGYM_EINVAL, "out of host memory%s", "");
    bool ok = true;
    for (int v = 0; v < M && ok; ++v)
      for (int j = topo->in_ptr[v]; j < topo->in_ptr[v + 1] && ok; ++j) {
        const int k = topo->in_eid[j], u = topo->in_col[j];
        ok = !seen[k] && topo->out_col[k] == v && k >= topo->out_ptr[u] && k < topo->out_ptr[u + 1];
        seen[k] = 1;
      }
    free(seen);
    if (!ok) return fail(nullptr, CYGYM_EINVAL, "in_eid does not match the out-CSR%s", "");
  }
  {   // edges evolve_network may add are kept per env and merged into the rows by neighbour id
    const int K = topo->max_extra_edges;
    if (K < 0 || K > 4096 || E + K > 65535) return fail(nullptr, CYGYM_EINVAL, "max_extra_edges must be in [0, 4096] and n_edges + max_extra_edges <= 65535%s", "");
    bool sorted = true;
    for (int u = 0; u < M && sorted && K > 0; ++u) {
      for (int k = topo->out_ptr[u] + 1; k < topo->out_ptr[u + 1] && sorted; ++k) sorted = topo->out_col[k - 1] <= topo->out_col[k];
      for (int k = topo->in_ptr[u] + 1; k < topo->in_ptr[u + 1] && sorted; ++k) sorted = topo->in_col[k - 1] <= topo->in_col[k];
    }
    if (!sorted) return fail(nullptr, CYGYM_EINVAL, "max_extra_edges > 0 needs adjacency rows sorted by neighbour id%s", "");
  }
  cygym_handle* h = new (std::nothrow) cygym_handle();
  if (!h) return fail(nullptr, CYGYM_EINVAL, "out of host memory%s", "");
  memset(h, 0, sizeof(*h));
  h->device_id = device_id; h->n_envs = n_envs; h->c = *cfg;
  hipError_t e0 = hipSetDevice(device_id);
  if (e0 != hipSuccess) { fail(nullptr, CYGYM_EHIP, "hipSetDevice: %s", hipGetErrorString(e0)); delete h; return CYGYM_EHIP; }
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id) != hipSuccess || cus <= 0) cus = 256;
    h->few_waves = (long long)n_envs <= 16LL * cus && !getenv("CYGYM_NO_WIDE");
  }
  DevTopo& t = h->t;
  t.M = M; t.X = X; t.E = E; t.EW = (E + 31) / 32 > 0 ? (E + 31) / 32 : 1;
  t.MC = (M + WAVE - 1) / WAVE; t.Mp = t.MC * WAVE;
  t.K = topo->max_extra_edges; t.KW = (t.K + 31) / 32;
  // Run-time sizes (not one of the compile-time device counts), M % 4 == 0: Device.compromised_by is not staged -- the few
  // actions that touch it go to global memory -- which buys LDS: at 2048 devices 2 KB per env, the difference between four
  // and five resident waves per CU (the lean kernels; with an extra-edge list four either way)
  // (decided in choose_launch: only where it buys a resident wave -- the global-memory accesses cost 3-6 % otherwise)
  t.cby_global = 0;
  t.lists_global = 0;
  t.x_bytes = x_section_bytes(t, false);   // (choose_launch sets the one in effect)
  // one blob, laid out exactly as the LDS-shared section (see DevTopo)
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 16); return (int)o; };
  t.o_optr = take((size_t)(M + 1) * 2); t.o_ocol = take((size_t)(E > 0 ? E : 1) * 2);
  t.o_dst = take(M); t.o_vul = take(M); t.o_nap = take(M);
  t.o_iptr = take((size_t)(M + 1) * 2);   // in-row bounds: always staged (block / unblock and evolve read them per lane)
  t.o_os = take((size_t)M * 4); t.o_ver = take((size_t)M * 4); t.o_ano = take((size_t)M * 4);   // LDS only when that is free
  t.o_icol = take((size_t)(E > 0 ? E : 1) * 2); t.o_ieid = take((size_t)(E > 0 ? E : 1) * 2);
  t.o_oeid = take((size_t)(E > 0 ? E : 1) * 2);
  h->o_maps_end = (int)off;
  {  // env_setup (cg_tick.hpp) derives every section offset from o_dst at the compile-time sizes: hold the layout to that
    const int A_M = (M + 15) & ~15, A_P = (2 * (M + 1) + 15) & ~15, A_F = (4 * M + 15) & ~15, a_e = t.o_dst - A_P;
    if (t.o_optr != 0 || t.o_ocol != A_P || t.o_vul != t.o_dst + A_M || t.o_nap != t.o_dst + 2 * A_M || t.o_iptr != t.o_dst + 3 * A_M ||
        t.o_os != t.o_iptr + A_P || t.o_ver != t.o_os + A_F || t.o_ano != t.o_ver + A_F || t.o_icol != t.o_ano + A_F ||
        t.o_ieid != t.o_icol + a_e || t.o_oeid != t.o_ieid + a_e) { delete h; return fail(nullptr, CYGYM_EINVAL, "internal: blob layout%s", ""); }
  }
  const int o_apl = take(topo->det_apl ? (size_t)CG_DET_APL_N * 8 : 0);   // global only: read by trained scans
  t.blob_bytes = (int)off;
  t.multi = 0;   // duplicate (u,v) out-entries? (env._blocked holds pairs, so duplicates share their state)
  for (int u = 0; u < M && !t.multi; ++u)
    for (int k = topo->out_ptr[u]; k < topo->out_ptr[u + 1] && !t.multi; ++k)
      for (int k2 = k + 1; k2 < topo->out_ptr[u + 1]; ++k2)
        if (topo->out_col[k2] == topo->out_col[k]) { t.multi = 1; break; }
  h->max_row = 0;
  for (int u = 0; u < M; ++u) {
    const int lo = topo->out_ptr[u + 1] - topo->out_ptr[u], li = topo->in_ptr[u + 1] - topo->in_ptr[u];
    if (lo > h->max_row) h->max_row = lo;
    if (li > h->max_row) h->max_row = li;
  }
  // A compile-time size whose longest row exceeds the device count (duplicate edges) runs on the run-time-size kernels: the
  // compile-time ones count and select a row's bits in a fixed number of words (pool_pick<NW>, cg_defender.hpp).
  t.ct = ((M == 64 && h->max_row <= 64) || (M == 256 && h->max_row <= 256)) ? 1 : 0;
  if (choose_launch(h, M > 8 ? M / 8 : 1) != 0) { delete h; return fail(nullptr, CYGYM_EUNSUPPORTED, "topology does not fit in LDS%s", ""); }
  uint8_t* host = (uint8_t*)calloc(1, off);
  if (!host) { delete h; return fail(nullptr, CYGYM_EINVAL, "out of host memory%s", ""); }
  memcpy(host + t.o_dst, topo->dstatic, M); memcpy(host + t.o_vul, topo->vuln, M); memcpy(host + t.o_nap, topo->napps, M);
  for (int u = 0; u < M; ++u) {   // library-private static bit: out-row == every other device, ascending
    host[t.o_dst + u] &= (uint8_t)(CG_D_DC | CG_D_SERVER);
    if (topo->out_ptr[u + 1] - topo->out_ptr[u] != M - 1) continue;
    bool full = true;
    for (int j = 0; j < M - 1 && full; ++j) full = topo->out_col[topo->out_ptr[u] + j] == j + (j >= u ? 1 : 0);
    if (full) host[t.o_dst + u] |= 0x04;
  }
  memcpy(host + t.o_os, topo->os_val, (size_t)M * 4); memcpy(host + t.o_ver, topo->version, (size_t)M * 4);
  memcpy(host + t.o_ano, topo->anomaly, (size_t)M * 4);
  if (topo->det_apl) memcpy(host + o_apl, topo->det_apl, (size_t)CG_DET_APL_N * 8);
  for (int i = 0; i <= M; ++i) { ((uint16_t*)(host + t.o_optr))[i] = (uint16_t)topo->out_ptr[i]; ((uint16_t*)(host + t.o_iptr))[i] = (uint16_t)topo->in_ptr[i]; }
  for (int k = 0; k < E; ++k) {
    ((uint16_t*)(host + t.o_ocol))[k] = (uint16_t)topo->out_col[k];
    ((uint16_t*)(host + t.o_icol))[k] = (uint16_t)topo->in_col[k];
    ((uint16_t*)(host + t.o_ieid))[k] = (uint16_t)topo->in_eid[k];
    ((uint16_t*)(host + t.o_oeid))[topo->in_eid[k]] = (uint16_t)k;   // inverse map: in-entry of every out-slot
  }
  hipError_t e1 = hipMalloc(&h->dev_blob, off);
  if (e1 == hipSuccess) e1 = hipMemcpy(h->dev_blob, host, off, hipMemcpyHostToDevice);
  free(host);
  if (e1 == hipSuccess) e1 = hipEventCreate(&h->ev0);
  if (e1 == hipSuccess) e1 = hipEventCreate(&h->ev1);
  if (e1 != hipSuccess) { fail(nullptr, CYGYM_EHIP, "cygym_create: %s", hipGetErrorString(e1)); cygym_destroy(h); return CYGYM_EHIP; }
  uint8_t* d = (uint8_t*)h->dev_blob;
  t.blob = d;
  t.dstatic = d + t.o_dst; t.vuln = d + t.o_vul; t.napps = d + t.o_nap;
  t.os_val = (const float*)(d + t.o_os); t.version = (const float*)(d + t.o_ver); t.anomaly = (const float*)(d + t.o_ano);
  t.out_ptr = (const uint16_t*)(d + t.o_optr); t.out_col = (const uint16_t*)(d + t.o_ocol);
  t.in_ptr = (const uint16_t*)(d + t.o_iptr); t.in_col = (const uint16_t*)(d + t.o_icol); t.in_eid = (const uint16_t*)(d + t.o_ieid);
  t.apl = topo->det_apl ? (const double*)(d + o_apl) : nullptr;
  // opt in to large dynamic LDS for every instantiation we may launch
  hipError_t e2 = set_lds_attr(h);
  if (e2 != hipSuccess) { fail(nullptr, CYGYM_EHIP, "hipFuncSetAttribute: %s", hipGetErrorString(e2)); cygym_destroy(h); return CYGYM_EHIP; }
  *out = h;
  return CYGYM_OK;
}

void cygym_destroy(cygym_handle* h) {
  if (!h) return;
  if (h->dev_blob) (void)hipFree(h->dev_blob);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  delete h;
}

int cygym_set_config(cygym_handle* h, const cygym_config* cfg) {
  if (!h || !cfg) return fail(h, CYGYM_EINVAL, "cygym_set_config: bad argument%s", "");
  if (!cfg->fast_scan && h->bound && (!h->b.hist || !h->b.anomaly))
    return fail(h, CYGYM_EINVAL, "fast_scan=False (per-log scan path) needs the `hist` and `anomaly` planes bound%s", "");
  h->c = *cfg;
  return CYGYM_OK;
}

static int check_buffers(cygym_handle* h, const cygym_buffers* b, bool snapshot) {
  if (!b || !b->live || !b->stash || !b->blocked || !b->blocked_in || !b->ring || !b->ienv || !b->fenv)
    return fail(h, CYGYM_EINVAL, "buffer struct has a null plane%s", "");
  if (h->t.K > 0 && !b->extra) return fail(h, CYGYM_EINVAL, "max_extra_edges > 0 needs the `extra` plane%s", "");
  if (snapshot ? (b->n_envs != 1 && b->n_envs != h->n_envs) : (b->n_envs != h->n_envs))
    return fail(h, CYGYM_EINVAL, "buffer struct has the wrong leading dimension%s", "");
  return CYGYM_OK;
}

int cygym_bind(cygym_handle* h, const cygym_buffers* state) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_bind: null handle%s", "");
  int rc = check_buffers(h, state, false);
  if (rc) return rc;
  if (!h->c.fast_scan && (!state->hist || !state->anomaly))
    return fail(h, CYGYM_EINVAL, "fast_scan=False (per-log scan path) needs the `hist` and `anomaly` planes bound%s", "");
  const bool was_full = full_feature(h);
  h->b = *state;
  h->bound = true;
  // The launch was planned at cygym_create, before it was known whether a forest / history buffer would select the
  // full-feature kernels (other register budget, no WIDE shape): re-plan now that it is.
  if (full_feature(h) != was_full) {
    if (choose_launch(h, h->max_devs) != 0) return fail(h, CYGYM_EUNSUPPORTED, "topology does not fit in LDS%s", "");
    HIPCHK(h, hipSetDevice(h->device_id));
    HIPCHK(h, set_lds_attr(h));
  }
  return CYGYM_OK;
}

static KParams make_params(cygym_handle* h) {
  KParams P;
  memset(&P, 0, sizeof(P));
  P.t = h->t; P.c = h->c; P.b = h->b; P.n_envs = h->n_envs;
  P.env_begin = 0; P.env_end = h->n_envs;
  P.wave_lds = h->wave_lds; P.shared_lds = h->shared_lds;
  P.dbg = h->dbg;
  return P;
}

int cygym_derive(cygym_handle* h, const cygym_buffers* bufs, void* stream) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_derive: null handle%s", "");
  int rc = check_buffers(h, bufs, true);
  if (rc) return rc;
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(derive_kernel, dim3((bufs->n_envs + waves_per_block - 1) / waves_per_block), dim3(threads), 0,
                     (hipStream_t)stream, P, *bufs);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_reset(cygym_handle* h, const cygym_buffers* snapshot, const int32_t* env_ids, int32_t n, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_reset: handle not bound%s", "");
  if (!snapshot) {
    if (!h->has_snap) return fail(h, CYGYM_EINVAL, "cygym_reset: no snapshot given or registered%s", "");
    snapshot = &h->snap;
  }
  int rc = check_buffers(h, snapshot, true);
  if (rc) return rc;
  if (!env_ids) n = h->n_envs;
  if (n <= 0) return CYGYM_OK;
  if (n > h->n_envs) return fail(h, CYGYM_EINVAL, "cygym_reset: more ids than envs%s", "");
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  P.snap = *snapshot;
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(reset_kernel, dim3((n + waves_per_block - 1) / waves_per_block), dim3(threads), 0,
                     (hipStream_t)stream, P, env_ids, n);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_randomize(cygym_handle* h, const int32_t* env_ids, int32_t n, uint32_t* scratch, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_randomize: handle not bound%s", "");
  if (!scratch) return fail(h, CYGYM_EINVAL, "cygym_randomize: null scratch buffer%s", "");
  if (!env_ids) n = h->n_envs;
  if (n <= 0) return CYGYM_OK;
  if (n > h->n_envs) return fail(h, CYGYM_EINVAL, "cygym_randomize: more ids than envs%s", "");
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(randomize_kernel, dim3((n + waves_per_block - 1) / waves_per_block), dim3(threads), 0,
                     (hipStream_t)stream, P, env_ids, n, scratch);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_set_snapshot(cygym_handle* h, const cygym_buffers* snapshot) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_set_snapshot: null handle%s", "");
  if (!snapshot) { h->has_snap = false; memset(&h->snap, 0, sizeof(h->snap)); return CYGYM_OK; }
  int rc = check_buffers(h, snapshot, true);
  if (rc) return rc;
  h->snap = *snapshot;
  h->has_snap = true;
  return CYGYM_OK;
}

static int launch_ticks(cygym_handle* h, int32_t n_ticks, int32_t env_begin, int32_t n, const cygym_actions* a,
                        const cygym_outputs* o, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_step: handle not bound%s", "");
  if (n_ticks > 1 && (!h->c.fast_scan || h->b.anomaly) && a && o) {
    // The per-log scan path and the per-env anomaly plane it writes live in the per-tick kernels only (the rollout kernels
    // have no registers to spare for them): a rollout of such a handle is issued as n_ticks single-tick launches over
    // the same [n_ticks][N] arrays.
    const size_t N = (size_t)h->n_envs, G = (size_t)(a->max_groups > 0 ? a->max_groups : 1), L = (size_t)(a->max_devs > 0 ? a->max_devs : 1);
    const size_t M = (size_t)h->t.M, WA = 4 * M + (size_t)h->c.max_exploits;
    for (int32_t t = 0; t < n_ticks; ++t) {
      cygym_actions at = *a;
      cygym_outputs ot = *o;
      const size_t r = (size_t)t * N;
      if (at.mode) at.mode += r;
      if (at.n_groups) at.n_groups += r;
      if (at.atype) at.atype += r * G;
      if (at.n_exploit) at.n_exploit += r * G;
      if (at.exploit) at.exploit += r * G * CG_MAX_EXPLOITS;
      if (at.app) at.app += r * G;
      if (at.dev_cnt) at.dev_cnt += r * G;
      if (at.dev_idx) at.dev_idx += r * L;
      if (ot.obs) ot.obs += r * M * 6;
      if (ot.raw) ot.raw += r;
      if (ot.shaped) ot.shaped += r;
      if (ot.done) ot.done += r;
      if (ot.obs_def) ot.obs_def += r * M * 6;
      if (ot.obs_att) ot.obs_att += r * WA;
      const int rc = launch_ticks(h, 1, env_begin, n, &at, &ot, stream);
      if (rc != CYGYM_OK) return rc;
    }
    return CYGYM_OK;
  }
  if (n_ticks < 1) return fail(h, CYGYM_EINVAL, "cygym_rollout: n_ticks must be >= 1%s", "");
  if (env_begin < 0 || n < 0 || env_begin > h->n_envs - n) return fail(h, CYGYM_EINVAL, "cygym_step_range: env range outside [0, n_envs)%s", "");
  if (!a || !o || !a->mode || !a->n_groups || !a->atype || !a->n_exploit || !a->exploit || !a->app ||
      !a->dev_cnt || !a->dev_idx || !o->raw || !o->shaped || !o->done)
    return fail(h, CYGYM_EINVAL, "cygym_step: null action / output pointer%s", "");
  if (a->max_groups < 1 || a->max_devs < 1) return fail(h, CYGYM_EINVAL, "cygym_step: max_groups / max_devs must be >= 1%s", "");
  if (a->max_devs > 32767) return fail(h, CYGYM_EINVAL, "cygym_step: max_devs too large%s", "");
  if (a->max_devs > h->max_devs) {   // the device list lives in LDS: re-plan the launch for a longer list
    if (choose_launch(h, a->max_devs) != 0) return fail(h, CYGYM_EUNSUPPORTED, "device list does not fit in LDS%s", "");
    HIPCHK(h, hipSetDevice(h->device_id));   // (the attribute belongs to the HANDLE's device, whatever the caller's current one is)
    HIPCHK(h, set_lds_attr(h));
  }
  if (h->c.auto_reset && !h->has_snap) return fail(h, CYGYM_EINVAL, "auto_reset needs cygym_set_snapshot first%s", "");
  if (n == 0) return CYGYM_OK;
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  P.a = *a; P.o = *o;
  P.n_ticks = n_ticks;
  P.snap = h->snap;
  P.env_begin = env_begin; P.env_end = env_begin + n;
  // The parameter block travels as the kernel argument only (the rollout kernel re-reads it from the kernarg
  // segment): nothing is uploaded or shared between launches, so launches of one handle on different streams
  // are independent as long as their env ranges are disjoint.  hipGetLastError below reports launch-time errors;
  // a fault inside the kernel surfaces at the caller's next synchronisation.
  const int wpb = n_ticks > 1 ? h->wpb_fused : h->wpb;
  const int lds = h->shared_lds + h->wave_lds * wpb;
  const dim3 grid((n + wpb - 1) / wpb), block(wpb * WAVE);
  hipStream_t s = (hipStream_t)stream;
  {
    fill_hot(P);
    void* args[] = {(void*)&P};
    HIPCHK(h, hipLaunchKernel(pick_kernel(h, n_ticks > 1), grid, block, args, (size_t)lds, s));
  }
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_step(cygym_handle* h, const cygym_actions* a, const cygym_outputs* o, void* stream) {
  return launch_ticks(h, 1, 0, h ? h->n_envs : 0, a, o, stream);
}

int cygym_step_range(cygym_handle* h, int32_t env_begin, int32_t n, const cygym_actions* a, const cygym_outputs* o,
                     void* stream) {
  return launch_ticks(h, 1, env_begin, n, a, o, stream);
}

int cygym_rollout(cygym_handle* h, int32_t n_ticks, const cygym_actions* a, const cygym_outputs* o, void* stream) {
  return launch_ticks(h, n_ticks, 0, h ? h->n_envs : 0, a, o, stream);
}

int cygym_step_actor(cygym_handle* h, const cygym_actions* a, const cygym_outputs* o, const cygym_actor_mlp* mlp,
                     const cygym_action_vectors* layout, const cygym_actions* next, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_step_actor: handle not bound%s", "");
  if (!a || !o || !mlp || !layout || !next) return fail(h, CYGYM_EINVAL, "cygym_step_actor: null argument%s", "");
  if (!a->mode || !a->n_groups || !a->atype || !a->n_exploit || !a->exploit || !a->app || !a->dev_cnt || !a->dev_idx || !o->raw || !o->shaped || !o->done ||
      a->max_groups < 1 || a->max_devs < 1 || a->max_devs > 32767)
    return fail(h, CYGYM_EINVAL, "cygym_step_actor: bad action / output tensors%s", "");
  if (a->max_devs > h->max_devs) {   // the device list lives in LDS: re-plan the launch for a longer list (as cygym_step does)
    if (choose_launch(h, a->max_devs) != 0) return fail(h, CYGYM_EUNSUPPORTED, "device list does not fit in LDS%s", "");
    HIPCHK(h, hipSetDevice(h->device_id));
    HIPCHK(h, set_lds_attr(h));
  }
  // the shape both halves share: the lean WIDE per-tick kernel (256 devices, one 16-wave workgroup per CU = 16 envs) over the whole batch
  if (h->t.M != 256 || !h->wide || h->wpb != 16 || full_feature(h) || (h->n_envs & 15))
    return fail(h, CYGYM_EUNSUPPORTED, "cygym_step_actor: 256 devices, a fixed topology without detector buffers, a multiple of 16 envs and at most 16 envs per CU%s", "");
  if (!next->atype || !next->n_exploit || !next->exploit || !next->app || !next->dev_cnt || !next->dev_idx || next->max_groups < 1 || next->max_devs < 1)
    return fail(h, CYGYM_EINVAL, "cygym_step_actor: bad destination%s", "");
  if (h->c.auto_reset && !h->has_snap) return fail(h, CYGYM_EINVAL, "auto_reset needs cygym_set_snapshot first%s", "");
  if (mlp->obs_role < 1 || mlp->obs_role > 2 || !mlp->w_head || mlp->n_hidden < 1 || mlp->n_hidden > CG_MLP_MAX_HIDDEN ||
      mlp->K != (mlp->obs_role == 1 ? 6 * h->t.M : 4 * h->t.M + h->c.max_exploits))
    return fail(h, CYGYM_EINVAL, "cygym_step_actor: the actor reads the role view built on chip (obs_role 1 / 2, K = 6 M / 4 M + MaxExploits)%s", "");
  for (int l = 0; l < mlp->n_hidden; ++l)
    if (!mlp->w[l] || mlp->width[l] < 16 || mlp->width[l] > 256 || (mlp->width[l] & 15))
      return fail(h, CYGYM_EUNSUPPORTED, "cygym_step_actor: hidden widths must be multiples of 16 up to 256%s", "");
  if (layout->rows || layout->n != h->n_envs || layout->n_devices != h->t.M || layout->n_types < 0 || layout->n_exploits < 0 || layout->n_apps < 0)
    return fail(h, CYGYM_EINVAL, "cygym_step_actor: the actor acts for every env, in env order (rows == NULL, n == n_envs)%s", "");
  if (mlp->n_groups > 1 && (mlp->rows_per_group < 16 || (mlp->rows_per_group & 15)))
    return fail(h, CYGYM_EINVAL, "cygym_step_actor: rows_per_group must be a multiple of 16%s", "");
  const long long n_out = (long long)layout->n_types + layout->n_devices + layout->n_exploits + layout->n_apps;
  const int opl = (int)((n_out + 63) / 64);
  if (opl != 5 && opl != 6) return fail(h, CYGYM_EUNSUPPORTED, "cygym_step_actor: action vectors of 257 to 384 entries%s", "");
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  P.a = *a; P.o = *o;
  P.n_ticks = 1;
  P.snap = h->snap;
  P.env_begin = 0; P.env_end = h->n_envs;
  const MlpPlan pl = mlp_plan(mlp->K, mlp->n_hidden, mlp->width, opl * 64);
  size_t lds = (size_t)pl.total * sizeof(float);
  const size_t lds_tick = (size_t)h->shared_lds + (size_t)h->wave_lds * 16;
  if (lds_tick > lds) lds = lds_tick;
  if (lds > 160 * 1024) return fail(h, CYGYM_EUNSUPPORTED, "cygym_step_actor: the layer shapes do not fit in LDS%s", "");
#if CG_HAS_MT(256)
  const void* k = opl == 5 ? (const void*)tick_actor_kernel<5> : (const void*)tick_actor_kernel<6>;
#else
  const void* k = nullptr;   // development subset build without the 256-device kernels
  if (!k) return fail(h, CYGYM_EUNSUPPORTED, "cygym_step_actor: not in this build%s", "");
#endif
  {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> raised;
    std::lock_guard<std::mutex> lk(mu);
    if (!raised.count({k, h->device_id})) {
      HIPCHK(h, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised.insert({k, h->device_id});
    }
  }
  MlpView view = {h->b.live, h->t.os_val, h->t.version, h->t.anomaly, h->b.anomaly, h->t.M, h->t.X, h->c.max_exploits, mlp->obs_role};
  fill_hot(P);
  void* args[] = {(void*)&P, (void*)mlp, (void*)layout, (void*)next, &view};
  HIPCHK(h, hipLaunchKernel(k, dim3(h->n_envs / 16), dim3(16 * WAVE), args, lds, (hipStream_t)stream));
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_observe(cygym_handle* h, int32_t role, float* out, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_observe: handle not bound%s", "");
  if (!out || role < 0 || role > 2) return fail(h, CYGYM_EINVAL, "cygym_observe: bad argument%s", "");
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(observe_kernel, dim3((h->n_envs + waves_per_block - 1) / waves_per_block), dim3(threads), 0,
                     (hipStream_t)stream, P, role, out);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_gen_actions(cygym_handle* h, int32_t tick, int32_t* mode, int32_t* n_groups, int32_t* atype,
                      int32_t* n_exploit, int32_t* exploit, int32_t* app, int32_t* dev_cnt, int16_t* dev_idx,
                      int32_t max_devs, void* stream) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_gen_actions: null handle%s", "");
  if (!mode || !n_groups || !atype || !n_exploit || !exploit || !app || !dev_cnt || !dev_idx || max_devs < 1)
    return fail(h, CYGYM_EINVAL, "cygym_gen_actions: bad argument%s", "");
  HIPCHK(h, hipSetDevice(h->device_id));
  KParams P = make_params(h);
  hipLaunchKernelGGL(gen_actions_kernel, dim3((h->n_envs + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, tick,
                     mode, n_groups, atype, n_exploit, exploit, app, dev_cnt, dev_idx, max_devs);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_write_actions(cygym_handle* h, const cygym_action_rows* src, const cygym_actions* dst, void* stream) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_write_actions: null handle%s", "");
  if (!src || !dst || !src->atype || !src->exploit || !src->app || (!src->dev_mask && (!src->dev_idx || !src->dev_cnt)))
    return fail(h, CYGYM_EINVAL, "cygym_write_actions: null source pointer%s", "");
  if (!dst->atype || !dst->n_exploit || !dst->exploit || !dst->app || !dst->dev_cnt || !dst->dev_idx || dst->max_groups < 1 ||
      dst->max_devs < 1)
    return fail(h, CYGYM_EINVAL, "cygym_write_actions: bad destination%s", "");
  if (src->n < 0 || (!src->rows && src->n > h->n_envs)) return fail(h, CYGYM_EINVAL, "cygym_write_actions: bad row count%s", "");
  if (src->n == 0) return CYGYM_OK;
  HIPCHK(h, hipSetDevice(h->device_id));
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(write_actions_kernel, dim3((src->n + waves_per_block - 1) / waves_per_block), dim3(threads), 0,
                     (hipStream_t)stream, *src, *dst, h->t.M, h->n_envs);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_decode_actions(cygym_handle* h, const cygym_action_vectors* src, const cygym_actions* dst, void* stream) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_decode_actions: null handle%s", "");
  if (!src || !dst || !src->vec) return fail(h, CYGYM_EINVAL, "cygym_decode_actions: null source pointer%s", "");
  if (!dst->atype || !dst->n_exploit || !dst->exploit || !dst->app || !dst->dev_cnt || !dst->dev_idx || dst->max_groups < 1 ||
      dst->max_devs < 1)
    return fail(h, CYGYM_EINVAL, "cygym_decode_actions: bad destination%s", "");
  if (src->n_types < 0 || src->n_exploits < 0 || src->n_apps < 0 || src->n_devices != h->t.M ||
      (long long)src->stride < (long long)src->n_types + src->n_devices + src->n_exploits + src->n_apps)
    return fail(h, CYGYM_EINVAL, "cygym_decode_actions: row layout does not fit the stride / the handle's device count%s", "");
  if (src->n < 0 || (!src->rows && src->n > h->n_envs)) return fail(h, CYGYM_EINVAL, "cygym_decode_actions: bad row count%s", "");
  if (src->epsilon_thr && !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_decode_actions: epsilon > 0 needs a bound handle%s", "");
  if (src->n == 0) return CYGYM_OK;
  HIPCHK(h, hipSetDevice(h->device_id));
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(decode_actions_kernel, dim3((src->n + waves_per_block - 1) / waves_per_block), dim3(threads), 0,
                     (hipStream_t)stream, *src, *dst, h->n_envs, h->b.ienv, h->c.seed, h->c.env_id_base);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_group_actions(cygym_handle* h, const cygym_device_types* src, const cygym_actions* dst, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_group_actions: handle not bound%s", "");
  if (!src || !dst || !src->types) return fail(h, CYGYM_EINVAL, "cygym_group_actions: null source pointer%s", "");
  if (!dst->n_groups || !dst->atype || !dst->n_exploit || !dst->exploit || !dst->app || !dst->dev_cnt || !dst->dev_idx || dst->max_groups < 1 ||
      dst->max_devs < 1)
    return fail(h, CYGYM_EINVAL, "cygym_group_actions: bad destination%s", "");
  if (src->n_types < 1 || src->n_types > 32 || (!src->visible && src->role != 1 && src->role != 2))
    return fail(h, CYGYM_EINVAL, "cygym_group_actions: 1 to 32 action types; role 1 or 2 when no visibility mask is given%s", "");
  if (src->n < 0 || (!src->rows && src->n > h->n_envs)) return fail(h, CYGYM_EINVAL, "cygym_group_actions: bad row count%s", "");
  if (src->n == 0) return CYGYM_OK;
  HIPCHK(h, hipSetDevice(h->device_id));
  const int threads = 256, waves_per_block = threads / WAVE;
  hipLaunchKernelGGL(group_actions_kernel, dim3((src->n + waves_per_block - 1) / waves_per_block), dim3(threads), 0, (hipStream_t)stream,
                     *src, *dst, h->t.M, h->n_envs, (const uint8_t*)h->b.live, h->b.ienv, h->c.seed, h->c.env_id_base);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_sample_group_actions(cygym_handle* h, const cygym_device_logits* src, const cygym_actions* dst, void* stream) {
  if (!h || !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_sample_group_actions: handle not bound%s", "");
  if (!src || !dst || !src->logits || !src->types_out) return fail(h, CYGYM_EINVAL, "cygym_sample_group_actions: null source pointer%s", "");
  if (!dst->n_groups || !dst->atype || !dst->n_exploit || !dst->exploit || !dst->app || !dst->dev_cnt || !dst->dev_idx || dst->max_groups < 1 ||
      dst->max_devs < 1)
    return fail(h, CYGYM_EINVAL, "cygym_sample_group_actions: bad destination%s", "");
  if (src->n_types < 1 || src->n_types > 32 || src->n_exp < 0 || src->n_exp > 32 || src->n_app < 0 || src->n_app > 32 || (src->role != 1 && src->role != 2))
    return fail(h, CYGYM_EINVAL, "cygym_sample_group_actions: 1 to 32 action types, at most 32 exploit / app logits, role 1 or 2%s", "");
  if (src->n < 0 || (!src->rows && src->n > h->n_envs)) return fail(h, CYGYM_EINVAL, "cygym_sample_group_actions: bad row count%s", "");
  if (src->n == 0) return CYGYM_OK;
  HIPCHK(h, hipSetDevice(h->device_id));
  const size_t lds = (size_t)SAMPLE_WPB * ((size_t)((h->t.M + 63) & ~63) + (size_t)WAVE * src->n_types * sizeof(float));
  hipLaunchKernelGGL(sample_group_actions_kernel, dim3((src->n + SAMPLE_WPB - 1) / SAMPLE_WPB), dim3(SAMPLE_WPB * WAVE), lds, (hipStream_t)stream,
                     *src, *dst, h->t.M, h->n_envs, (const uint8_t*)h->b.live, h->b.ienv, h->c.seed, h->c.env_id_base);
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_actor_head_decode(cygym_handle* h, const cygym_actor_head* head, const cygym_action_vectors* src,
                            const cygym_actions* dst, void* stream) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_actor_head_decode: null handle%s", "");
  if (!head || !src || !dst || !head->hidden || !head->weight_t) return fail(h, CYGYM_EINVAL, "cygym_actor_head_decode: null source pointer%s", "");
  if (!dst->atype || !dst->n_exploit || !dst->exploit || !dst->app || !dst->dev_cnt || !dst->dev_idx || dst->max_groups < 1 ||
      dst->max_devs < 1)
    return fail(h, CYGYM_EINVAL, "cygym_actor_head_decode: bad destination%s", "");
  if (src->n_types < 0 || src->n_exploits < 0 || src->n_apps < 0 || src->n_devices != h->t.M || head->H < 1 || head->hidden_stride < head->H)
    return fail(h, CYGYM_EINVAL, "cygym_actor_head_decode: bad layout%s", "");
  const long long n_out = (long long)src->n_types + src->n_devices + src->n_exploits + src->n_apps;
  if (head->H > 256 || n_out > (long long)HEAD_OPL_MAX * WAVE)
    return fail(h, CYGYM_EUNSUPPORTED, "cygym_actor_head_decode: H > 256 or more than 512 outputs%s", "");
  if (src->n < 0 || (!src->rows && src->n > h->n_envs)) return fail(h, CYGYM_EINVAL, "cygym_actor_head_decode: bad row count%s", "");
  if (src->epsilon_thr && !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_actor_head_decode: epsilon > 0 needs a bound handle%s", "");
  if (src->n == 0) return CYGYM_OK;
  HIPCHK(h, hipSetDevice(h->device_id));
  const int n_out_p = ((int)n_out + 63) & ~63;
  if (head->weight_pitch != n_out_p) return fail(h, CYGYM_EINVAL, "cygym_actor_head_decode: weight_pitch must be n_out rounded up to 64%s", "");
  if (head->n_groups > 1 && (head->rows_per_group < 16 || (head->rows_per_group & 15) || (long long)head->n_groups * head->rows_per_group < src->n))
    return fail(h, CYGYM_EINVAL, "cygym_actor_head_decode: rows_per_group must be a multiple of 16 and the groups must cover the rows%s", "");
  size_t lds = (size_t)n_out_p * HEAD_KC * sizeof(float);
  const void* k = nullptr;
  const bool mfma = (head->H & 3) == 0 && !getenv("CYGYM_HEAD_SCALAR");   // matrix-core variant: 16 rows per workgroup
  if (mfma) {
    lds = ((size_t)16 * n_out_p + (size_t)16 * (head->H + 1)) * sizeof(float);   // outputs + the hidden tile
    switch (n_out_p / WAVE) {
      case 1: k = (const void*)actor_head_mfma_kernel<1>; break;
      case 2: k = (const void*)actor_head_mfma_kernel<2>; break;
      case 3: k = (const void*)actor_head_mfma_kernel<3>; break;
      case 4: k = (const void*)actor_head_mfma_kernel<4>; break;
      case 5: k = (const void*)actor_head_mfma_kernel<5>; break;
      case 6: k = (const void*)actor_head_mfma_kernel<6>; break;
      case 7: k = (const void*)actor_head_mfma_kernel<7>; break;
      default: k = (const void*)actor_head_mfma_kernel<8>; break;
    }
  } else
  switch (n_out_p / WAVE) {   // outputs per lane
    case 1: k = (const void*)actor_head_kernel<1>; break;
    case 2: k = (const void*)actor_head_kernel<2>; break;
    case 3: k = (const void*)actor_head_kernel<3>; break;
    case 4: k = (const void*)actor_head_kernel<4>; break;
    case 5: k = (const void*)actor_head_kernel<5>; break;
    case 6: k = (const void*)actor_head_kernel<6>; break;
    case 7: k = (const void*)actor_head_kernel<7>; break;
    default: k = (const void*)actor_head_kernel<8>; break;
  }
  if (!mfma) HIPCHK(h, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, HEAD_OPL_MAX * WAVE * HEAD_KC * (int)sizeof(float)));
  const int rows_per_wg = 16;   // (both variants: 16 waves, one row each to decode)
  int n_envs = h->n_envs;
  const int32_t* ienv = h->b.ienv;
  uint64_t seed = h->c.seed;
  int64_t base = h->c.env_id_base;
  void* args[] = {(void*)head, (void*)src, (void*)dst, &n_envs, &ienv, &seed, &base};
  HIPCHK(h, hipLaunchKernel(k, dim3((src->n + rows_per_wg - 1) / rows_per_wg), dim3(16 * WAVE), args, lds, (hipStream_t)stream));
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_actor_mlp_decode(cygym_handle* h, const cygym_actor_mlp* mlp, const cygym_action_vectors* src, const cygym_actions* dst,
                           void* stream) {
  if (!h) return fail(h, CYGYM_EINVAL, "cygym_actor_mlp_decode: null handle%s", "");
  if (!mlp || !src || !dst || (!mlp->obs && !mlp->obs_role) || !mlp->w_head) return fail(h, CYGYM_EINVAL, "cygym_actor_mlp_decode: null source pointer%s", "");
  if (!dst->atype || !dst->n_exploit || !dst->exploit || !dst->app || !dst->dev_cnt || !dst->dev_idx || dst->max_groups < 1 ||
      dst->max_devs < 1)
    return fail(h, CYGYM_EINVAL, "cygym_actor_mlp_decode: bad destination%s", "");
  if (src->n_types < 0 || src->n_exploits < 0 || src->n_apps < 0 || src->n_devices != h->t.M || mlp->K < 1 || (!mlp->obs_role && mlp->obs_stride < mlp->K))
    return fail(h, CYGYM_EINVAL, "cygym_actor_mlp_decode: bad layout%s", "");
  if (mlp->obs_role) {
    if (!h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_actor_mlp_decode: obs_role needs a bound handle%s", "");
    if (mlp->obs_role < 1 || mlp->obs_role > 2 || (h->t.M & 1) ||
        mlp->K != (mlp->obs_role == 1 ? 6 * h->t.M : 4 * h->t.M + h->c.max_exploits))
      return fail(h, CYGYM_EUNSUPPORTED, "cygym_actor_mlp_decode: obs_role 1 / 2 with K = 6 M / 4 M + MaxExploits and an even device count%s", "");
  }
  if (mlp->n_hidden < 1 || mlp->n_hidden > CG_MLP_MAX_HIDDEN) return fail(h, CYGYM_EUNSUPPORTED, "cygym_actor_mlp_decode: 1 to 3 hidden layers%s", "");
  for (int l = 0; l < mlp->n_hidden; ++l) {
    if (!mlp->w[l]) return fail(h, CYGYM_EINVAL, "cygym_actor_mlp_decode: null weight pointer%s", "");
    if (mlp->width[l] < 16 || mlp->width[l] > 256 || (mlp->width[l] & 15))
      return fail(h, CYGYM_EUNSUPPORTED, "cygym_actor_mlp_decode: hidden widths must be multiples of 16 up to 256%s", "");
  }
  const long long n_out = (long long)src->n_types + src->n_devices + src->n_exploits + src->n_apps;
  if (n_out > 8192) return fail(h, CYGYM_EUNSUPPORTED, "cygym_actor_mlp_decode: more than 8192 outputs%s", "");
  if (src->n < 0 || (!src->rows && src->n > h->n_envs)) return fail(h, CYGYM_EINVAL, "cygym_actor_mlp_decode: bad row count%s", "");
  if (src->epsilon_thr && !h->bound) return fail(h, CYGYM_ENOTBOUND, "cygym_actor_mlp_decode: epsilon > 0 needs a bound handle%s", "");
  if (mlp->n_groups > 1 && (mlp->rows_per_group < 16 || (mlp->rows_per_group & 15)))   // (row r: actor (r / rows_per_group) % n_groups)
    return fail(h, CYGYM_EINVAL, "cygym_actor_mlp_decode: rows_per_group must be a multiple of 16%s", "");
  if (!mlp->obs_role && (unsigned long long)(mlp->obs_by_env ? h->n_envs : src->n) * (unsigned long long)mlp->obs_stride >= (1ull << 32))
    return fail(h, CYGYM_EUNSUPPORTED, "cygym_actor_mlp_decode: observation matrices of 2^32 floats or more%s", "");   // (32-bit row offsets in the kernel)
  if (src->n == 0) return CYGYM_OK;
  HIPCHK(h, hipSetDevice(h->device_id));
  const bool wide_out = n_out > (long long)HEAD_OPL_MAX * WAVE;   // wider than 512: produced and decoded in chunks of 512 outputs
  const int n_out_p = wide_out ? HEAD_OPL_MAX * WAVE : (((int)n_out + 63) & ~63);
  const MlpPlan pl = mlp_plan(mlp->K, mlp->n_hidden, mlp->width, n_out_p);
  const size_t lds = (size_t)pl.total * sizeof(float);
  if (lds > 160 * 1024) return fail(h, CYGYM_EUNSUPPORTED, "cygym_actor_mlp_decode: the layer shapes do not fit in LDS%s", "");
  // widest vector the observation rows allow (base address and row stride)
  const uintptr_t al = mlp->obs_role ? 0 : ((uintptr_t)mlp->obs | ((uintptr_t)mlp->obs_stride * 4));   // (K itself may be anything <= obs_stride)
  const int vw = mlp->obs_role ? 0 : (al & 15) == 0 ? 4 : (al & 7) == 0 ? 2 : 1;   // (0: the role view built on chip)
  const void* k = nullptr;
#define CG_MLP_CASE(O) case O: k = vw == 0 ? (const void*)actor_mlp_kernel<O, 0> : vw == 4 ? (const void*)actor_mlp_kernel<O, 4> : vw == 2 ? (const void*)actor_mlp_kernel<O, 2> : (const void*)actor_mlp_kernel<O, 1>; break;
  switch (wide_out ? 0 : n_out_p / WAVE) {
    CG_MLP_CASE(0) CG_MLP_CASE(1) CG_MLP_CASE(2) CG_MLP_CASE(3) CG_MLP_CASE(4) CG_MLP_CASE(5) CG_MLP_CASE(6) CG_MLP_CASE(7)
    default: k = vw == 0 ? (const void*)actor_mlp_kernel<8, 0> : vw == 4 ? (const void*)actor_mlp_kernel<8, 4> : vw == 2 ? (const void*)actor_mlp_kernel<8, 2> : (const void*)actor_mlp_kernel<8, 1>; break;
  }
#undef CG_MLP_CASE
  {   // raise the kernel's dynamic-LDS limit once per variant and device (not per launch: this sits in a closed loop's tick)
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> raised;
    std::lock_guard<std::mutex> lk(mu);
    if (!raised.count({k, h->device_id})) {
      HIPCHK(h, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      raised.insert({k, h->device_id});
    }
  }
  int n_envs = h->n_envs;
  const int32_t* ienv = h->b.ienv;
  uint64_t seed = h->c.seed;
  int64_t base = h->c.env_id_base;
  unsigned long long* st = h->dbg;   // (diagnostic builds: cygym_set_debug)
  MlpView view = {h->b.live, h->t.os_val, h->t.version, h->t.anomaly, h->b.anomaly, h->t.M, h->t.X, h->c.max_exploits, mlp->obs_role};
  void* args[] = {(void*)mlp, (void*)src, (void*)dst, &n_envs, &ienv, &seed, &base, &st, &view};
  HIPCHK(h, hipLaunchKernel(k, dim3((src->n + 15) / 16), dim3(MLP_THREADS), args, lds, (hipStream_t)stream));
  HIPCHK(h, hipGetLastError());
  return CYGYM_OK;
}

int cygym_fit_forests(const uint16_t* rows, const int64_t* row_ptr, const uint32_t* seeds, const int32_t* n_fits,
                      const double* sstar, int32_t n, int32_t n_threads, uint32_t* out, uint8_t* failed) {
  if (!rows || !row_ptr || !seeds || !sstar || !out || n < 0) return fail(nullptr, CYGYM_EINVAL, "cygym_fit_forests: bad argument%s", "");
  for (int32_t i = 0; i < n; ++i)
    if (row_ptr[i + 1] <= row_ptr[i] || row_ptr[i + 1] - row_ptr[i] > 65536)
      return fail(nullptr, CYGYM_EINVAL, "cygym_fit_forests: every request needs between 1 and 65536 training rows%s", "");
  if (n == 0) return 0;
  int nt = n_threads < 1 ? 1 : n_threads;
  if (nt > n) nt = n;
  if (nt > 64) nt = 64;
  std::vector<int> bad((size_t)nt, 0);
  auto work = [&](int tid) {   // requests dealt round-robin: neighbours have similar sizes
    for (int32_t i = tid; i < n; i += nt) {
      const int rc = cg_iforest::fit_one(rows + 2 * row_ptr[i], (long)(row_ptr[i + 1] - row_ptr[i]), seeds[i], n_fits ? n_fits[i] : 1,
                                         sstar, out + (size_t)i * CG_FOREST_WORDS);
      if (failed) failed[i] = rc != 0;
      bad[(size_t)tid] += rc != 0;
    }
  };
  if (nt == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back(work, t);
    for (auto& t : th) t.join();
  }
  int total = 0;
  for (int b : bad) total += b;
  return total;
}

/* diagnostic builds only (-DCG_STAMPS): per-env phase stamps, [N][16] uint64 device buffer (NULL to disable) */
int cygym_launch_plan(const cygym_handle* h, int32_t* out) {
  if (!h || !out) return fail(nullptr, CYGYM_EINVAL, "cygym_launch_plan: null argument%s", "");
  out[0] = h->wpb; out[1] = h->wpb_fused; out[2] = h->wave_lds; out[3] = h->shared_lds;
  out[4] = h->t.cby_global; out[5] = h->t.lists_global; out[6] = 0; out[7] = h->wide ? 1 : 0;
  return CYGYM_OK;
}

int cygym_set_debug(cygym_handle* h, void* buf) {
  if (!h) return fail(h, CYGYM_EINVAL, "null handle%s", "");
  h->dbg = (unsigned long long*)buf;
  return CYGYM_OK;
}

int cygym_timer_start(cygym_handle* h, void* stream) {
  if (!h) return fail(h, CYGYM_EINVAL, "null handle%s", "");
  HIPCHK(h, hipEventRecord(h->ev0, (hipStream_t)stream));
  return CYGYM_OK;
}
int cygym_timer_stop(cygym_handle* h, void* stream, float* ms) {
  if (!h || !ms) return fail(h, CYGYM_EINVAL, "null argument%s", "");
  HIPCHK(h, hipEventRecord(h->ev1, (hipStream_t)stream));
  HIPCHK(h, hipEventSynchronize(h->ev1));
  HIPCHK(h, hipEventElapsedTime(ms, h->ev0, h->ev1));
  return CYGYM_OK;
}

}  // extern "C"
