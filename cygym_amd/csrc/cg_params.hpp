// cg_params.hpp -- Device-side parameter blocks (topology blob layout, kernel parameters) and the diagnostic stamp macros.
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_PARAMS_HPP
#define CG_PARAMS_HPP


constexpr int WAVE = 64;

// The shared topology lives in ONE packed device blob whose layout is also the layout of the
// workgroup-shared LDS section (copied with 16-byte loads): byte offsets o_* into the blob.
//   [optr u16 M+1][ocol u16 E][dst u8 M][vul u8 M][nap u8 M][iptr u16 M+1] | [os f32 M][ver f32 M][ano f32 M] | [icol u16 E][ieid u16 E][oeid u16 E]
//   (ieid: out-slot of an in-entry; oeid: in-entry of an out-slot)
// The first `lds_bytes` bytes are staged in LDS: up to and including the float columns (in_lds), or without them
// when that buys more resident waves (choose_launch); the in-CSR columns are read from the L2-resident blob.
struct DevTopo {
  int M, X, E, EW, MC, Mp;
  const uint8_t* blob;
  int o_optr, o_ocol, o_os, o_ver, o_ano, o_dst, o_vul, o_nap, o_iptr, o_icol, o_ieid, o_oeid;
  int blob_bytes, lds_bytes, in_lds, multi;
  int K, KW, x_bytes;   // extra-edge list: capacity, blocked-bit words, bytes of its per-wave LDS section
  int cby_global;       // run-time sizes with M % 4 == 0: the comp_by plane stays in global memory (3 planes staged, Env::cby_g)
  int ct;               // 64 or 256 devices AND no row longer than the device count: the compile-time-size kernels apply (else the run-time ones)
  int lists_global;     // ... and so do the tick's device list, the extra-edge list and the in-row bounds (choose_launch: where that buys a resident wave)
  const double* apl;    // [CG_DET_APL_N] leaf-term table of the trained detector (global; tail of the blob), or nullptr
  // global views (host-side convenience; kernels outside the tick use them)
  const uint8_t *dstatic, *vuln, *napps;
  const float *os_val, *version, *anomaly;
  const uint16_t *out_ptr, *out_col;   // u16: E <= 65535, M <= 2048
  const uint16_t *in_ptr, *in_col, *in_eid;
};

// The fields the tick kernel's PROLOGUE reads (env range, the pointers of every load it issues, the LDS carve-up), gathered
// at the head of the kernel argument: four 64-byte lines instead of fifteen.  The prologue reads them through the plain
// (not laundered) kernarg pointer, so the compiler fetches them in ONE batch of scalar loads at kernel entry; read next
// to their uses through the laundered pointer -- which is right for the ~150 scalars of the tick body, see cg_tick_body.inc
// -- they formed a chain of a dozen dependent scalar-cache round trips, three of them cold misses, in front of the first
// global load (profiles/r04_*: ~3 k of the 6.7 k cycles every env spent staging).  Same member names as KParams, so that
// env_setup() takes either.  Filled by fill_hot() on the host from the complete KParams.
struct KHot {
  struct {
    int M, MC, EW, K, KW, x_bytes, cby_global, multi, lds_bytes, lists_global;
    const uint8_t* blob;
    int o_optr, o_ocol, o_os, o_ver, o_ano, o_dst, o_vul, o_nap, o_iptr, o_icol, o_ieid, o_oeid;
  } t;
  struct { uint64_t seed; int64_t env_id_base; } c;
  struct { uint8_t *live, *stash; uint32_t *blocked, *blocked_in; uint16_t* ring; int32_t* ienv; double* fenv; } b;
  struct { const int32_t *mode, *n_groups, *atype, *n_exploit, *exploit, *app, *dev_cnt; const int16_t* dev_idx; int max_groups, max_devs; } a;
  int env_begin, env_end, wave_lds, shared_lds;
};

struct KParams {
  KHot h;               // FIRST: the prologue reads it at kernarg offset 0
  DevTopo t;
  cygym_config c;
  cygym_buffers b;
  cygym_buffers snap;   // snap.flags == nullptr when absent
  cygym_actions a;
  cygym_outputs o;
  int n_envs;
  int env_begin, env_end;   // envs this launch ticks (cygym_step_range); action / output arrays stay indexed by env id
  int n_ticks;          // ticks per launch (cygym_rollout); actions / outputs are [n_ticks][N] arrays
  int wave_lds;         // bytes of LDS per wave
  int shared_lds;       // bytes of the workgroup-shared LDS section
  unsigned long long* dbg;   // diagnostic builds (-DCG_STAMPS): [N][CG_DBG_W] s_memtime stamps per env
};

inline void fill_hot(KParams& P) {   // host side, once the rest of P is complete
  KHot& h = P.h;
  h.t.M = P.t.M; h.t.MC = P.t.MC; h.t.EW = P.t.EW; h.t.K = P.t.K; h.t.KW = P.t.KW; h.t.x_bytes = P.t.x_bytes;
  h.t.cby_global = P.t.cby_global; h.t.multi = P.t.multi; h.t.lds_bytes = P.t.lds_bytes; h.t.lists_global = P.t.lists_global; h.t.blob = P.t.blob;
  h.t.o_optr = P.t.o_optr; h.t.o_ocol = P.t.o_ocol; h.t.o_os = P.t.o_os; h.t.o_ver = P.t.o_ver; h.t.o_ano = P.t.o_ano;
  h.t.o_dst = P.t.o_dst; h.t.o_vul = P.t.o_vul; h.t.o_nap = P.t.o_nap; h.t.o_iptr = P.t.o_iptr; h.t.o_icol = P.t.o_icol;
  h.t.o_ieid = P.t.o_ieid; h.t.o_oeid = P.t.o_oeid;
  h.c.seed = P.c.seed; h.c.env_id_base = P.c.env_id_base;
  h.b.live = P.b.live; h.b.stash = P.b.stash; h.b.blocked = P.b.blocked; h.b.blocked_in = P.b.blocked_in; h.b.ring = P.b.ring;
  h.b.ienv = P.b.ienv; h.b.fenv = P.b.fenv;
  h.a.mode = P.a.mode; h.a.n_groups = P.a.n_groups; h.a.atype = P.a.atype; h.a.n_exploit = P.a.n_exploit; h.a.exploit = P.a.exploit;
  h.a.app = P.a.app; h.a.dev_cnt = P.a.dev_cnt; h.a.dev_idx = P.a.dev_idx; h.a.max_groups = P.a.max_groups; h.a.max_devs = P.a.max_devs;
  h.env_begin = P.env_begin; h.env_end = P.env_end; h.wave_lds = P.wave_lds; h.shared_lds = P.shared_lds;
}

// how a kernel sees its parameters: as the by-value kernel argument (per-tick kernel), or through a constant-
// address-space pointer to the kernarg segment itself, re-read at the top of every tick (rollout kernel)
template <bool FUSED> struct KParamsOf { using type = const KParams; };
template <> struct KParamsOf<true> { using type = const __attribute__((address_space(4))) KParams; };

#define CG_DBG_W 28   // uint64 slots per env in the stamp buffer of diagnostic builds
#ifdef CG_STAMPS
#define SUBSTAMP(k) do { if (P.dbg && e.lane == 0) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); P.dbg[(size_t)e.env * CG_DBG_W + (k)] = _t; } } while (0)
#define SUBVAL(k, v) do { if (P.dbg && e.lane == 0) P.dbg[(size_t)e.env * CG_DBG_W + (k)] = (unsigned long long)(v); } while (0)
#define STAMP(k) do { if (P.dbg && lane == 0) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); P.dbg[(size_t)env * CG_DBG_W + (k)] = _t; } } while (0)
#else
#define STAMP(k) do {} while (0)
#define SUBSTAMP(k) do {} while (0)
#define SUBVAL(k, v) do {} while (0)
#endif

#define COLD(c) __builtin_expect(!!(c), 0)   // rarely-taken extra-edge paths: keep them out of the hot layout

#endif  // CG_PARAMS_HPP
