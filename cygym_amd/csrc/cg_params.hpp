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
  const double* apl;    // [CG_DET_APL_N] leaf-term table of the trained detector (global; tail of the blob), or nullptr
  // global views (host-side convenience; kernels outside the tick use them)
  const uint8_t *dstatic, *vuln, *napps;
  const float *os_val, *version, *anomaly;
  const uint16_t *out_ptr, *out_col;   // u16: E <= 65535, M <= 2048
  const uint16_t *in_ptr, *in_col, *in_eid;
};

struct KParams {
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
  unsigned long long* dbg;   // diagnostic builds (-DCG_STAMPS): [N][16] s_memtime stamps per env
};

// how a kernel sees its parameters: as the by-value kernel argument (per-tick kernel), or through a constant-
// address-space pointer to the kernarg segment itself, re-read at the top of every tick (rollout kernel)
template <bool FUSED> struct KParamsOf { using type = const KParams; };
template <> struct KParamsOf<true> { using type = const __attribute__((address_space(4))) KParams; };

#ifdef CG_STAMPS
#define SUBSTAMP(k) do { if (P.dbg && e.lane == 0) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); P.dbg[(size_t)e.env * 16 + (k)] = _t; } } while (0)
#define SUBVAL(k, v) do { if (P.dbg && e.lane == 0) P.dbg[(size_t)e.env * 16 + (k)] = (unsigned long long)(v); } while (0)
#define STAMP(k) do { if (P.dbg && lane == 0) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); P.dbg[(size_t)env * 16 + (k)] = _t; } } while (0)
#else
#define STAMP(k) do {} while (0)
#define SUBSTAMP(k) do {} while (0)
#define SUBVAL(k, v) do {} while (0)
#endif

#define COLD(c) __builtin_expect(!!(c), 0)   // rarely-taken extra-edge paths: keep them out of the hot layout

#endif  // CG_PARAMS_HPP
