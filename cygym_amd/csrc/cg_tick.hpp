// cg_tick.hpp -- The tick kernel: step_kernel<WPB, MT, FUSED, XE> (volt_typhoon_env.py:818-1333, 612-779).
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_TICK_HPP
#define CG_TICK_HPP

// ---------------- the tick ----------------
// MT: devices per env when known at compile time (64, 256: chunk loops unroll and their LDS latencies
// overlap), 0 = any M at run time.
// XE: the FULL-FEATURE instantiation -- follows the edges evolve_network adds (extra-edge list), keeps the long
// comm-log history (cygym_buffers.hist) and walks the trained detector's forest (cygym_buffers.forest).  The lean
// instantiation (none of those buffers bound) carries none of that code.
// Per-wave LDS carve + pointer table of one env (must match wave_lds_bytes on the host).
// One lane per device PAIR: 12 floats = three 16-byte stores; static columns read as float2.
// anomaly column of an env that owns a per-env plane (cygym_buffers.anomaly): global memory this wave may have written
// in this very tick (slow scan path) -- read around the vector L1
__device__ __forceinline__ float2 ld_ano2(const float2* p, bool vol) {
  if (!vol) return *p;
  const volatile float* q = (const volatile float*)p;
  return make_float2(q[0], q[1]);
}
__device__ __forceinline__ float ld_ano(const float* p, bool vol) { return vol ? *(const volatile float*)p : *p; }
template <int GP>   // pairs per lane and step
__device__ __forceinline__ void write_obs(const uint8_t* flags, const float* osv, const float* ver, const float* ano,
                                          float* obs, int M, int lane, bool vol = false) {
  if (!(M & 1)) {
    float4* out4 = (float4*)obs;
    const int npairs = M >> 1;
    const uint16_t* F2 = (const uint16_t*)flags;
    const float2* os2 = (const float2*)osv;
    const float2* ve2 = (const float2*)ver;
    const float2* an2 = (const float2*)ano;
    // GP pairs per step, loads first: one round trip per step for the static columns (L2 at the sizes that keep
    // them out of LDS), not one per pair -- the compiler will not move a load above the stores of the previous pair
    for (int p0 = lane; p0 < npairs; p0 += GP * WAVE) {
      uint32_t f2[GP];
      float2 o[GP], v[GP], a[GP];
#pragma unroll
      for (int j = 0; j < GP; ++j) {
        const int p = p0 + j * WAVE, pc = p < npairs ? p : npairs - 1;
        f2[j] = F2[pc]; o[j] = os2[pc]; v[j] = ve2[pc]; a[j] = ld_ano2(an2 + pc, vol);
      }
      if constexpr (GP > 1) {
#pragma unroll
        for (int j = 0; j < GP; ++j) { PIN(f2[j]); PIN(o[j].x); PIN(o[j].y); PIN(v[j].x); PIN(v[j].y); PIN(a[j].x); PIN(a[j].y); }
      }
#pragma unroll
      for (int j = 0; j < GP; ++j) {
        const int p = p0 + j * WAVE;
        if (p >= npairs) break;
        const uint32_t fa = f2[j] & 0xFFu, fb = f2[j] >> 8;
        out4[3 * p + 0] = make_float4(o[j].x, v[j].x, (float)(fa & 1u), a[j].x);
        out4[3 * p + 1] = make_float4((float)((fa >> 2) & 1u), (float)((fa >> 4) & 1u), o[j].y, v[j].y);
        out4[3 * p + 2] = make_float4((float)(fb & 1u), a[j].y, (float)((fb >> 2) & 1u), (float)((fb >> 4) & 1u));
      }
    }
  } else {   // odd M: rows are not 16-byte aligned across envs
    for (int d = lane; d < M; d += WAVE) {
      const uint32_t f = flags[d];
      obs[6 * d + 0] = osv[d]; obs[6 * d + 1] = ver[d]; obs[6 * d + 2] = (float)(f & 1u); obs[6 * d + 3] = ld_ano(ano + d, vol);
      obs[6 * d + 4] = (float)((f >> 2) & 1u); obs[6 * d + 5] = (float)((f >> 4) & 1u);
    }
  }
}

// Role views of the state the tick leaves behind (cygym_outputs.obs_def / obs_att), straight from the env's flag plane
// in LDS -- what cygym_observe computes from global memory in a launch of its own.
// _get_defender_state (CyberDefenseEnv.py:243-257): rows of not-yet-added or non-attacker-owned devices are all -1,
// column 2 (isCompromised) is -1 everywhere.  One lane per device PAIR, three 16-byte stores (even M).
__device__ __forceinline__ void write_obs_def(const uint8_t* flags, const float* osv, const float* ver, const float* ano,
                                              float* out, int M, int lane, bool vol = false) {
  if (!(M & 1)) {
    float4* out4 = (float4*)out;
    const int npairs = M >> 1;
    const uint16_t* F2 = (const uint16_t*)flags;
    for (int p = lane; p < npairs; p += WAVE) {
      const uint32_t f2 = F2[p];
      const float2 o = ((const float2*)osv)[p], v = ((const float2*)ver)[p], a = ld_ano2((const float2*)ano + p, vol);
      const uint32_t fa = f2 & 0xFFu, fb = f2 >> 8;
      const bool ha = (fa & CG_F_NYA) || !(fa & CG_F_OWNED), hb = (fb & CG_F_NYA) || !(fb & CG_F_OWNED);
      const float ka = (float)((fa >> 2) & 1u), kb = (float)((fb >> 2) & 1u);
      out4[3 * p + 0] = ha ? make_float4(-1.f, -1.f, -1.f, -1.f) : make_float4(o.x, v.x, -1.f, a.x);
      out4[3 * p + 1] = make_float4(ha ? -1.f : ka, ha ? -1.f : 0.f, hb ? -1.f : o.y, hb ? -1.f : v.y);
      out4[3 * p + 2] = hb ? make_float4(-1.f, -1.f, -1.f, -1.f) : make_float4(-1.f, a.y, kb, 0.f);
    }
  } else {
    for (int d = lane; d < M; d += WAVE) {
      const uint32_t f = flags[d];
      const bool h = (f & CG_F_NYA) || !(f & CG_F_OWNED);
      out[6 * d + 0] = h ? -1.f : osv[d]; out[6 * d + 1] = h ? -1.f : ver[d]; out[6 * d + 2] = -1.f;
      out[6 * d + 3] = h ? -1.f : ld_ano(ano + d, vol); out[6 * d + 4] = h ? -1.f : (float)((f >> 2) & 1u); out[6 * d + 5] = h ? -1.f : 0.f;
    }
  }
}
// _get_attacker_state (CyberDefenseEnv.py:194-241): per device (os, version, compromised, known) when the attacker sees
// it (known, added, attacker-owned), else -1; then MaxExploits availability bits.  Rows are 8-byte aligned when the
// row width is even: two 8-byte stores per device.
__device__ __forceinline__ void write_obs_att(const uint8_t* flags, const float* osv, const float* ver, float* out,
                                              int M, int X, int max_exploits, int lane) {
  const int W = 4 * M + max_exploits;
  if (!(W & 1)) {
    float2* out2 = (float2*)out;
    for (int d = lane; d < M; d += WAVE) {
      const uint32_t f = flags[d];
      const bool vis = (f & CG_F_KNOWN) && !(f & CG_F_NYA) && (f & CG_F_OWNED);
      const float o = osv[d], v = ver[d];
      out2[2 * d + 0] = vis ? make_float2(o, v) : make_float2(-1.f, -1.f);
      out2[2 * d + 1] = vis ? make_float2((float)(f & 1u), 1.f) : make_float2(-1.f, -1.f);
    }
  } else {
    for (int d = lane; d < M; d += WAVE) {
      const uint32_t f = flags[d];
      const bool vis = (f & CG_F_KNOWN) && !(f & CG_F_NYA) && (f & CG_F_OWNED);
      out[4 * d + 0] = vis ? osv[d] : -1.f; out[4 * d + 1] = vis ? ver[d] : -1.f;
      out[4 * d + 2] = vis ? (float)(f & 1u) : -1.f; out[4 * d + 3] = vis ? 1.f : -1.f;
    }
  }
  if (lane < max_exploits) out[4 * M + lane] = lane < X ? 1.f : 0.f;
}

struct WaveAux { uint64_t* srcb; int32_t* park; };
// MAPS: the in-CSR columns and slot maps are staged in LDS too (the WIDE per-tick kernel, one 16-wave workgroup per CU)
template <bool MAPS, bool RT, class KP>   // RT: run-time size (comp_by may stay in global memory: one plane less in LDS)
__device__ __forceinline__ WaveAux env_setup(Env& e, uint8_t* smem, const KP& P, int M, int MC, int Mp, int MS,
                                             int wave, int lane, int env) {
  uint8_t* wb = smem + P.shared_lds + (size_t)wave * P.wave_lds;
  e.flags = wb; e.busy = wb + MS; e.wl = wb + 2 * MS; e.cby = wb + 3 * MS;
  const int n_planes = (RT && P.t.cby_global) ? 3 : 4;   // (comp_by left in global memory: Env::cby_g)
  e.scr = (uint32_t*)(wb + ((n_planes * MS + 15) & ~15));
  e.blk = e.scr + Mp + Mp / 2;   // scratch: [Mp] words + [Mp] halfwords (must match wave_lds_bytes on the host)
  e.bin = e.blk + ((P.t.EW + 3) & ~3);
  e.ring = (uint16_t*)(e.bin + ((P.t.EW + 3) & ~3));
  e.marks = (uint32_t*)(e.ring + 2 * CG_LOG_RING);
  WaveAux x;
  x.srcb = (uint64_t*)(e.marks + ((Mp / 32 + 2) & ~1));
  e.lsrc = (uint16_t*)(x.srcb + MC);
  e.devl = (int16_t*)(e.lsrc + Mp);
  x.park = (int32_t*)(wb + P.wave_lds - 128);   // [16 i32 + 3 f64] per-env scalars between fused ticks
  e.xk = (uint32_t*)(wb + P.wave_lds - 128 - P.t.x_bytes);
  e.xb = e.xk + P.t.K;
  e.xmo = (uint64_t*)(e.xb + ((P.t.KW + 1) & ~1));
  e.xmi = e.xmo + MC;
  e.K = P.t.K;
  e.optr = (const uint16_t*)(smem + P.t.o_optr); e.ocol = (const uint16_t*)(smem + P.t.o_ocol);
  e.osv = (const float*)(smem + P.t.o_os); e.ver = (const float*)(smem + P.t.o_ver); e.ano = (const float*)(smem + P.t.o_ano);   // valid when P.t.in_lds
  e.dst = smem + P.t.o_dst; e.vul = smem + P.t.o_vul; e.nap = smem + P.t.o_nap;
  e.iptr_l = (const uint16_t*)(smem + P.t.o_iptr);
  if constexpr (MAPS) {   // LDS: block / unblock picks no longer pay a global-memory hop per pass
    e.icol_g = (const uint16_t*)(smem + P.t.o_icol); e.ieid_g = (const uint16_t*)(smem + P.t.o_ieid); e.oeid_g = (const uint16_t*)(smem + P.t.o_oeid);
  } else {
    e.icol_g = (const uint16_t*)(P.t.blob + P.t.o_icol); e.ieid_g = (const uint16_t*)(P.t.blob + P.t.o_ieid); e.oeid_g = (const uint16_t*)(P.t.blob + P.t.o_oeid);
  }
  e.M = M; e.MC = MC; e.MS = MS; e.lane = lane; e.env = env;
  e.cbits = 32 - __builtin_clz((unsigned)(4 * ((MS / 4 + WAVE - 1) / WAVE)));
  e.env_id = (uint32_t)(P.c.env_id_base + env);
  e.seed = P.c.seed;
  e.multi = P.t.multi != 0;
  e.stash = P.b.stash + (size_t)env * 4 * M;
  e.cby_g = (RT && P.t.cby_global) ? P.b.live + (size_t)env * 4 * M + 3 * (size_t)M : nullptr;
  return x;
}

// FUSED: cygym_rollout (n_ticks > 1).  The per-env scalars are parked in LDS between ticks so that they are
// not loop-carried registers; the single-tick instantiation has a compile-time trip count of 1.
template <int WPB, int MT, bool FUSED, bool XE, bool WIDE>
#ifndef CG_WGP0
#define CG_WGP0 4   // run-time sizes: words / observation pairs per lane and staged step
#endif
#ifndef CG_OBS_GP0
#define CG_OBS_GP0 4
#endif
#ifndef CG_LEAN_LB
#define CG_LEAN_LB 6
#endif
#ifndef CG_WIDE_CR
#define CG_WIDE_CR 1
#endif
// (a one-wave workgroup at a run-time size is only chosen when LDS, not registers, limits residency: 3 there.)
// Register budget (second launch-bound = minimum waves per SIMD): the fused kernel and the WIDE per-tick kernel must keep 4 waves per SIMD (16 per CU: with one wave
// per env and <= 16 envs per CU that is the whole batch in ONE residency round -- at 3 per SIMD a quarter of the
// batch would wait for a second round).  The lean per-tick kernel at a compile-time size is capped for 6 waves per SIMD
// (80 VGPRs, no spills): batches that oversubscribe the chip (16384 envs) step 9 % faster than at 5.
__global__ __launch_bounds__(WPB * WAVE, FUSED ? CG_FUSED_LB : (XE ? (WPB == 1 && MT == 0 ? 3 : CG_LB) : (WIDE ? 4 : (MT && MT <= 256 ? (WPB > 1 && WPB <= 8 ? CG_LEAN_LB : CG_LEAN_LB - 1) : 1)))) void step_kernel(const KParams P0) {
  extern __shared__ __align__(16) uint8_t smem[];
#ifdef CG_STAMPS
  unsigned long long t_entry;   // before the first parameter load: stamp 0 - t_entry = the cold kernarg round trip
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_entry) :: "memory");
#endif
  // every use below goes through `P`: the by-value argument for the single-tick kernel; for the fused one a
  // pointer to the kernarg segment itself (the struct is the only argument, so it sits at offset 0), so that it
  // can be re-read, opaquely, at the top of every tick.  That pointer is in the CONSTANT address space: LLVM then
  // knows that the pointers it loads from there are global ones (not LDS / scratch) and emits global_* instead
  // of flat_* memory instructions (a flat access also ticks the LDS counter, so every LDS wait would queue
  // behind it).  No device copy of the parameters, hence no upload before the launch and nothing shared
  // between launches of one handle.
#ifndef CG_KARG_PTR
#define CG_KARG_PTR 1
#endif
  using KPT = typename KParamsOf<FUSED || CG_KARG_PTR>::type;
  KPT* pk;
  if constexpr (FUSED || CG_KARG_PTR) {
    // laundered: loads through it are not known dereferenceable at kernel entry, so the compiler leaves each one
    // next to its use instead of hoisting ~150 scalars to the top and spilling them into VGPR lanes
    const uint64_t pv = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t plo = (uint32_t)pv, phi = (uint32_t)(pv >> 32);
    asm volatile("" : "+s"(plo), "+s"(phi));
    pk = (KPT*)(((uint64_t)phi << 32) | plo);
  } else pk = &P0;
#define P (*pk)
  const int M = MT ? MT : P.t.M, MC = MT ? (MT + WAVE - 1) / WAVE : P.t.MC, Mp = MC * WAVE, MS = (M + 3) & ~3;
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;   // (run-time-size rollout kernel: laundered per tick, below)
  const int env = uni(P.env_begin + blockIdx.x * WPB + wave);
  const bool live = env < P.env_end;
  const int G = P.a.max_groups, L = P.a.max_devs;

  Env e;
  WaveAux aux = env_setup<WIDE, MT == 0>(e, smem, P, M, MC, Mp, MS, wave, lane, live ? env : 0);
  uint64_t* srcb = aux.srcb;
  int32_t* park = aux.park;
  e.env = env;
  e.blk_dirty = e.ring_dirty = e.x_dirty = false;

  STAMP(0);
  // ---- issue every global load of this tick up front (one memory latency, not a chain) ----
  const size_t so = (size_t)(live ? env : 0) * 4 * M;
  const uint8_t* g_live = P.b.live + so;
  int32_t ie[CG_I_COUNT];
  double fe[CG_D_COUNT];
  int mode = 0, ng = 0, at0 = 8, cnt0 = 0, nexp0 = 0, app0 = -1;
  int ex0 = -1;   // first exploit id of an attacker spread: fetched as soon as the header says so (its latency hides
                  // behind the staging), not inside the spread where the heaviest envs would wait a full round trip
  // Prefetch depths (items per lane held in registers between the load and the LDS store): at a compile-time size
  // the whole state is one item per lane; at run-time sizes (up to 2048 devices: 8 items of the live block, 5
  // blocked words, 8 KB of topology per wave) deep enough that the prologue stays ONE memory round trip instead of
  // a load -> wait -> store chain per item.
  constexpr int PF_LIVE = MT ? (MT / 4 + WAVE - 1) / WAVE : 8;
  uint4 rl[PF_LIVE];
#pragma unroll
  for (int j = 0; j < PF_LIVE; ++j) rl[j] = make_uint4(0, 0, 0, 0);
  uint32_t ringw = 0;
  constexpr int PF_BLK = MT ? 2 : 5, PF_DEV = 1;   // words / list entries per lane prefetched into registers
  uint32_t bw[PF_BLK], bwi[PF_BLK];
  int16_t dv[PF_DEV];
  const bool vec = (M & 3) == 0;
  // uint4 items of the live block when M % 4 == 0: all [4][M] bytes, or -- comp_by left in global memory (run-time sizes) --
  // the first three planes, rounded up to whole items (the few bytes of plane 3 that come along land in the padding
  // in front of the scratch area)
  const bool cbg = MT == 0 && P.t.cby_global;
  const int items = cbg ? (3 * M + 15) >> 4 : M >> 2;
  const int items_all = M >> 2;                          // the whole [4][M] block (stash copies)
  const int items_wb = cbg ? (3 * M) >> 4 : M >> 2;      // write-back: whole items of the staged planes only ...
  const int tail_wb = cbg ? ((3 * M) & 15) >> 2 : 0;     // ... and the remaining words one by one (plane 3 is not ours to write)
  // Run-time sizes (up to 2048 devices, where one wave per SIMD is resident and nothing hides a round trip): every
  // prologue load is pinned before the first LDS store.  A load whose only use sits in a conditional block is
  // otherwise SUNK into that block, next to its s_waitcnt, and the staging runs as a chain of 8-16 dependent
  // round trips.  (At the compile-time sizes, with 4+ waves per SIMD, the pinned form measured +-2 % and costs
  // registers: not applied there.)
#define KEEP4(r) asm volatile("" :: "v"((r).x), "v"((r).y), "v"((r).z), "v"((r).w))
  constexpr int PF_BLOB = MT ? 4 : (WPB >= 16 ? 2 : WPB >= 8 ? 4 : 8);   // 16-byte items per lane: 32 KB per workgroup in the pinned round
  const uint4* blob_src = (const uint4*)P.t.blob;
  const int n16 = P.t.lds_bytes >> 4;
  constexpr int stride = WPB * WAVE;
  uint4 br[PF_BLOB];
#pragma unroll
  for (int j = 0; j < PF_BLOB; ++j) { const int i = threadIdx.x + j * stride; br[j] = blob_src[i < n16 ? i : n16 - 1]; }
  if constexpr (MT == 0) {
    // straight-line: every load unconditional at a clamped (always valid) address, the per-lane state before the
    // per-env scalars (loads return in order: the header's readfirstlane then waits for everything at once), the
    // first exploit id fetched whether or not the action is a spread.  Waves past the end of the batch read env 0.
    const int envc = live ? env : 0;
    const int EW = P.t.EW;   // >= 1 (cygym_create)
#pragma unroll
    for (int j = 0; j < PF_LIVE; ++j) {
      const int i = lane + j * WAVE;
      if (vec) rl[j] = ((const uint4*)g_live)[i < items ? i : items - 1];   // (uniform condition; items >= 1 when vec)
    }
    ringw = ((const uint32_t*)(P.b.ring + (size_t)envc * CG_LOG_RING * 2))[lane < CG_LOG_RING ? lane : CG_LOG_RING - 1];
    const uint32_t* gb0 = P.b.blocked + (size_t)envc * EW;
    const uint32_t* gbi0 = P.b.blocked_in + (size_t)envc * EW;
#pragma unroll
    for (int j = 0; j < PF_BLK; ++j) { int w = lane + j * WAVE; w = w < EW ? w : EW - 1; bw[j] = gb0[w]; bwi[j] = gbi0[w]; }
#pragma unroll
    for (int j = 0; j < PF_DEV; ++j) { const int q = lane + j * WAVE; dv[j] = P.a.dev_idx[(size_t)envc * L + (q < L ? q : L - 1)]; }
    if constexpr (!FUSED) {   // (the rollout kernel loads each tick's header at the top of its tick loop)
      mode = P.a.mode[envc];
      ng = P.a.n_groups[envc];
      at0 = P.a.atype[(size_t)envc * G];
      cnt0 = P.a.dev_cnt[(size_t)envc * G];
      nexp0 = P.a.n_exploit[(size_t)envc * G];
      app0 = P.a.app[(size_t)envc * G];
      ex0 = P.a.exploit[(size_t)envc * G * CG_MAX_EXPLOITS];
    }
    const int32_t* g = P.b.ienv + (size_t)envc * CG_I_COUNT;
#pragma unroll
    for (int i = 0; i < CG_I_COUNT; ++i) ie[i] = g[i];
    const double* gf = P.b.fenv + (size_t)envc * CG_D_COUNT;
#pragma unroll
    for (int i = 0; i < CG_D_COUNT; ++i) fe[i] = gf[i];
#pragma unroll
    for (int j = 0; j < PF_LIVE; ++j) KEEP4(rl[j]);
#pragma unroll
    for (int j = 0; j < PF_BLK; ++j) asm volatile("" :: "v"(bw[j]), "v"(bwi[j]));
    asm volatile("" :: "v"(ringw), "v"((int)dv[0]));
  } else if (live) {
    const int32_t* g = P.b.ienv + (size_t)env * CG_I_COUNT;
#pragma unroll
    for (int i = 0; i < CG_I_COUNT; ++i) ie[i] = g[i];
    const double* gf = P.b.fenv + (size_t)env * CG_D_COUNT;
#pragma unroll
    for (int i = 0; i < CG_D_COUNT; ++i) fe[i] = gf[i];
    mode = P.a.mode[env];
    ng = P.a.n_groups[env];
    at0 = P.a.atype[(size_t)env * G];
    cnt0 = P.a.dev_cnt[(size_t)env * G];
    nexp0 = P.a.n_exploit[(size_t)env * G];
    app0 = P.a.app[(size_t)env * G];
    if ((mode & 0xFF) == CG_MODE_ATTACKER && at0 == 1 && ng == 0) ex0 = P.a.exploit[(size_t)env * G * CG_MAX_EXPLOITS];
    if (vec) {
#pragma unroll
      for (int j = 0; j < PF_LIVE; ++j) { const int i = lane + j * WAVE; rl[j] = ((const uint4*)g_live)[i < items ? i : 0]; }
    }
    if (lane < CG_LOG_RING) ringw = ((const uint32_t*)(P.b.ring + (size_t)env * CG_LOG_RING * 2))[lane];
#pragma unroll
    for (int j = 0; j < PF_BLK; ++j) {
      int w = lane + j * WAVE;
      bw[j] = w < P.t.EW ? P.b.blocked[(size_t)env * P.t.EW + w] : 0u;
      bwi[j] = w < P.t.EW ? P.b.blocked_in[(size_t)env * P.t.EW + w] : 0u;
    }
#pragma unroll
    for (int j = 0; j < PF_DEV; ++j) { int q = lane + j * WAVE; dv[j] = q < L ? P.a.dev_idx[(size_t)env * L + q] : (int16_t)0; }
  }
  // ---- workgroup-shared topology blob -> LDS ----
  {
    if constexpr (MT == 0) {
#pragma unroll
      for (int j = 0; j < PF_BLOB; ++j) KEEP4(br[j]);
    }
    uint4* dstp = (uint4*)smem;
#pragma unroll
    for (int j = 0; j < PF_BLOB; ++j) { const int i = threadIdx.x + j * stride; if (i < n16) dstp[i] = br[j]; }
    for (int i = threadIdx.x + PF_BLOB * stride; i < n16; i += stride) dstp[i] = blob_src[i];
  }
#undef KEEP4
  if (live) {
    if (vec) {
#pragma unroll
      for (int j = 0; j < PF_LIVE; ++j) { const int i = lane + j * WAVE; if (i < items) ((uint4*)e.flags)[i] = rl[j]; }
      for (int i = lane + PF_LIVE * WAVE; i < items; i += WAVE) ((uint4*)e.flags)[i] = ((const uint4*)g_live)[i];
    } else {
      for (int pl = 0; pl < 4; ++pl)
        for (int i = lane; i < MS; i += WAVE) e.flags[pl * MS + i] = i < M ? g_live[pl * M + i] : (pl == 0 ? (uint8_t)CG_F_NYA : (uint8_t)0);
    }
    if (lane < CG_LOG_RING) ((uint32_t*)e.ring)[lane] = ringw;
    const uint32_t* gb = P.b.blocked + (size_t)env * P.t.EW;
#pragma unroll
    for (int j = 0; j < PF_BLK; ++j) { int w = lane + j * WAVE; if (w < P.t.EW) { e.blk[w] = bw[j]; e.bin[w] = bwi[j]; } }
    for (int w = lane + PF_BLK * WAVE; w < P.t.EW; w += WAVE) { e.blk[w] = gb[w]; e.bin[w] = P.b.blocked_in[(size_t)env * P.t.EW + w]; }
    const int16_t* gd = P.a.dev_idx + (size_t)env * L;
#pragma unroll
    for (int j = 0; j < PF_DEV; ++j) { int q = lane + j * WAVE; if (q < L) e.devl[q] = dv[j]; }
    for (int q = lane + PF_DEV * WAVE; q < L; q += WAVE) e.devl[q] = gd[q];
  }
  __syncthreads();   // the only workgroup barrier: waves diverge per env from here on
  if (!live) return;
  STAMP(1);
#ifndef CG_NO_UNIFORM_SCALARS
  if constexpr (!FUSED) {
    // The 16 + 3 per-env scalars were fetched with vector loads (a uniform address into memory the kernel also
    // writes is not eligible for the scalar cache), i.e. into 22 VGPRs that stay live to the write-back.  Telling
    // the compiler they are uniform moves them to SGPRs: the VGPRs go back to the per-lane work (the WIDE kernel
    // spilled 7 of them, an f64 accumulator among them), and a spilled SGPR costs a v_writelane, not scratch traffic.
#pragma unroll
    for (int i = 0; i < CG_I_COUNT; ++i) ie[i] = __builtin_amdgcn_readfirstlane(ie[i]);
#pragma unroll
    for (int i = 0; i < CG_D_COUNT; ++i)
      fe[i] = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(fe[i])), __builtin_amdgcn_readfirstlane(__double2loint(fe[i])));
    mode = __builtin_amdgcn_readfirstlane(mode); ng = __builtin_amdgcn_readfirstlane(ng); at0 = __builtin_amdgcn_readfirstlane(at0);
    cnt0 = __builtin_amdgcn_readfirstlane(cnt0); nexp0 = __builtin_amdgcn_readfirstlane(nexp0); app0 = __builtin_amdgcn_readfirstlane(app0);
    ex0 = __builtin_amdgcn_readfirstlane(ex0);   // (the tick's action header: uniform as well)
  }
#endif

  const int NW = MS >> 2;
  // word loops run in groups of WGP words per lane, loads first (see the chunk loops of the spread): one LDS round trip
  // per group instead of one per word; a single word per lane at the compile-time sizes
  constexpr int CGP = MT ? ((MT + WAVE - 1) / WAVE < 4 ? (MT + WAVE - 1) / WAVE : 4) : 4;   // chunks per staged group
  constexpr int WGP = MT ? (((MT + 3) / 4 + WAVE - 1) / WAVE < 4 ? ((MT + 3) / 4 + WAVE - 1) / WAVE : 4) : CG_WGP0;

  // ---- ticks of this launch: 1 for cygym_step, T for cygym_rollout (state stays in LDS / registers;
  // no cross-env synchronisation between ticks) ----
  const int n_ticks = FUSED ? P.n_ticks : 1;
  // Rollout kernel: EVERY tick, the first one included, starts from the same point -- scalars parked in LDS, the
  // env view re-derived through a laundered parameter pointer, the tick's header and device list loaded here.
  // With tick 0 special-cased (its header prefetched with the state) every per-env value reached the loop as a
  // phi of "prologue version" and "re-derived version" and stayed live across the whole body: the full-feature
  // rollout kernels spilled 10-47 VGPRs at the 128 cap; now none does.  (Laundering the lane / wave ids per tick as
  // well stops the hoisting of per-lane addresses and fits 96 VGPRs = 5 waves per SIMD, but the recomputation costs
  // 3-4 % per tick: +6 % at 16384 envs, -3 % at 4096 and 65536 -- measured, not adopted.)
  if constexpr (FUSED) {
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < CG_I_COUNT; ++i) park[i] = ie[i];
#pragma unroll
      for (int i = 0; i < CG_D_COUNT; ++i) ((double*)(park + CG_I_COUNT))[i] = fe[i];
    }
    wsync();
  }
  for (int tk = 0; tk < n_ticks; ++tk) {
  // run-time sizes only: also keep per-lane addresses and masks from being hoisted out of the tick loop (the chunk
  // loops are not unrolled there, so the recomputation is cheap, and without it the kernel spilled 19-42 VGPRs)
  if constexpr (FUSED && MT == 0) asm volatile("" : "+v"(lane), "+v"(wave));
  const size_t te = (size_t)tk * P.n_envs + env;   // row of this (tick, env) in the action / output arrays
  if (FUSED && tk > 0) STAMP(0);   // diagnostic builds: the stamps then describe the LAST tick of the rollout
  if (FUSED) {
    // Re-derive everything uniform from the device copy of the parameters: keeping ~200 loop-invariant
    // scalars alive across the tick body would spill SGPRs into VGPRs and halve the occupancy.
    {
      const uint64_t pv = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
      uint32_t plo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pv);
      uint32_t phi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pv >> 32));
      asm volatile("" : "+s"(plo), "+s"(phi));   // opaque: nothing derived from it is hoisted out of the tick loop
      pk = (KPT*)(((uint64_t)phi << 32) | plo);
    }
    aux = env_setup<WIDE, MT == 0>(e, smem, P, M, MC, Mp, MS, wave, lane, env);
    srcb = aux.srcb; park = aux.park;
    // parked in LDS: an LDS load lands in a VGPR; readfirstlane tells the compiler the value is uniform, so the
    // 22 per-env scalars live in SGPRs across the tick body instead of 22 of the 128 VGPRs
#pragma unroll
    for (int i = 0; i < CG_I_COUNT; ++i) ie[i] = __builtin_amdgcn_readfirstlane(park[i]);
#pragma unroll
    for (int i = 0; i < CG_D_COUNT; ++i)
      fe[i] = __hiloint2double(__builtin_amdgcn_readfirstlane(park[CG_I_COUNT + 2 * i + 1]),
                               __builtin_amdgcn_readfirstlane(park[CG_I_COUNT + 2 * i]));
    mode = P.a.mode[te];
    ng = P.a.n_groups[te];
    at0 = P.a.atype[te * G];
    cnt0 = P.a.dev_cnt[te * G];
    nexp0 = P.a.n_exploit[te * G];
    app0 = P.a.app[te * G];
    if ((mode & 0xFF) == CG_MODE_ATTACKER && at0 == 1 && ng == 0) ex0 = P.a.exploit[te * G * CG_MAX_EXPLOITS];
    const int16_t* gd = P.a.dev_idx + te * L;
    for (int q = lane; q < L; q += WAVE) e.devl[q] = gd[q];
    wsync();
  }
  if (FUSED && tk > 0) STAMP(1);
  if (ng < 0) continue;   // n_groups < 0: this env does not tick (per-env stepping inside a batch); its parked scalars stand
  const int16_t* devs = e.devl;
  uint32_t* const F = (uint32_t*)e.flags;
  uint32_t* const Bz = (uint32_t*)e.busy;
  uint32_t* const Wl = (uint32_t*)e.wl;
  const uint32_t* const Ds = (const uint32_t*)e.dst;
  const bool partial = (mode & CG_MODE_PARTIAL) && ng == 0;   // step(action, agent_cnt != len(net)) :1207
  const int baseline = CG_MODE_BASELINE_OF(mode, P.c.baseline);   // env.base_line: per env and tick when the mode word carries it
  mode &= 0xFF;
  e.tick = (uint32_t)ie[CG_I_RNG_TICK];
  e.eflags = ie[CG_I_FLAGS];
  e.log_total = ie[CG_I_LOG_TOTAL];
  if (COLD(XE && (!FUSED || tk == 0) && x_cnt(e) > 0)) {   // this env carries edges evolve_network added: stage its list
    const uint32_t* xg = P.b.extra + (size_t)env * (P.t.K + P.t.KW);
    const int nx = x_cnt(e) < P.t.K ? x_cnt(e) : P.t.K;
    for (int j = lane; j < nx; j += WAVE) e.xk[j] = xg[j];
    for (int w = lane; w < P.t.KW; w += WAVE) e.xb[w] = xg[P.t.K + w];
    wsync();
    x_masks(e);
  }
  double cost = 0.0;
  bool dirty = false;
  int last_atype = -1;

  if (ng == 0) {   // ---- step(action) volt_typhoon_env.py:818-1333 ----
    int at = at0;
    int Ld = cnt0;
    if (Ld > L) Ld = L;
    if (Ld < 0) Ld = 0;
    if (mode == CG_MODE_DEFENDER) { if (!(at >= 0 && at < P.c.n_def_actions)) at = 8; }
    else                          { if (!(at >= 0 && at < P.c.n_att_actions)) at = 3; }
    for (int w0 = lane; w0 < NW; w0 += WGP * WAVE) {   // :904-908 decay of the cached busy set
      uint32_t bj[WGP], fj[WGP];
#pragma unroll
      for (int j = 0; j < WGP; ++j) { const int w = w0 + j * WAVE, wc = w < NW ? w : w0; bj[j] = Bz[wc]; fj[j] = F[wc]; }
      if constexpr (WGP > 1) {
#pragma unroll
        for (int j = 0; j < WGP; ++j) { PIN(bj[j]); PIN(fj[j]); }
      }
#pragma unroll
      for (int j = 0; j < WGP; ++j) { const int w = w0 + j * WAVE; if (w < NW) Bz[w] = bj[j] - (((fj[j] >> 6) & ONES) & nz01(bj[j])); }
    }
    wsync();
    if (mode == CG_MODE_DEFENDER) {
      if (baseline != 0) at = 8;   // :913-914
      def_global(e, P, at, devs, Ld, cost, dirty, false, ie, fe);
      if (at == 1 || at == 4 || at == 5 || at == 6 || at == 7 || at == 9 || at == 12 || at == 13)
        if (Ld > 0) def_per_device<XE, WIDE, XE && !FUSED>(e, P, at, devs, Ld, app0, cost, dirty, ie, fe);
    } else if (baseline != 3 && (at == 1 || at == 2)) {
      // :1127 snapshot of the sources.  Chunk loops are STAGED in groups of four: the LDS reads of a group are issued
      // before its first store (the compiler cannot reorder an LDS load over an LDS store it cannot disambiguate, so a
      // plain chunk loop pays one LDS round trip per chunk, one after the other)
#pragma nounroll
      for (int c0 = 0; c0 < MC; c0 += CGP) {
        uint32_t fj[CGP];
#pragma unroll
        for (int j = 0; j < CGP; ++j) { const int d = (c0 + j) * WAVE + lane; fj[j] = d < M ? e.flags[d] : 0u; }
#pragma unroll
        for (int j = 0; j < CGP; ++j) {
          const uint64_t m = ballot(fj[j] & (CG_F_COMP | CG_F_OWNED));
          if (lane == 0 && c0 + j < MC) srcb[c0 + j] = m;
        }
      }
      wsync();
      if (at == 1) {
        int ne = nexp0;
        if (ne > CG_MAX_EXPLOITS) ne = CG_MAX_EXPLOITS;
        __builtin_amdgcn_s_setprio(3);   // the spread bounds the launch: win issue arbitration over short envs
        attacker_spread<XE, (MT == 0 && WPB <= 8) ? 4 : (WIDE ? CG_WIDE_CR : 1), CGP, WIDE>(e, P, P.a.exploit + te * G * CG_MAX_EXPLOITS, ex0, ne, srcb);
        __builtin_amdgcn_s_setprio(0);
      } else {
        attacker_probe<XE>(e, srcb, cost);
      }
    }
    last_atype = at;
  } else {   // ---- step_grouped(groups) :694-779 via _step_apply_only :612-692 ----
    uint8_t* occ = (uint8_t*)(e.scr + Mp);   // second scratch half: clean-stall occurrence numbers
    for (int i = lane; i < Mp / 4; i += WAVE) ((uint32_t*)occ)[i] = 0;
    wsync();
    const int16_t* dp = devs;
    int used = 0;
    for (int g = 0; g < ng && g < G; ++g) {
      int at = P.a.atype[te * G + g];
      int Ld = P.a.dev_cnt[te * G + g];
      if (Ld < 0) Ld = 0;
      if (used + Ld > L) Ld = L - used;
      if (mode == CG_MODE_DEFENDER && at == 0) at = 8;
      else if (mode == CG_MODE_ATTACKER && at == 0) at = 3;
      if (mode == CG_MODE_DEFENDER) {
        if (baseline != 0) at = 8;
        def_global(e, P, at, dp, Ld, cost, dirty, true, ie, fe);
        if (at == 1 && Ld > 0) def_clean(e, P, dp, Ld, cost, ie, fe, occ);
      }
      dp += Ld; used += Ld;
    }
    for (int w = lane; w < NW; w += WAVE) { uint32_t b = Bz[w]; Bz[w] = b - nz01(b); }   // _tick_busy_time_once :607
    wsync();
  }

  STAMP(2);
  // ---- fused word pass: workload advance (:1242-1261 / :705-725) + every per-tick count ----
  int c_fin = 0, c_act = 0, c_idle = 0, c_fsrv = 0, c_comp = 0, c_cdc = 0;
  for (int w0 = lane; w0 < NW; w0 += WGP * WAVE) {
    uint32_t fj[WGP], bj[WGP], lj[WGP], sj[WGP];
#pragma unroll
    for (int j = 0; j < WGP; ++j) { const int w = w0 + j * WAVE, wc = w < NW ? w : w0; fj[j] = F[wc]; bj[j] = Bz[wc]; lj[j] = Wl[wc]; sj[j] = Ds[wc]; }
    if constexpr (WGP > 1) {
#pragma unroll
      for (int j = 0; j < WGP; ++j) { PIN(fj[j]); PIN(bj[j]); PIN(lj[j]); PIN(sj[j]); }
    }
#pragma unroll
    for (int j = 0; j < WGP; ++j) {
    const int w = w0 + j * WAVE;
    if (w >= NW) break;
    uint32_t f = fj[j], b = bj[j], l = lj[j], st = sj[j];
    const uint32_t nya = (f >> 4) & ONES;
    const uint32_t step = partial ? 0u : (~nz01(b) & ~nya & nz01(l) & ONES);   // idle-of-stall, active, has a job
    l -= step;
    const uint32_t fin = step & ~nz01(l);
    const uint32_t adv = (f >> 7) & ONES;
    f &= ~((fin & adv) << 7);
    Wl[w] = l; F[w] = f;
    const uint32_t act = ~nya & ONES;
    const uint32_t idl = act & ~nz01(b) & ~nz01(l);
    const uint32_t cmp = f & ~nya & ~(f >> 1) & ONES;          // compromised, active, not attacker-owned
    c_fin += __popc(fin & ~adv);
    c_act += __popc(act);
    c_idle += __popc(idl);
    c_fsrv += __popc(idl & (st >> 1));
    c_comp += __popc(cmp);
    c_cdc += __popc(cmp & st);
    }
  }
  // six per-lane counts (each <= 32, sums <= 2048) as three packed 16 + 16-bit wave reductions on the DPP path
  const uint32_t s_fa = (uint32_t)wave_sum(c_fin | (c_act << 16));
  const uint32_t s_is = (uint32_t)wave_sum(c_idle | (c_fsrv << 16));
  const uint32_t s_cd = (uint32_t)wave_sum(c_comp | (c_cdc << 16));
  const int current_work = (int)(s_fa & 0xFFFFu), n_active = (int)(s_fa >> 16);
  const int n_idle = (int)(s_is & 0xFFFFu), n_fsrv = (int)(s_is >> 16);
  const int n_comp = (int)(s_cd & 0xFFFFu), n_comp_dc = (int)(s_cd >> 16);
  ie[CG_I_WORK_DONE] += current_work;
  wsync();
  if (!partial) arrivals(e, P, ie[CG_I_STEP_NUM], n_active, n_idle, n_fsrv);   // changes wl only: the counts above stand

  STAMP(3);
  // ---- rewards (:1267-1304 / :732-748) ----
  if (ng == 0) ie[CG_I_COMP_CNT] += n_comp;
  ie[CG_I_LAST_NCOMP] = n_comp;
  double raw, shaped;
  {
    double def_work = P.c.work_scale * current_work;
    if (mode == CG_MODE_DEFENDER) {
      raw = cost + def_work - n_comp * P.c.comp_scale;
      shaped = raw;
    } else {
      double r = cost + P.c.comp_scale * (n_comp + 10 * n_comp_dc);
      double phi = (double)n_comp / (double)M;
      if (!(e.eflags & CG_E_PREV_SET)) { fe[CG_D_PREV_ATT_POT] = phi; e.eflags |= CG_E_PREV_SET; }
      double inc = P.c.gamma * phi - fe[CG_D_PREV_ATT_POT];
      double bonus = 0.1 * inc + 0.0;
      fe[CG_D_PREV_ATT_POT] = P.c.gamma * phi;
      raw = r;
      shaped = r + bonus;
    }
  }

  STAMP(4);
  // ---- observation (_get_state CyberDefenseEnv.py:146-191), before evolve.  The static float columns come from
  // LDS, or (large M, where leaving them out of LDS buys resident waves) from the L2-resident blob.
  constexpr int OBS_GP = MT ? ((MT / 2 + WAVE - 1) / WAVE < 4 ? (MT / 2 + WAVE - 1) / WAVE : 4) : CG_OBS_GP0;
  if (P.o.obs) {
    bool dyn_ano = false;
    if constexpr (XE && !FUSED) dyn_ano = COLD(P.b.anomaly != nullptr);   // this env's own Device.anomaly_score plane (slow scan path; per-tick kernels only)
    if (dyn_ano) write_obs<OBS_GP>(e.flags, P.t.in_lds ? e.osv : (const float*)(P.t.blob + P.t.o_os), P.t.in_lds ? e.ver : (const float*)(P.t.blob + P.t.o_ver),
                                   P.b.anomaly + (size_t)env * M, P.o.obs + te * M * 6, M, lane, true);
    else if (P.t.in_lds) write_obs<OBS_GP>(e.flags, e.osv, e.ver, e.ano, P.o.obs + te * M * 6, M, lane);
    else write_obs<OBS_GP>(e.flags, (const float*)(P.t.blob + P.t.o_os), (const float*)(P.t.blob + P.t.o_ver),
                           (const float*)(P.t.blob + P.t.o_ano), P.o.obs + te * M * 6, M, lane);
  }

  STAMP(5);
  if (!partial) {   // :1307-1312
    ie[CG_I_STEP_NUM] += 1;
    if (mode == CG_MODE_ATTACKER) ie[CG_I_ATT_STEP] += 1; else ie[CG_I_DEF_STEP] += 1;
  }
  const bool done = ie[CG_I_STEP_NUM] > P.c.episode_limit;
  if (dirty || umod(ie[CG_I_STEP_NUM], P.c.evolve_period) == 0) evolve<XE>(e, P);
  if (ng == 0) {   // :1330 rebuild of the cached busy set
    for (int w0 = lane; w0 < NW; w0 += WGP * WAVE) {
      uint32_t fj[WGP], bj[WGP];
#pragma unroll
      for (int j = 0; j < WGP; ++j) { const int w = w0 + j * WAVE, wc = w < NW ? w : w0; fj[j] = F[wc]; bj[j] = Bz[wc]; }
      if constexpr (WGP > 1) {
#pragma unroll
        for (int j = 0; j < WGP; ++j) { PIN(fj[j]); PIN(bj[j]); }
      }
#pragma unroll
      for (int j = 0; j < WGP; ++j) { const int w = w0 + j * WAVE; if (w < NW) F[w] = (fj[j] & ~(ONES * CG_F_BUSYC)) | (nz01(bj[j]) << 6); }
    }
  }
  wsync();
  ie[CG_I_RNG_TICK] += 1;
  ie[CG_I_LAST_ATYPE] = last_atype;
  ie[CG_I_LOG_TOTAL] = e.log_total;
  ie[CG_I_FLAGS] = e.eflags | (__any(e.eflags & CG_E_BUSY_SAT) ? CG_E_BUSY_SAT : 0);

  if (lane == 0) {
    P.o.raw[te] = raw;
    P.o.shaped[te] = shaped;
    P.o.done[te] = done ? 1 : 0;
    if (COLD(P.o.ret != nullptr) && P.o.alive && P.o.alive[env]) {   // episode returns of a rollout loop (do_agent.py:266-274)
      P.o.ret[(size_t)env * 2 + (mode & 1)] += raw;
      if (done) P.o.alive[env] = 0;
    }
  }

  if (done && P.c.auto_reset && P.snap.live) {   // reload the initial snapshot; the RNG tick stays monotone
    const int si = P.snap.n_envs == 1 ? 0 : env;
    const size_t ss = (size_t)si * 4 * M;
    wsync();
    if (vec) {
      for (int i = lane; i < items; i += WAVE) ((uint4*)e.flags)[i] = ((const uint4*)(P.snap.live + ss))[i];
      if (cbg) {   // comp_by of the reloaded env: snapshot -> global, word by word
        const uint32_t* cs = (const uint32_t*)(P.snap.live + ss + 3 * (size_t)M);
        uint32_t* cd = (uint32_t*)e.cby_g;
#pragma nounroll
        for (int i = lane; i < M / 4; i += WAVE) cd[i] = cs[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      }
    } else {
      for (int pl = 0; pl < 4; ++pl)
        for (int i = lane; i < M; i += WAVE) e.flags[pl * MS + i] = P.snap.live[ss + pl * M + i];
    }
    if (vec) {   // 16 bytes per lane and trip (byte by byte this copy was 128 dependent round trips at 2048 devices)
#pragma nounroll
      for (int i = lane; i < items_all; i += WAVE) ((uint4*)(P.b.stash + so))[i] = ((const uint4*)(P.snap.stash + ss))[i];
    } else {
      for (int i = lane; i < 4 * M; i += WAVE) P.b.stash[so + i] = P.snap.stash[ss + i];
    }
    for (int w = lane; w < P.t.EW; w += WAVE) {
      e.blk[w] = P.snap.blocked[(size_t)si * P.t.EW + w];
      e.bin[w] = P.snap.blocked_in[(size_t)si * P.t.EW + w];
    }
    if (lane < CG_LOG_RING) ((uint32_t*)e.ring)[lane] = ((const uint32_t*)(P.snap.ring + (size_t)si * CG_LOG_RING * 2))[lane];
    e.blk_dirty = e.ring_dirty = true;
    if constexpr (XE) {   // the pickled env carries its logger and detector too (volt_typhoon_env.py:1904-1936)
      if (COLD(P.b.hist && P.snap.hist)) {
        const uint32_t* hs = (const uint32_t*)(P.snap.hist + (size_t)si * CG_HIST_RING * 2);
        uint32_t* hd = (uint32_t*)(P.b.hist + (size_t)env * CG_HIST_RING * 2);
#pragma nounroll
        for (int i = lane; i < CG_HIST_RING; i += WAVE) hd[i] = hs[i];
      }
      if (!FUSED && COLD(P.b.anomaly && P.snap.anomaly)) {
        const float* as = P.snap.anomaly + (size_t)si * M;
        float* ad = P.b.anomaly + (size_t)env * M;
#pragma nounroll
        for (int i = lane; i < M; i += WAVE) ad[i] = as[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      }
      if (COLD(P.b.forest && P.snap.forest)) {
        const uint32_t* fs = P.snap.forest + (size_t)si * CG_FOREST_WORDS;
        uint32_t* fd = P.b.forest + (size_t)env * CG_FOREST_WORDS;
#pragma nounroll
        for (int i = lane; i < CG_FOREST_WORDS; i += WAVE) fd[i] = fs[i];
      }
    }
    if (COLD(XE && P.t.K > 0)) {   // the snapshot's extra-edge list (normally empty) replaces the episode's
      const int ns = P.snap.extra ? (int)((uint32_t)P.snap.ienv[(size_t)si * CG_I_COUNT + CG_I_FLAGS] >> CG_E_NX_SHIFT) : 0;
      const uint32_t* xs = P.snap.extra + (size_t)si * (P.t.K + P.t.KW);
      for (int j = lane; j < ns; j += WAVE) e.xk[j] = xs[j];
      for (int w = lane; w < P.t.KW; w += WAVE) e.xb[w] = ns > 0 ? xs[P.t.K + w] : 0u;
      e.eflags = (e.eflags & 0xFFFF) | (ns << CG_E_NX_SHIFT);
      wsync();
      x_masks(e);
      e.x_dirty = true;
    }
    const int32_t keep_tick = ie[CG_I_RNG_TICK];
    const int32_t* g = P.snap.ienv + (size_t)si * CG_I_COUNT;
#pragma unroll
    for (int i = 0; i < CG_I_COUNT; ++i) ie[i] = g[i];
    ie[CG_I_RNG_TICK] = keep_tick;
    if (!(P.t.K > 0 && P.snap.extra)) ie[CG_I_FLAGS] &= 0xFFFF;
    const double* gf = P.snap.fenv + (size_t)si * CG_D_COUNT;
#pragma unroll
    for (int i = 0; i < CG_D_COUNT; ++i) fe[i] = gf[i];
    wsync();
  }
  // ---- optional role views of the state this tick leaves behind (after evolve / auto-reset): what the next actor's
  // policy reads (cygym_outputs.obs_def / obs_att); replaces a cygym_observe launch per tick of a closed loop ----
  if (COLD(P.o.obs_def != nullptr) || COLD(P.o.obs_att != nullptr)) {
    const float* osv = P.t.in_lds ? e.osv : (const float*)(P.t.blob + P.t.o_os);
    const float* ver = P.t.in_lds ? e.ver : (const float*)(P.t.blob + P.t.o_ver);
    const float* ano = P.t.in_lds ? e.ano : (const float*)(P.t.blob + P.t.o_ano);
    bool dyn_ano = false;
    if constexpr (XE && !FUSED) dyn_ano = COLD(P.b.anomaly != nullptr);
    if (dyn_ano) ano = P.b.anomaly + (size_t)env * M;
    if (P.o.obs_def) write_obs_def(e.flags, osv, ver, ano, P.o.obs_def + te * M * 6, M, lane, dyn_ano);
    if (P.o.obs_att) write_obs_att(e.flags, osv, ver, P.o.obs_att + te * (size_t)(4 * M + P.c.max_exploits), M, P.t.X, P.c.max_exploits, lane);
  }
  if (FUSED && tk + 1 < n_ticks) {   // park the scalars for the next tick
    if (lane == 0) {
#pragma unroll
      for (int i = 0; i < CG_I_COUNT; ++i) park[i] = ie[i];
#pragma unroll
      for (int i = 0; i < CG_D_COUNT; ++i) ((double*)(park + CG_I_COUNT))[i] = fe[i];
    }
    wsync();
  }
  if (FUSED) STAMP(6);
  }   // for tk

  STAMP(6);
  // ---- write back: the whole [4][M] live block with 16-byte stores ----
  // (the lane id is laundered: the per-lane global addresses of the write-back are recomputed here instead of being
  // kept -- or spilled -- from the prologue, where the same addresses were used for the loads)
  asm volatile("" : "+v"(lane));
  if (vec) {
    for (int i = lane; i < items_wb; i += WAVE) ((uint4*)(P.b.live + so))[i] = ((const uint4*)e.flags)[i];
    if (lane < tail_wb) ((uint32_t*)(P.b.live + so))[items_wb * 4 + lane] = ((const uint32_t*)e.flags)[items_wb * 4 + lane];
  } else {
    for (int pl = 0; pl < 4; ++pl)
      for (int i = lane; i < M; i += WAVE) P.b.live[so + pl * M + i] = e.flags[pl * MS + i];
  }
  if (e.blk_dirty)
    for (int w = lane; w < P.t.EW; w += WAVE) {
      P.b.blocked[(size_t)env * P.t.EW + w] = e.blk[w];
      P.b.blocked_in[(size_t)env * P.t.EW + w] = e.bin[w];
    }
  if (e.ring_dirty && lane < CG_LOG_RING)
    ((uint32_t*)(P.b.ring + (size_t)env * CG_LOG_RING * 2))[lane] = ((const uint32_t*)e.ring)[lane];
  if (COLD(XE && e.x_dirty)) {
    uint32_t* xg = P.b.extra + (size_t)env * (P.t.K + P.t.KW);
    const int nx = (int)((uint32_t)ie[CG_I_FLAGS] >> CG_E_NX_SHIFT);
    for (int j = lane; j < nx; j += WAVE) xg[j] = e.xk[j];
    for (int w = lane; w < P.t.KW; w += WAVE) xg[P.t.K + w] = e.xb[w];
  }
  if (lane == 0) {
    const uint32_t sticky = (uint32_t)ie[CG_I_FLAGS] & (CG_E_TOPO_OVF | CG_E_BUSY_SAT | CG_E_DET_PENDING | CG_E_UNPINNED);
    if (COLD(sticky != 0u) && P.o.status) atomicOr(P.o.status, sticky);
    int32_t* g = P.b.ienv + (size_t)env * CG_I_COUNT;
#pragma unroll
    for (int i = 0; i < CG_I_COUNT; ++i) g[i] = ie[i];
    double* gf = P.b.fenv + (size_t)env * CG_D_COUNT;
#pragma unroll
    for (int i = 0; i < CG_D_COUNT; ++i) gf[i] = fe[i];
  }
  STAMP(7);
#ifdef CG_STAMPS
  if (P.dbg && lane == 0) {
    P.dbg[(size_t)env * 16 + 8] = (unsigned long long)(long long)ie[CG_I_LAST_ATYPE];
    P.dbg[(size_t)env * 16 + 9] = (unsigned long long)mode | ((P.dbg[(size_t)env * 16] - t_entry) << 8);
  }
#endif
#undef P
}

#endif  // CG_TICK_HPP
