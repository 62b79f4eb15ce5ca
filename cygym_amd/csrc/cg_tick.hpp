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
// How the observation leaves the chip.  0: plain stores (lines stay dirty in the XCD's L2 and are written back when the
// launch ends: ~1.5 us of a one-tick launch's 3.2 us gap to the next launch, tools/exp_gap.py); 1: non-temporal; 2: sc1
// (written through as the kernel runs).
#ifndef CG_OBS_STORE
#define CG_OBS_STORE 0
#endif
typedef float cg_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void obs_store4(float4* p, float4 v) {
  [[maybe_unused]] const cg_f4v vv = {v.x, v.y, v.z, v.w};
  [[maybe_unused]] const uint64_t pa = (uint64_t)p;
#if CG_OBS_STORE == 1
  __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); __builtin_nontemporal_store(v.z, &p->z); __builtin_nontemporal_store(v.w, &p->w);
#elif CG_OBS_STORE == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(pa), "v"(vv) : "memory");
#elif CG_OBS_STORE == 3
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(pa), "v"(vv) : "memory");
#elif CG_OBS_STORE == 4
  asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(pa), "v"(vv) : "memory");
#else
  *p = v;
#endif
}
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
        obs_store4(&out4[3 * p + 0], make_float4(o[j].x, v[j].x, (float)(fa & 1u), a[j].x));
        obs_store4(&out4[3 * p + 1], make_float4((float)((fa >> 2) & 1u), (float)((fa >> 4) & 1u), o[j].y, v[j].y));
        obs_store4(&out4[3 * p + 2], make_float4((float)(fb & 1u), a[j].y, (float)((fb >> 2) & 1u), (float)((fb >> 4) & 1u)));
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
// The observation through an LDS stage (the WIDE per-tick kernel: one 16-wave workgroup per CU, LDS to spare): a one-tick
// launch leaves its 25 MB of observations dirty in the XCDs' L2s, and writing them back when the launch ends is ~1.5 us of
// the 3.2 us that separate two launches (tools/exp_gap.py).  Written THROUGH as the kernel runs (sc1) they cost that much
// less at the end -- but only as whole lines: in the pair-per-lane layout above a store instruction covers 3 KB in 16-byte
// pieces every 48 bytes, and as partial-line write-throughs that was 20 % slower than the plain stores.  So the three
// float4 of 64 pairs (3 KB) cross LDS once and leave as three store instructions of 1 KB each, lane after lane.
// PS pairs (48 bytes each) per step through `stage` (PS * 48 bytes of this wave's LDS): the WIDE kernel owns a 3 KB stage (64
// pairs: one pair per lane and step).  (Measured and dropped for the other per-tick kernels, which would have to borrow the
// env's 1.5 KB scratch area -- 32 pairs per step, four steps: 16384 x 256 +-0, 4096 x 64 -3 %, 4096 x 2048 -4 %: their launches are
// longer or their observations smaller, so the end-of-launch write-back weighs less than the extra LDS passes.)
__device__ __forceinline__ void write_obs_staged(const uint8_t* flags, const float* osv, const float* ver, const float* ano,
                                                 float* obs, int M, int lane, float4* stage, int PS) {
  const int npairs = M >> 1;   // (M even: checked by the caller)
  const uint16_t* F2 = (const uint16_t*)flags;
  const float2* os2 = (const float2*)osv;
  const float2* ve2 = (const float2*)ver;
  const float2* an2 = (const float2*)ano;
  for (int p0 = 0; p0 < npairs; p0 += PS) {
    for (int q = lane; q < PS; q += WAVE) {
      const int p = p0 + q, pc = p < npairs ? p : npairs - 1;
      const uint32_t f2 = F2[pc];
      const float2 o = os2[pc], v = ve2[pc], a = an2[pc];
      const uint32_t fa = f2 & 0xFFu, fb = f2 >> 8;
      stage[3 * q + 0] = make_float4(o.x, v.x, (float)(fa & 1u), a.x);
      stage[3 * q + 1] = make_float4((float)((fa >> 2) & 1u), (float)((fa >> 4) & 1u), o.y, v.y);
      stage[3 * q + 2] = make_float4((float)(fb & 1u), a.y, (float)((fb >> 2) & 1u), (float)((fb >> 4) & 1u));
    }
    wsync();
    const int n16 = (npairs - p0 < PS ? npairs - p0 : PS) * 3;   // 16-byte items of this step
    float4* out4 = (float4*)obs + 3 * p0;
    for (int i = lane; i < n16; i += WAVE) {
      const float4 w = stage[i];
      const cg_f4v wv = {w.x, w.y, w.z, w.w};
      const uint64_t pa = (uint64_t)(out4 + i);
      asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(pa), "v"(wv) : "memory");
    }
    wsync();
  }
}

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

#define CG_OBS_STAGE_BYTES 3072   // 64 pairs x 48 bytes
struct WaveAux { uint64_t* srcb; int32_t* park; };
// MAPS: the in-CSR columns and slot maps are staged in LDS too (the WIDE per-tick kernel, one 16-wave workgroup per CU)
template <bool MAPS, bool RT, class KP>   // RT: run-time size (comp_by may stay in global memory: one plane less in LDS)
__device__ __forceinline__ WaveAux env_setup(Env& e, uint8_t* smem, const KP& P, int M, int MC, int Mp, int MS,
                                             int wave, int lane, int env) {
  uint8_t* wb = smem + P.shared_lds + (size_t)wave * P.wave_lds;
  e.flags = wb; e.busy = wb + MS; e.wl = wb + 2 * MS; e.cby = wb + 3 * MS;
  const int n_planes = (RT && P.t.cby_global) ? 3 : 4;   // (comp_by left in global memory: Env::cby_g)
  e.scr = (uint32_t*)(wb + ((n_planes * MS + 15) & ~15));
  // scratch: [Mp] words + [Mp] halfwords at the compile-time sizes, [Mp] words at run-time sizes -- where the spread's T table
  // is 16 bits wide and shares them with its `cur` array (must match wave_lds_bytes on the host)
  e.blk = e.scr + (RT ? Mp : Mp + Mp / 2);
  e.bin = e.blk + ((P.t.EW + 3) & ~3);
  e.ring = (uint16_t*)(e.bin + ((P.t.EW + 3) & ~3));
  e.marks = (uint32_t*)(e.ring + 2 * CG_LOG_RING);
  WaveAux x;
  x.srcb = (uint64_t*)(e.marks + ((Mp / 32 + 2) & ~1));
  e.lsrc = (uint16_t*)(x.srcb + MC);
  e.devl = (int16_t*)(e.lsrc + Mp);
  x.park = (int32_t*)(wb + P.wave_lds - 128);   // [16 i32 + 3 f64] per-env scalars between fused ticks
  e.xk = (uint32_t*)(wb + P.wave_lds - 128 - P.t.x_bytes);
  // (below the extra-edge section; the host adds the bytes: 3 KB for the WIDE kernel, 1.5 KB -- all 32 pairs -- at 64 devices)
  e.obs_stage = (float4*)(wb + P.wave_lds - 128 - P.t.x_bytes - (M == 64 ? CG_OBS_STAGE_BYTES / 2 : CG_OBS_STAGE_BYTES));
  e.xb = e.xk + P.t.K;
  e.xmo = (uint64_t*)(e.xb + ((P.t.KW + 1) & ~1));
  const bool lists_global = RT && P.t.lists_global;   // device list / extra-edge list / in-row bounds read where they lie (see choose_launch)
  if (lists_global) { e.xmo = (uint64_t*)e.xk; e.xk = nullptr; e.xb = nullptr; e.devl = nullptr; }   // (the tick body points them at the env's global rows)
  e.xmi = e.xmo + MC;
  e.K = P.t.K;
#ifndef CG_AFFINE_BLOB
#define CG_AFFINE_BLOB 1
#endif
  if constexpr (CG_AFFINE_BLOB && !RT) {
    // Compile-time device count: the blob's twelve section offsets (cygym_create: every section padded to 16 bytes) are all
    // "o_dst + constant" or "+ k * padded column bytes" -- two run-time scalars instead of twelve kept (spilled, restored) for the
    // whole kernel, and the constants fold into the instructions' offset fields.
    const int A_M = (M + 15) & ~15, A_P = (2 * (M + 1) + 15) & ~15, A_F = (4 * M + 15) & ~15;
    const int o_dst = P.t.o_dst, a_e = o_dst - A_P;   // a_e: padded bytes of one u16[E] column (o_ocol == A_P)
    const uint8_t* sd = smem + o_dst;
    e.optr = (const uint16_t*)smem; e.ocol = (const uint16_t*)(smem + A_P);
    e.dst = sd; e.vul = sd + A_M; e.nap = sd + 2 * A_M;
    e.iptr_l = (const uint16_t*)(sd + 3 * A_M);
    e.osv = (const float*)(sd + 3 * A_M + A_P); e.ver = (const float*)(sd + 3 * A_M + A_P + A_F); e.ano = (const float*)(sd + 3 * A_M + A_P + 2 * A_F);   // valid when P.t.in_lds
    const int o_icol = o_dst + 3 * A_M + A_P + 3 * A_F;
    const uint8_t* mb = MAPS ? (const uint8_t*)smem : P.t.blob;   // MAPS: block / unblock picks no longer pay a global-memory hop per pass
    e.icol_g = (const uint16_t*)(mb + o_icol); e.ieid_g = (const uint16_t*)(mb + o_icol + a_e); e.oeid_g = (const uint16_t*)(mb + o_icol + 2 * a_e);
  } else {
  e.optr = (const uint16_t*)(smem + P.t.o_optr); e.ocol = (const uint16_t*)(smem + P.t.o_ocol);
  e.osv = (const float*)(smem + P.t.o_os); e.ver = (const float*)(smem + P.t.o_ver); e.ano = (const float*)(smem + P.t.o_ano);   // valid when P.t.in_lds
  e.dst = smem + P.t.o_dst; e.vul = smem + P.t.o_vul; e.nap = smem + P.t.o_nap;
  e.iptr_l = lists_global ? (const uint16_t*)(P.t.blob + P.t.o_iptr) : (const uint16_t*)(smem + P.t.o_iptr);
  if constexpr (MAPS) {   // LDS: block / unblock picks no longer pay a global-memory hop per pass
    e.icol_g = (const uint16_t*)(smem + P.t.o_icol); e.ieid_g = (const uint16_t*)(smem + P.t.o_ieid); e.oeid_g = (const uint16_t*)(smem + P.t.o_oeid);
  } else {
    e.icol_g = (const uint16_t*)(P.t.blob + P.t.o_icol); e.ieid_g = (const uint16_t*)(P.t.blob + P.t.o_ieid); e.oeid_g = (const uint16_t*)(P.t.blob + P.t.o_oeid);
  }
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
#include "cg_tick_body.inc"
}

#endif  // CG_TICK_HPP
