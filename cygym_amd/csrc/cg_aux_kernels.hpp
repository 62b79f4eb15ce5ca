// cg_aux_kernels.hpp -- reset / randomize / derive / observe kernels and the synthetic action script of bench.py.
// Part of the device code gathered by cg_device.hpp (included inside namespace cygym_k, in order); not a standalone header.
#ifndef CG_AUX_KERNELS_HPP
#define CG_AUX_KERNELS_HPP

// ---------------- reset / randomize / observe / action script ----------------
__global__ void reset_kernel(KParams P, const int32_t* env_ids, int n) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n) return;
  const int env = env_ids ? env_ids[wave] : wave;
  if (env < 0 || env >= P.n_envs) return;
  const int M = P.t.M;
  const int si = P.snap.n_envs == 1 ? 0 : env;
  const size_t so = (size_t)env * 4 * M, ss = (size_t)si * 4 * M;
  int32_t tick = P.b.ienv[(size_t)env * CG_I_COUNT + CG_I_RNG_TICK];
  for (int i = lane; i < 4 * M; i += WAVE) {
    P.b.live[so + i] = P.snap.live[ss + i];
    P.b.stash[so + i] = P.snap.stash[ss + i];
  }
  for (int w = lane; w < P.t.EW; w += WAVE) {
    P.b.blocked[(size_t)env * P.t.EW + w] = P.snap.blocked[(size_t)si * P.t.EW + w];
    P.b.blocked_in[(size_t)env * P.t.EW + w] = P.snap.blocked_in[(size_t)si * P.t.EW + w];
  }
  if (lane < CG_LOG_RING)
    ((uint32_t*)(P.b.ring + (size_t)env * CG_LOG_RING * 2))[lane] = ((const uint32_t*)(P.snap.ring + (size_t)si * CG_LOG_RING * 2))[lane];
  const int XW = P.t.K + P.t.KW;
  if (P.t.K > 0 && P.snap.extra)
    for (int w = lane; w < XW; w += WAVE) P.b.extra[(size_t)env * XW + w] = P.snap.extra[(size_t)si * XW + w];
  if (P.b.hist && P.snap.hist)   // the pickled env carries its logger and detector too (volt_typhoon_env.py:1904-1936)
    for (int i = lane; i < CG_HIST_RING; i += WAVE)
      ((uint32_t*)(P.b.hist + (size_t)env * CG_HIST_RING * 2))[i] = ((const uint32_t*)(P.snap.hist + (size_t)si * CG_HIST_RING * 2))[i];
  if (P.b.forest && P.snap.forest)
    for (int i = lane; i < CG_FOREST_WORDS; i += WAVE)
      P.b.forest[(size_t)env * CG_FOREST_WORDS + i] = P.snap.forest[(size_t)si * CG_FOREST_WORDS + i];
  if (P.b.anomaly && P.snap.anomaly)
    for (int i = lane; i < M; i += WAVE) P.b.anomaly[(size_t)env * M + i] = P.snap.anomaly[(size_t)si * M + i];
  if (lane < CG_I_COUNT) {
    int32_t v = P.snap.ienv[(size_t)si * CG_I_COUNT + lane];
    if (lane == CG_I_RNG_TICK) v = tick;   // the draw counter is monotone across episodes
    if (lane == CG_I_FLAGS && !(P.t.K > 0 && P.snap.extra)) v &= 0xFFFF;   // no extra-edge list to restore
    P.b.ienv[(size_t)env * CG_I_COUNT + lane] = v;
  }
  if (lane < CG_D_COUNT) P.b.fenv[(size_t)env * CG_D_COUNT + lane] = P.snap.fenv[(size_t)si * CG_D_COUNT + lane];
}

// randomize_compromise_and_ownership volt_typhoon_env.py:330-383; wave per env, global memory only.
__global__ void randomize_kernel(KParams P, const int32_t* env_ids, int n, uint32_t* keybuf) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= n) return;
  const int env = env_ids ? env_ids[wave] : wave;
  if (env < 0 || env >= P.n_envs) return;
  const int M = P.t.M, MC = P.t.MC;
  uint8_t* flags = P.b.live + (size_t)env * 4 * M;
  int32_t* ie = P.b.ienv + (size_t)env * CG_I_COUNT;
  const uint32_t tick = (uint32_t)ie[CG_I_RNG_TICK];
  const uint32_t env_id = (uint32_t)(P.c.env_id_base + env);
  uint32_t* key = keybuf + (size_t)wave * MC * WAVE;
  int cnt = 0, k_owned = 0, k_comp = 0;
  for (int c = 0; c < MC; ++c) {
    int d = c * WAVE + lane;
    bool el = d < M && !(flags[d] & CG_F_NYA) && !(P.t.dstatic[d] & CG_D_DC);
    if (d < MC * WAVE) key[d] = el ? cg_draw(P.c.seed, env_id, tick, CG_SITE_SHUFFLE, d, 0) : 0u;
    cnt += __popcll(ballot(el));
    k_owned += __popcll(ballot(el && (flags[d] & CG_F_OWNED)));
    k_comp += __popcll(ballot(el && (flags[d] & CG_F_COMP)));
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  if (lane == 0) ie[CG_I_RNG_TICK] = (int32_t)(tick + 1);
  if (lane == 0) ie[CG_I_FLAGS] &= ~CG_E_STAR_OK;
  if (cnt == 0 || (k_owned == 0 && k_comp == 0)) return;
  int extra = k_comp - k_owned; if (extra < 0) extra = 0;
  for (int c = 0; c < MC; ++c) {
    int d = c * WAVE + lane;
    bool el = d < M && !(flags[d] & CG_F_NYA) && !(P.t.dstatic[d] & CG_D_DC);
    if (!el) continue;
    uint32_t kd = key[d];
    int rank = 0;
    for (int o = 0; o < M; ++o) {
      bool eo = !(flags[o] & CG_F_NYA) && !(P.t.dstatic[o] & CG_D_DC);
      uint32_t ko = key[o];
      rank += (eo && (ko < kd || (ko == kd && o < d))) ? 1 : 0;
    }
    uint8_t f = (uint8_t)(flags[d] & ~(CG_F_OWNED | CG_F_COMP | CG_F_KNOWN));
    if (rank < k_owned) f |= (CG_F_OWNED | CG_F_COMP | CG_F_KNOWN);
    else if (rank < k_owned + extra) f |= (CG_F_COMP | CG_F_KNOWN);
    // flags are rewritten after every lane has read the eligibility bits (NYA/DC do not change)
    flags[d] = f;
  }
}

// blocked_in[j] = blocked[in_eid[j]]: the derived in-order mirror of the blocked bits (wave per env)
__global__ void derive_kernel(KParams P, cygym_buffers bufs) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= bufs.n_envs) return;
  const uint32_t* blk = bufs.blocked + (size_t)wave * P.t.EW;
  uint32_t* bin = bufs.blocked_in + (size_t)wave * P.t.EW;
  for (int j0 = 0; j0 < P.t.EW * 32; j0 += WAVE) {
    const int j = j0 + lane;
    bool bit = false;
    if (j < P.t.E) { const int k = P.t.in_eid[j]; bit = (blk[k >> 5] >> (k & 31)) & 1u; }
    const uint64_t m = __ballot(bit);
    if (lane == 0) {
      bin[j0 >> 5] = (uint32_t)m;
      if ((j0 >> 5) + 1 < P.t.EW) bin[(j0 >> 5) + 1] = (uint32_t)(m >> 32);
    }
  }
}

// role views: CyberDefenseEnv.py:146-257
__global__ void observe_kernel(KParams P, int role, float* out) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= P.n_envs) return;
  const int env = wave, M = P.t.M;
  const uint8_t* flags = P.b.live + (size_t)env * 4 * M;
  if (role == 0 || role == 1) {
    float* o = out + (size_t)env * 6 * M;
    for (int i = lane; i < 6 * M; i += WAVE) {
      int d = i / 6, col = i - d * 6;
      uint8_t f = flags[d];
      float x = col == 0 ? P.t.os_val[d] : col == 1 ? P.t.version[d] : col == 2 ? ((f & CG_F_COMP) ? 1.f : 0.f)
              : col == 3 ? (P.b.anomaly ? P.b.anomaly[(size_t)env * M + d] : P.t.anomaly[d]) : col == 4 ? ((f & CG_F_KNOWN) ? 1.f : 0.f) : ((f & CG_F_NYA) ? 1.f : 0.f);
      if (role == 1 && ((f & CG_F_NYA) || !(f & CG_F_OWNED) || col == 2)) x = -1.f;
      o[i] = x;
    }
  } else {
    const int W = 4 * M + P.c.max_exploits;
    float* o = out + (size_t)env * W;
    for (int i = lane; i < W; i += WAVE) {
      float x;
      if (i < 4 * M) {
        int d = i >> 2, col = i & 3;
        uint8_t f = flags[d];
        bool vis = (f & CG_F_KNOWN) && !(f & CG_F_NYA) && (f & CG_F_OWNED);
        x = !vis ? -1.f : col == 0 ? P.t.os_val[d] : col == 1 ? P.t.version[d]
          : col == 2 ? ((f & CG_F_COMP) ? 1.f : 0.f) : ((f & CG_F_KNOWN) ? 1.f : 0.f);
      } else {
        x = (i - 4 * M) < P.t.X ? 1.f : 0.f;
      }
      o[i] = x;
    }
  }
}

#include "cg_decode.hpp"      // decode of one row by one wave + the whole-actor kernel (templates: also used by cg_inst_actor.hip)

// cygym_write_actions: one wave per source row.  A device mask is compacted to the ascending id list with ballots
// (rank of a chosen device = chosen devices below it), the first max_devs of them; entries past the count are zeroed.
__global__ void write_actions_kernel(cygym_action_rows src, cygym_actions dst, int M, int n_envs) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= src.n) return;
  const int row = src.rows ? src.rows[wave] : wave;
  if (row < 0 || row >= n_envs) return;
  const int G = dst.max_groups, L = dst.max_devs;
  int16_t* out = const_cast<int16_t*>(dst.dev_idx) + (size_t)row * L;
  int cnt;
  if (src.dev_mask) {
    const uint8_t* mk = src.dev_mask + (size_t)wave * M;
    int base = 0;
    for (int d0 = 0; d0 < M; d0 += WAVE) {
      const int d = d0 + lane;
      const bool on = d < M && mk[d] != 0;
      const uint64_t m = __ballot(on);
      const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (on && pos < L) out[pos] = (int16_t)d;
      base += __popcll(m);
    }
    cnt = base < L ? base : L;
  } else {
    cnt = src.dev_cnt[wave];
    if (cnt > L) cnt = L;
    if (cnt < 0) cnt = 0;
    const int16_t* in = src.dev_idx + (size_t)wave * L;
    for (int q = lane; q < cnt; q += WAVE) out[q] = in[q];
  }
  for (int q = cnt + lane; q < L; q += WAVE) out[q] = 0;
  if (lane == 0) {
    const int ex = src.exploit[wave];
    const_cast<int32_t*>(dst.atype)[(size_t)row * G] = src.atype[wave];
    const_cast<int32_t*>(dst.exploit)[(size_t)row * G * CG_MAX_EXPLOITS] = ex;
    const_cast<int32_t*>(dst.n_exploit)[(size_t)row * G] = ex >= 0 ? 1 : 0;
    const_cast<int32_t*>(dst.app)[(size_t)row * G] = src.app[wave];
    const_cast<int32_t*>(dst.dev_cnt)[(size_t)row * G] = cnt;
  }
}

// cygym_group_actions (IPPO.py:560-572 for a batch): one wave per row.  For every action type in ascending order the
// devices that sampled it are ranked with ballots (ascending id = list order); a single-device type keeps the r-th of
// them, r uniform from the addressed Philox draw.  `ty` = the row's types (global memory or LDS), `vis` an explicit mask or
// nullptr = the role's visibility read off the flag plane `fl`.
__device__ __forceinline__ void group_row(const uint8_t* ty, const uint8_t* vis, const uint8_t* fl, const uint32_t want, const int M,
                                          const int n_types, const int noop, const uint32_t single_mask, const int ex, const int app,
                                          const cygym_actions& dst, const int row, const uint32_t tick, const uint64_t seed,
                                          const int64_t env_id_base, uint32_t* status, const int lane) {
  const int G = dst.max_groups, L = dst.max_devs;
  int16_t* out = const_cast<int16_t*>(dst.dev_idx) + (size_t)row * L;
  int32_t* o_at = const_cast<int32_t*>(dst.atype) + (size_t)row * G;
  int32_t* o_ne = const_cast<int32_t*>(dst.n_exploit) + (size_t)row * G;
  int32_t* o_ex = const_cast<int32_t*>(dst.exploit) + (size_t)row * G * CG_MAX_EXPLOITS;
  int32_t* o_app = const_cast<int32_t*>(dst.app) + (size_t)row * G;
  int32_t* o_cnt = const_cast<int32_t*>(dst.dev_cnt) + (size_t)row * G;
  int g = 0, base = 0;
  bool cut = false;
  for (int t = 0; t < n_types; ++t) {
    if (t == noop) continue;
    int total = 0;
    for (int d0 = 0; d0 < M; d0 += WAVE) {
      const int d = d0 + lane;
      const bool on = d < M && ty[d] == t && (vis ? vis[d] != 0 : ((fl[d] & (want | CG_F_NYA)) == want));
      total += __popcll(__ballot(on));
    }
    if (total == 0) continue;
    if (g >= G) { cut = true; break; }
    int pick = -1;   // single-device type: index of the chosen device among the type's devices
    if ((single_mask >> t) & 1u) {
      const cg_u32x4 r = cg_philox4x32_10((uint32_t)(env_id_base + row), tick, CG_SITE_GROUP_PICK, (uint32_t)t, (uint32_t)seed, (uint32_t)(seed >> 32));
      pick = (int)cg_index(r.v[0], (uint32_t)total);
    }
    int seen = 0;
    for (int d0 = 0; d0 < M; d0 += WAVE) {
      const int d = d0 + lane;
      const bool on = d < M && ty[d] == t && (vis ? vis[d] != 0 : ((fl[d] & (want | CG_F_NYA)) == want));
      const uint64_t m = __ballot(on);
      const int rank = seen + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
      if (on) {
        const int pos = pick < 0 ? rank : (rank == pick ? 0 : -1);
        if (pos >= 0 && base + pos < L) out[base + pos] = (int16_t)d;
      }
      seen += __popcll(m);
    }
    int cnt = pick < 0 ? total : 1;
    if (base + cnt > L) { cnt = L - base; cut = true; }
    if (lane == 0) { o_at[g] = t; o_ne[g] = 1; o_ex[(size_t)g * CG_MAX_EXPLOITS] = ex; o_app[g] = app; o_cnt[g] = cnt; }
    base += cnt;
    ++g;
  }
  if (g == 0) {   // [(noop, [0], [], 0)]
    if (lane == 0) { o_at[0] = noop; o_ne[0] = 1; o_ex[0] = 0; o_app[0] = 0; o_cnt[0] = 0; }
    g = 1;
  }
  if (lane == 0) {
    const_cast<int32_t*>(dst.n_groups)[row] = g;
    if (cut && status) atomicOr(status, CG_DECODE_TRUNCATED);
  }
}
__global__ void group_actions_kernel(cygym_device_types src, cygym_actions dst, int M, int n_envs, const uint8_t* live,
                                     const int32_t* ienv, uint64_t seed, int64_t env_id_base) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= src.n) return;
  const int row = src.rows ? src.rows[wave] : wave;
  if (row < 0 || row >= n_envs) return;
  const uint32_t want = src.role == 2 ? (CG_F_KNOWN | CG_F_OWNED) : CG_F_OWNED;
  const uint32_t tick = (uint32_t)ienv[(size_t)row * CG_I_COUNT + CG_I_RNG_TICK];
  group_row(src.types + (size_t)wave * M, src.visible ? src.visible + (size_t)wave * M : nullptr, live + (size_t)row * 4 * M, want, M, src.n_types,
            src.noop, src.single_mask, src.exploit ? src.exploit[wave] : 0, src.app ? src.app[wave] : 0, dst, row, tick, seed, env_id_base,
            src.status, lane);
}

// cygym_sample_group_actions (IPPO.py:524-572 for a batch): one wave per row, a lane per device.  Sampling = the inverse CDF
// of softmax(logits) walked with u = addressed Philox draw / 2^32; the sampled types stay in LDS for the grouping.
constexpr int SAMPLE_WPB = 4;
__device__ __forceinline__ int sample_head(const float* l, const int K, const uint32_t u32, const bool greedy, float& logp) {
  float mx = -__builtin_inff();
  int am = 0;
  for (int k = 0; k < K; ++k) { const float x = l[k]; if (x > mx) { mx = x; am = k; } }   // (first maximum)
  float S = 0.f;
  for (int k = 0; k < K; ++k) S += __expf(l[k] - mx);
  int pick = am;
  if (!greedy) {
    const float target = (float)u32 * (1.0f / 4294967296.0f) * S;
    float acc = 0.f;
    pick = K - 1;
    for (int k = 0; k < K; ++k) { acc += __expf(l[k] - mx); if (acc > target) { pick = k; break; } }
  }
  logp = l[pick] - mx - __logf(S);
  return pick;
}
__global__ __launch_bounds__(SAMPLE_WPB * WAVE) void sample_group_actions_kernel(cygym_device_logits src, cygym_actions dst, int M, int n_envs,
                                                                                 const uint8_t* live, const int32_t* ienv, uint64_t seed,
                                                                                 int64_t env_id_base) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave = blockIdx.x * SAMPLE_WPB + wv;
  if (wave >= src.n) return;
  const int row = src.rows ? src.rows[wave] : wave;
  if (row < 0 || row >= n_envs) return;
  const int Mp = (M + 63) & ~63;
  const int K = src.n_types;
  const size_t per_wave = (size_t)Mp + (size_t)WAVE * K * sizeof(float);   // sampled types of the row + the logits of 64 devices
  uint8_t* ty = smem + (size_t)wv * per_wave;
  float* lg = reinterpret_cast<float*>(ty + Mp);
  const uint8_t* fl = live + (size_t)row * 4 * M;
  const uint32_t want = src.role == 2 ? (CG_F_KNOWN | CG_F_OWNED) : CG_F_OWNED;
  const uint32_t tick = (uint32_t)ienv[(size_t)row * CG_I_COUNT + CG_I_RNG_TICK];
  const uint32_t env_g = (uint32_t)(env_id_base + row), k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const bool greedy = src.greedy != 0;
  float lp = 0.f;
  for (int d0 = 0; d0 < M; d0 += WAVE) {
    // the logits of 64 devices: coalesced into LDS (a lane's own K logits lie 4 K bytes from its neighbour's), then every lane
    // walks its own row there
    const int nd = M - d0 < WAVE ? M - d0 : WAVE;
    const float* gl = src.logits + ((size_t)wave * M + d0) * K;
    wsync();
    for (int i = lane; i < nd * K; i += WAVE) lg[i] = gl[i];
    wsync();
    const int d = d0 + lane;
    int t = 0;
    if (d < M) {
      const bool vis = (fl[d] & (want | CG_F_NYA)) == want;
      if (vis) {   // never samples an invisible device: label 0, no log-probability (IPPO.py:530-537)
        const cg_u32x4 r = cg_philox4x32_10(env_g, tick, CG_SITE_SAMPLE, (uint32_t)d & 0xFFFFu, k0, k1);
        float l1;
        t = sample_head(lg + lane * K, K, r.v[0], greedy, l1);
        lp += l1;
      }
      ty[d] = (uint8_t)t;
      src.types_out[(size_t)wave * M + d] = (uint8_t)t;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) lp += __shfl_xor(lp, off);
  int ex = 0, app = 0;
  if (src.exp_logits && src.n_exp > 0) {   // (every lane walks the same few entries)
    const cg_u32x4 r = cg_philox4x32_10(env_g, tick, CG_SITE_SAMPLE, 1u << 16, k0, k1);
    float l1;
    ex = sample_head(src.exp_logits + (size_t)wave * src.n_exp, src.n_exp, r.v[0], greedy, l1);
    lp += l1;
  }
  if (src.app_logits && src.n_app > 0) {
    const cg_u32x4 r = cg_philox4x32_10(env_g, tick, CG_SITE_SAMPLE, 2u << 16, k0, k1);
    float l1;
    app = sample_head(src.app_logits + (size_t)wave * src.n_app, src.n_app, r.v[0], greedy, l1);
    lp += l1;
  }
  if (lane == 0) {
    if (src.logp_out) src.logp_out[wave] = lp;
    if (src.exp_out) src.exp_out[wave] = ex;
    if (src.app_out) src.app_out[wave] = app;
  }
  wsync();   // (the wave's own LDS bytes: written above by other lanes)
  group_row(ty, nullptr, fl, want, M, K, src.noop, src.single_mask, ex, app, dst, row, tick, seed, env_id_base, src.status, lane);
}

// cygym_decode_actions (do_agent.py:970-998 for a batch): one wave per row.  argmax = first maximum (np.argmax).
__device__ __forceinline__ int wave_argmax(const float* v, int n, int lane) {
  float best = -__builtin_inff();
  int bi = 0x7FFFFFFF;
  for (int i = lane; i < n; i += WAVE) { const float x = v[i]; if (x > best) { best = x; bi = i; } }   // (ascending i: first maximum per lane)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ob = __shfl_xor(best, off);
    const int oi = __shfl_xor(bi, off);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
  }
  return bi == 0x7FFFFFFF ? 0 : bi;
}
__global__ void decode_actions_kernel(cygym_action_vectors src, cygym_actions dst, int n_envs, const int32_t* ienv,
                                      uint64_t seed, int64_t env_id_base) {
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wave >= src.n) return;
  const int row = src.rows ? src.rows[wave] : wave;
  if (row < 0 || row >= n_envs) return;
  const int G = dst.max_groups, L = dst.max_devs, M = src.n_devices;
  const float* v = src.vec + (size_t)wave * src.stride;
  int at = src.n_types > 0 ? wave_argmax(v, src.n_types, lane) : 0;
  if (src.epsilon_thr && src.n_types > 0) {   // epsilon-greedy (do_agent.py:972-973)
    const uint32_t tick = (uint32_t)ienv[(size_t)row * CG_I_COUNT + CG_I_RNG_TICK];
    const cg_u32x4 r = cg_philox4x32_10((uint32_t)(env_id_base + row), tick, CG_SITE_EPS_TYPE, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    if ((uint64_t)r.v[0] < src.epsilon_thr) at = (int)cg_index(r.v[1], (uint32_t)src.n_types);
  }
  if (src.type_map && src.n_types > 0) at = src.type_map[at];
  const float* dv = v + src.n_types;
  int16_t* out = const_cast<int16_t*>(dst.dev_idx) + (size_t)row * L;
  int base = 0;
  for (int d0 = 0; d0 < M; d0 += WAVE) {
    const int d = d0 + lane;
    const bool on = d < M && dv[d] > 0.f;
    const uint64_t m = __ballot(on);
    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (on && pos < L) out[pos] = (int16_t)d;
    base += __popcll(m);
  }
  const int cnt = base < L ? base : L;
  for (int q = cnt + lane; q < L; q += WAVE) out[q] = 0;
  const int ex = src.n_exploits > 0 ? wave_argmax(dv + M, src.n_exploits, lane) : 0;
  const int app = src.n_apps > 0 ? wave_argmax(dv + M + src.n_exploits, src.n_apps, lane) : 0;
  if (lane == 0) {
    const_cast<int32_t*>(dst.atype)[(size_t)row * G] = at;
    const_cast<int32_t*>(dst.exploit)[(size_t)row * G * CG_MAX_EXPLOITS] = ex;
    const_cast<int32_t*>(dst.n_exploit)[(size_t)row * G] = 1;
    const_cast<int32_t*>(dst.app)[(size_t)row * G] = app;
    const_cast<int32_t*>(dst.dev_cnt)[(size_t)row * G] = cnt;
    if (base > L && src.status) atomicOr(src.status, CG_DECODE_TRUNCATED);
  }
}

// cygym_actor_head_decode: last Linear layer of the actor + decode_action, fused.  A workgroup of HEAD_WAVES waves stages
// a 64-row k-slab of the k-major weight matrix in LDS (lane j reads the weights of consecutive outputs j: conflict-free);
// every wave owns ONE row (16 waves per workgroup = 4 per SIMD: the LDS and reduction latencies of one row hide behind the
// other rows'); lane j accumulates outputs j, j + 64, ... (HEAD_OPL per lane) with the hidden activation of step k
// broadcast from the lane that holds it (v_readlane: an SGPR operand, no LDS traffic).  The action vector of a row lives
// in registers only; its arg-maxima are wave reductions on the DPP path over (order-preserving value bits, ~index) pairs.
template <int HEAD_OPL>
__global__ __launch_bounds__(HEAD_WAVES * WAVE) void actor_head_kernel(cygym_actor_head hd, cygym_action_vectors src, cygym_actions dst,
                                                                        int n_envs, const int32_t* ienv, uint64_t seed, int64_t env_id_base) {
  extern __shared__ __align__(16) uint8_t smem[];
  float* Wt = (float*)smem;   // [HEAD_KC][n_out_p]: a k-slab of the k-major weight matrix, copied as it lies
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_out = src.n_types + src.n_devices + src.n_exploits + src.n_apps;
  const int n_out_p = HEAD_OPL * WAVE;
  const int H = hd.H;
  const int srow = blockIdx.x * HEAD_WAVES + wave;   // this wave's source row
  const bool have = srow < src.n;
  if (hd.n_groups > 1) {   // a population of actors: this workgroup's 16 rows belong to ONE of them (rows_per_group % 16 == 0)
    const int grp = (blockIdx.x * HEAD_WAVES) / hd.rows_per_group;
    hd.weight_t += (size_t)grp * hd.H * hd.weight_pitch;
    if (hd.bias) hd.bias += (size_t)grp * (src.n_types + src.n_devices + src.n_exploits + src.n_apps);
  }
  float acc[HEAD_OPL];
#pragma unroll
  for (int i = 0; i < HEAD_OPL; ++i) {
    const int j = lane + i * WAVE;
    acc[i] = (hd.bias && j < n_out) ? hd.bias[j] : 0.f;
  }
  for (int k0 = 0; k0 < H; k0 += HEAD_KC) {
    const int kc = H - k0 < HEAD_KC ? H - k0 : HEAD_KC;
    const float hv = (have && lane < kc) ? hd.hidden[(size_t)srow * hd.hidden_stride + k0 + lane] : 0.f;
    __syncthreads();
    // stage rows k0 .. k0 + kc of weight_t (row pitch n_out_p: 16-byte aligned rows as wide as the LDS rows) -> Wt, a
    // plain copy, 16 bytes per lane, every load of a thread in flight at once
    {
      const int total4 = kc * n_out_p / 4;
      const float4* wsrc = (const float4*)(hd.weight_t + (size_t)k0 * n_out_p);
      float4* wdst = (float4*)Wt;
      constexpr int STRIDE = HEAD_WAVES * WAVE, UF = (HEAD_KC * HEAD_OPL * WAVE / 4 + STRIDE - 1) / STRIDE;
      float4 x[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u) { const int i = threadIdx.x + u * STRIDE; x[u] = wsrc[i < total4 ? i : total4 - 1]; }
#pragma unroll
      for (int u = 0; u < UF; ++u) { const int i = threadIdx.x + u * STRIDE; if (i < total4) wdst[i] = x[u]; }
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < kc; ++k) {
      const float hk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(hv), k));   // (k is uniform: an SGPR operand)
#pragma unroll
      for (int i = 0; i < HEAD_OPL; ++i) acc[i] = __builtin_fmaf(hk, Wt[k * n_out_p + lane + i * WAVE], acc[i]);
    }
  }
  if (!have) return;
  const int row = src.rows ? src.rows[srow] : srow;
  if (row < 0 || row >= n_envs) return;
  const int G = dst.max_groups, L = dst.max_devs, M = src.n_devices, nt = src.n_types;
  float v[HEAD_OPL];
#pragma unroll
  for (int i = 0; i < HEAD_OPL; ++i) v[i] = hd.tanh_out ? tanhf(acc[i]) : acc[i];
  // argmax of the outputs in [lo, hi) (first maximum, like np.argmax): per lane over its registers, then across the wave
  auto range_argmax = [&](int lo, int hi) -> int {
    uint32_t bh = 0u, bl = 0u;   // (0, 0): below every real candidate (order bits of a finite float are >= 0x00800000)
#pragma unroll
    for (int i = 0; i < HEAD_OPL; ++i) {
      const int j = lane + i * WAVE;
      const uint32_t ob = float_order_bits(v[i]);
      if (j >= lo && j < hi && ob > bh) { bh = ob; bl = ~(uint32_t)(j - lo); }   // ascending j per lane: first maximum
    }
    dpp_pair_max(bh, bl);
    const uint32_t rl = (uint32_t)__builtin_amdgcn_readlane((int)bl, 63), rh = (uint32_t)__builtin_amdgcn_readlane((int)bh, 63);
    return rh == 0u ? 0 : (int)~rl;
  };
  int at = nt > 0 ? range_argmax(0, nt) : 0;
  if (src.epsilon_thr && nt > 0) {   // epsilon-greedy (do_agent.py:972-973)
    const uint32_t tick = (uint32_t)ienv[(size_t)row * CG_I_COUNT + CG_I_RNG_TICK];
    const cg_u32x4 rr = cg_philox4x32_10((uint32_t)(env_id_base + row), tick, CG_SITE_EPS_TYPE, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    if ((uint64_t)rr.v[0] < src.epsilon_thr) at = (int)cg_index(rr.v[1], (uint32_t)nt);
  }
  if (src.type_map && nt > 0) at = src.type_map[at];
  int16_t* out = const_cast<int16_t*>(dst.dev_idx) + (size_t)row * L;
  int base = 0;
#pragma unroll
  for (int i = 0; i < HEAD_OPL; ++i) {   // (i, lane) ascending == output index ascending == device id ascending
    const int d = lane + i * WAVE - nt;
    const bool on = d >= 0 && d < M && v[i] > 0.f;
    const uint64_t m = __ballot(on);
    const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    if (on && pos < L) out[pos] = (int16_t)d;
    base += __popcll(m);
  }
  const int cnt = base < L ? base : L;
  for (int q = cnt + lane; q < L; q += WAVE) out[q] = 0;
  const int ex = src.n_exploits > 0 ? range_argmax(nt + M, nt + M + src.n_exploits) : 0;
  const int app = src.n_apps > 0 ? range_argmax(nt + M + src.n_exploits, n_out) : 0;
  if (lane == 0) {
    const_cast<int32_t*>(dst.atype)[(size_t)row * G] = at;
    const_cast<int32_t*>(dst.exploit)[(size_t)row * G * CG_MAX_EXPLOITS] = ex;
    const_cast<int32_t*>(dst.n_exploit)[(size_t)row * G] = 1;
    const_cast<int32_t*>(dst.app)[(size_t)row * G] = app;
    const_cast<int32_t*>(dst.dev_cnt)[(size_t)row * G] = cnt;
    if (base > L && src.status) atomicOr(src.status, CG_DECODE_TRUNCATED);
  }
}

// The same on the matrix cores (H % 4 == 0): a workgroup of 16 waves owns 16 rows; the [16 x H] x [H x n_out_p] product
// is cut into 16 x 16 output tiles (v_mfma_f32_16x16x4_f32, fp32 in, fp32 accumulate), tile t on wave t % 16; the A
// fragments (hidden activations) and B fragments (weights, k-major: 64-byte runs per k) come straight from global
// memory / L2 -- every weight is used once per workgroup, so there is nothing to stage; the 16 x n_out_p outputs pass
// through LDS once to get each row into one wave, which decodes it as above.
template <int HEAD_OPL>
__global__ __launch_bounds__(16 * WAVE) void actor_head_mfma_kernel(cygym_actor_head hd, cygym_action_vectors src, cygym_actions dst,
                                                                    int n_envs, const int32_t* ienv, uint64_t seed, int64_t env_id_base) {
  extern __shared__ __align__(16) uint8_t smem[];
  float* outs = (float*)smem;   // [16][n_out_p]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n_out = src.n_types + src.n_devices + src.n_exploits + src.n_apps;
  constexpr int n_out_p = HEAD_OPL * WAVE, n_tiles = n_out_p / 16;
  const int H = hd.H, nt = src.n_types;
  const int row0 = blockIdx.x * 16;
  if (hd.n_groups > 1) {   // a population of actors: this workgroup's 16 rows belong to ONE of them (rows_per_group % 16 == 0)
    const int grp = row0 / hd.rows_per_group;
    hd.weight_t += (size_t)grp * H * n_out_p;
    if (hd.bias) hd.bias += (size_t)grp * n_out;
  }
  // Everything the decode of this wave's row will need is requested NOW, ahead of the product: the row id, the env's rng
  // tick (epsilon-greedy), the type map (one entry per lane) and the bias -- a wave decodes one row, so a chain of
  // dependent global loads at the end (row id -> tick, arg-max -> type map) would be the kernel's whole duration.
  const int srow = row0 + wave;
  const bool have = srow < src.n;
  int row = have ? (src.rows ? src.rows[srow] : srow) : -1;
  if (row >= n_envs) row = -1;
  uint32_t tick = 0;
  if (src.epsilon_thr && row >= 0) tick = (uint32_t)ienv[(size_t)row * CG_I_COUNT + CG_I_RNG_TICK];
  int tmap = lane;
  if (src.type_map && lane < nt) tmap = src.type_map[lane];
  float bias_r[HEAD_OPL];
#pragma unroll
  for (int i = 0; i < HEAD_OPL; ++i) { const int j = lane + i * WAVE; bias_r[i] = (hd.bias && j < n_out) ? hd.bias[j] : 0.f; }
  // The 16 x H tile of hidden activations goes through LDS (wave w copies row w, coalesced; odd pitch): an A fragment
  // wants one value per lane from 16 DIFFERENT rows, which straight from global memory is 16 cache lines per load.
  float* hid = outs + 16 * n_out_p;   // [16][H + 1]
  const int hp = H + 1;
  for (int k = lane; k < H; k += WAVE) hid[wave * hp + k] = have ? hd.hidden[(size_t)srow * hd.hidden_stride + k] : 0.f;
  __syncthreads();
  const int ak = lane >> 4;          // k offset inside a k-step of 4 (A fragment: row lane % 16; B fragment: column lane % 16)
  const float* ap = hid + (lane & 15) * hp + ak;
  for (int t = wave; t < n_tiles; t += 16) {
    cg_floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* bp = hd.weight_t + (size_t)ak * n_out_p + t * 16 + (lane & 15);
#pragma unroll 16
    for (int k0 = 0; k0 < H; k0 += 4) {
      const float a = ap[k0];
      const float b = bp[(size_t)k0 * n_out_p];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    // D fragment: lane holds rows 4 * (lane / 16) + v, column lane % 16
#pragma unroll
    for (int v = 0; v < 4; ++v) outs[(4 * (lane >> 4) + v) * n_out_p + t * 16 + (lane & 15)] = acc[v];
  }
  __syncthreads();
  if (row < 0) return;
  head_decode_row<HEAD_OPL>(outs + wave * n_out_p, bias_r, hd.tanh_out, row, tick, tmap, src, dst, lane, seed, env_id_base);
}

// Synthetic action script of bench.py: alternating defender / attacker turns.
// Mirrored in numpy by cygym_amd/actions.py (tests check equality).
__global__ void gen_actions_kernel(KParams P, int tick, int32_t* mode, int32_t* n_groups, int32_t* atype,
                                   int32_t* n_exploit, int32_t* exploit, int32_t* app, int32_t* dev_cnt,
                                   int16_t* dev_idx, int max_devs) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= P.n_envs) return;
  const int M = P.t.M;
  const uint32_t env_id = (uint32_t)(P.c.env_id_base + env);
  cg_u32x4 r = cg_philox4x32_10(env_id, (uint32_t)tick, CG_SITE_ACTGEN, 0u, (uint32_t)P.c.seed, (uint32_t)(P.c.seed >> 32));
  const int m = tick & 1;
  mode[env] = m;
  n_groups[env] = 0;
  n_exploit[env] = 1;
  for (int j = 0; j < CG_MAX_EXPLOITS; ++j) exploit[(size_t)env * CG_MAX_EXPLOITS + j] = -1;
  exploit[(size_t)env * CG_MAX_EXPLOITS] = (int)cg_index(r.v[1], (uint32_t)(P.t.X > 0 ? P.t.X : 1));
  app[env] = (int)cg_index(r.v[2], 4u);
  int k = 0;
  if (m == CG_MODE_DEFENDER) {
    const int types[11] = {1, 4, 5, 6, 7, 8, 9, 11, 12, 13, 2};
    atype[env] = types[cg_index(r.v[0], 11u)];
    int kmax = M / 8 > 1 ? M / 8 : 1;
    if (kmax > max_devs) kmax = max_devs;
    k = 1 + (int)cg_index(r.v[3], (uint32_t)kmax);
    cg_u32x4 q = cg_philox4x32_10(env_id, (uint32_t)tick, CG_SITE_ACTGEN, 1u, (uint32_t)P.c.seed, (uint32_t)(P.c.seed >> 32));
    int a = (int)cg_index(q.v[0], (uint32_t)M);
    int stride = ((M & (M - 1)) == 0 && M > 1) ? (int)(2u * cg_index(q.v[1], (uint32_t)(M / 2)) + 1u) : 1;
    for (int j = 0; j < k; ++j) dev_idx[(size_t)env * max_devs + j] = (int16_t)((a + (long long)j * stride) % M);
  } else {
    atype[env] = 1 + (int)cg_index(r.v[0], 3u);
  }
  dev_cnt[env] = k;
}

#endif  // CG_AUX_KERNELS_HPP
