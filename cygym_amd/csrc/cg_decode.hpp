// cg_decode.hpp -- device code shared by the consumer-side kernels of the C-ABI unit (cg_aux_kernels.hpp) and the tick + actor
// unit (cg_inst_actor.hip): the decode of one action-vector row by one wave, and the whole-actor kernel body.  Templates and
// force-inlined device functions only (included in more than one translation unit, inside namespace cygym_k).
#ifndef CG_DECODE_HPP
#define CG_DECODE_HPP
typedef float cg_floatx4 __attribute__((ext_vector_type(4)));
constexpr int HEAD_WAVES = 16, HEAD_OPL_MAX = 8, HEAD_KC = 64;
// max over the wave of a (hi, lo) pair compared lexicographically; every lane active.  Result valid in lane 63.
__device__ __forceinline__ void dpp_pair_max(uint32_t& hi, uint32_t& lo) {
#define CG_PMAX(ctrl, rmask)                                                                          \
  {                                                                                                   \
    const uint32_t oh = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, (ctrl), (rmask), 0xf, false); \
    const uint32_t ol = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, (ctrl), (rmask), 0xf, false); \
    const bool take = oh > hi || (oh == hi && ol > lo);                                               \
    hi = take ? oh : hi; lo = take ? ol : lo;                                                         \
  }
  CG_PMAX(0x111, 0xf) CG_PMAX(0x112, 0xf) CG_PMAX(0x114, 0xf) CG_PMAX(0x118, 0xf) CG_PMAX(0x142, 0xa) CG_PMAX(0x143, 0xc)
#undef CG_PMAX
}
__device__ __forceinline__ uint32_t float_order_bits(float x) {   // a < b  <=>  bits(a) < bits(b) (finite values)
  const uint32_t u = __float_as_uint(x + 0.0f);   // (-0.0 + 0.0 = +0.0: the two zeros tie, like in np.argmax, and the first index wins)
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// Decode of ONE row by one wave (do_agent.py:970-998), shared by the matrix-core kernels: the row's action vector comes out
// of LDS (`outs_row`, n_out_p floats) plus the bias held in registers; everything the decode needs from global memory (row id,
// rng tick, type-map entry per lane) was requested by the caller ahead of the product.
template <int HEAD_OPL>
__device__ __forceinline__ void head_decode_row(const float* outs_row, const float (&bias_r)[HEAD_OPL], const int tanh_out, const int row,
                                                const uint32_t tick, const int tmap, const cygym_action_vectors& src,
                                                const cygym_actions& dst, const int lane, const uint64_t seed, const int64_t env_id_base) {
  const int G = dst.max_groups, L = dst.max_devs, M = src.n_devices, nt = src.n_types;
  const int n_out = nt + M + src.n_exploits + src.n_apps;
  float v[HEAD_OPL];
#pragma unroll
  for (int i = 0; i < HEAD_OPL; ++i) {
    const float x = outs_row[lane + i * WAVE] + bias_r[i];
    v[i] = tanh_out ? tanhf(x) : x;
  }
  auto range_argmax = [&](int lo, int hi) -> int {
    uint32_t bh = 0u, bl = 0u;
#pragma unroll
    for (int i = 0; i < HEAD_OPL; ++i) {
      if ((i + 1) * WAVE <= lo || i * WAVE >= hi) continue;   // (scalar branch: a register none of whose lanes is in range)
      const int j = lane + i * WAVE;
      const uint32_t ob = float_order_bits(v[i]);
      if (j >= lo && j < hi && ob > bh) { bh = ob; bl = ~(uint32_t)(j - lo); }
    }
    dpp_pair_max(bh, bl);
    const uint32_t rl = (uint32_t)__builtin_amdgcn_readlane((int)bl, 63), rh = (uint32_t)__builtin_amdgcn_readlane((int)bh, 63);
    return rh == 0u ? 0 : (int)~rl;
  };
  int at = nt > 0 ? range_argmax(0, nt) : 0;
  if (src.epsilon_thr && nt > 0) {   // epsilon-greedy (do_agent.py:972-973)
    const cg_u32x4 rr = cg_philox4x32_10((uint32_t)(env_id_base + row), tick, CG_SITE_EPS_TYPE, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
    if ((uint64_t)rr.v[0] < src.epsilon_thr) at = (int)cg_index(rr.v[1], (uint32_t)nt);
  }
  if (nt > 0) at = nt <= WAVE ? __shfl(tmap, at) : (src.type_map ? src.type_map[at] : at);
  int16_t* out = const_cast<int16_t*>(dst.dev_idx) + (size_t)row * L;
  int base = 0;
#pragma unroll
  for (int i = 0; i < HEAD_OPL; ++i) {
    if ((i + 1) * WAVE <= nt || i * WAVE >= nt + M) continue;   // (scalar branch: no device value in this register)
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

#include "cg_actor_mlp.hpp"
#endif  // CG_DECODE_HPP
