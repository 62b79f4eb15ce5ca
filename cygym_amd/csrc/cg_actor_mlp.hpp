// cg_actor_mlp.hpp -- cygym_actor_mlp_decode: the actor network of a closed-loop strategy (Linear-ReLU stack + last Linear layer,
// do_agent.py:357-370) and decode_action (do_agent.py:970-998) in ONE launch.  Included through cg_decode.hpp (C-ABI unit and
// the tick + actor unit).
//
// A workgroup of 16 waves owns 16 observation rows (= 16 envs).
//   * Layer 0, [16 x K] x [K x N0]: the observation tile is requested from HBM in one go (up to 1536 columns = 96 KB per
//     workgroup, 6 x 16 bytes per thread in flight: one workgroup per CU has to keep ~60 KB in flight to draw its share of the
//     HBM bandwidth), lands in LDS in stages of 512 columns, and each stage is multiplied as soon as it has landed
//     (v_mfma_f32_16x16x4_f32).  Output tile t (16 columns) and k-slice q go to wave q * n_tiles + t; the k-slices' partial
//     sums meet in LDS, where bias + ReLU are applied.
//   * LDS rows are XOR-swizzled at 16-byte granularity (slot ^ row): an A fragment is ONE ds_read_b128 per four
//     matrix instructions -- lane (row r = lane % 16, kk = lane / 16) reads A[r][16 g + 4 kk .. + 3] -- and the four
//     16-lane groups of that instruction hit 16 different slots (MI355X_MICROARCH.md, LDS table).
//   * Weights come PACKED in fragment order (cygym_abi.h): lane reads B[16 g + 4 kk + i][16 t + lane % 16], i = 0..3, as one
//     global_load_dwordx4, 1 KB contiguous per wave.  Every weight is read once per workgroup (from L2).
//   * Further hidden layers and the last layer read their A fragments from the hidden tile in LDS the same way; the
//     16 x n_out_p action vectors pass through LDS once to get each row into one wave, which decodes it (head_decode_row).
constexpr int MLP_NB = 8, MLP_NB_SMALL = 4, MLP_NB_ROLE = 8;   // k-groups per batch of weight requests: layer 0 / the later layers
constexpr int MLP_KT = 1536, MLP_STAGE = 512, MLP_STAGES = MLP_KT / MLP_STAGE, MLP_THREADS = 16 * WAVE;

// LDS plan (floats), the same arithmetic on both sides of the launch
struct MlpPlan {
  int kt;        // columns of the observation tile held at once: min(K rounded up to whole stages, MLP_KT)
  int region_a;  // observation tile, later the [16][n_out_p] action vectors
  int hp;        // pitch of a hidden tile: the widest hidden layer rounded up to 64
  int total;     // floats
};
__host__ __device__ inline MlpPlan mlp_plan(int K, int n_hidden, const int32_t* width, int n_out_p) {
  MlpPlan p;
  const int ks = (K + MLP_STAGE - 1) / MLP_STAGE * MLP_STAGE;   // (whole stages: the stage stores need no column guard)
  p.kt = ks < MLP_KT ? ks : MLP_KT;
  p.region_a = 16 * (p.kt > n_out_p ? p.kt : n_out_p);
  int wmax = 0;
#pragma unroll
  for (int l = 0; l < CG_MLP_MAX_HIDDEN; ++l) wmax = (l < n_hidden && width[l] > wmax) ? width[l] : wmax;   // (constant indices: the kernel reads kernarg words)
  p.hp = (wmax + 63) & ~63;
  p.total = p.region_a + 16 * 256 + 2 * 16 * p.hp;
  return p;
}

// obs_role != 0: the role view is not read from HBM but built in LDS from the batch's state -- the flag plane of the 16 envs
// (256 bytes per env at M = 256 instead of a 6 KB view) and the topology's static columns, exactly as the tick kernel's
// write_obs_def / write_obs_att build it (CyberDefenseEnv.py:194-257).
struct MlpView {
  const uint8_t* live;     // [N][4][M], plane 0 = flags
  const float *osv, *ver, *ano;   // [M] static columns
  const float* ano_dyn;    // [N][M] per-env anomaly plane (fast_scan = False) or nullptr
  int M, X, max_exploits, role;
};

// acc += A[16 x (16 per group)] x B over the k-groups g0, g0 + gstep, ... < g1, in batches of NB groups: the B fragments of a
// batch are requested together (one 16-byte load each, no branches: a group past the end re-reads the last valid one and
// is skipped by a scalar branch), then consumed.  a_row = this lane's row of the swizzled LDS tile, r = lane % 16,
// kk = lane / 16; wp = packed weights of this output tile, already offset by the lane, indexed by GLOBAL group
// (gofs = global index of the tile's local group 0; gmax = last global group of the tile).  g0, g1, gstep are wave-uniform.
template <int NB>
__device__ __forceinline__ void mlp_b_load(float4 (&b)[NB], const float4* wp, const int gofs, const int gmax, const int g0, const int gstep) {
#pragma unroll
  for (int u = 0; u < NB; ++u) {
    const int g = gofs + g0 + u * gstep;
    b[u] = wp[(size_t)(g < gmax ? g : gmax) * WAVE];
  }
}
// Two accumulator chains per wave (the matrix instructions of a k-group alternate between them; summed at the end).  Measured:
// no faster than one chain -- a dependent v_mfma_f32_16x16x4_f32 already issues back to back here -- and four chains spill; two
// cost four registers and keep the sums of even and odd k apart, which is what the exactness tests pin.
struct MlpAcc {
  cg_floatx4 c[2];
  __device__ __forceinline__ void zero() { c[0] = cg_floatx4{0.f, 0.f, 0.f, 0.f}; c[1] = c[0]; }
  __device__ __forceinline__ cg_floatx4 sum() const { return c[0] + c[1]; }
};
template <int NB>
__device__ __forceinline__ void mlp_mfma_batch(MlpAcc& acc, const float* a_row, const int r, const int kk, const float4 (&b)[NB],
                                               const int g0, const int g1, const int gstep) {
  // branch-free: the A fragments of four k-groups are requested from LDS together (a group past the end re-reads group
  // g1 - 1 and contributes zeros), then their sixteen matrix instructions run back to back
  constexpr int CH = NB < 4 ? NB : 4;
  const int gl = g1 > 0 ? g1 - 1 : 0;
#pragma unroll
  for (int c = 0; c < NB; c += CH) {
    float4 a[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int g = g0 + (c + u) * gstep, gc = g < gl ? g : gl;
      a[u] = *reinterpret_cast<const float4*>(a_row + ((((gc << 2) + kk) ^ r) << 2));
    }
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const bool ok = g0 + (c + u) * gstep < g1;   // (wave-uniform)
      const float ax = ok ? a[u].x : 0.f, ay = ok ? a[u].y : 0.f, az = ok ? a[u].z : 0.f, aw = ok ? a[u].w : 0.f;
      acc.c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ax, b[c + u].x, acc.c[0], 0, 0, 0);
      acc.c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ay, b[c + u].y, acc.c[1], 0, 0, 0);
      acc.c[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(az, b[c + u].z, acc.c[0], 0, 0, 0);
      acc.c[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw, b[c + u].w, acc.c[1], 0, 0, 0);
    }
  }
}
// The same over any number of groups, software-pipelined in half batches: while one half batch is multiplied the weights of
// the next one are in flight (requests past the end re-read the tile's last group and are never used).
template <int NB>
__device__ __forceinline__ void mlp_mfma_groups(MlpAcc& acc, const float* a_row, const int r, const int kk, const float4* wp,
                                                const int gofs, const int gmax, const int g0, const int g1, const int gstep) {
  constexpr int NH = NB / 2;
  if (g0 >= g1) return;
  float4 b0[NH], b1[NH];
  mlp_b_load<NH>(b0, wp, gofs, gmax, g0, gstep);
  for (int g = g0; g < g1; g += NB * gstep) {
    mlp_b_load<NH>(b1, wp, gofs, gmax, g + NH * gstep, gstep);
    mlp_mfma_batch<NH>(acc, a_row, r, kk, b0, g, g1, gstep);
    mlp_b_load<NH>(b0, wp, gofs, gmax, g + NB * gstep, gstep);
    mlp_mfma_batch<NH>(acc, a_row, r, kk, b1, g + NH * gstep, g1, gstep);
  }
}

// k-slice partial sums of a hidden layer (part[wave][16][16]) -> bias + ReLU -> swizzled hidden tile `out` (pitch hp).
// A thread owns outputs o = tid, tid + 1024, ... (< 16 N <= 4096); their biases are requested BEFORE the barrier that
// publishes the partial sums (mlp_bias_prefetch), so the round trip is not part of the chain.
__device__ __forceinline__ void mlp_bias_prefetch(float (&bv)[4], const float* bias, const int N) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = threadIdx.x + MLP_THREADS * i;
    bv[i] = (bias && o < 16 * N) ? bias[o % N] : 0.f;
  }
}
__device__ __forceinline__ void mlp_finish_hidden(const float* part, const float (&bv)[4], const int N, const int n_tiles, const int ksplit,
                                                  float* out, const int hp) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int o = threadIdx.x + MLP_THREADS * i;
    if (o < 16 * N) {
      const int row = o / N, c = o - row * N, t = c >> 4;
      float s = bv[i];
      for (int q = 0; q < ksplit; ++q) s += part[((q * n_tiles + t) << 8) + (row << 4) + (c & 15)];
      out[row * hp + ((((c >> 2) ^ row) << 2) | (c & 3))] = s > 0.f ? s : 0.f;
    }
  }
}

// FROM_LDS (tick_actor_kernel, cg_tick_actor.hpp): the flag planes of the 16 envs are still in LDS where the tick left them --
// row r's at lds_flags + r * lds_pitch -- and source row == env id; they are read before the observation tile overwrites them.
template <int HEAD_OPL, int VW, bool FROM_LDS>
__device__ __forceinline__ void actor_mlp_body(cygym_actor_mlp ml, const cygym_action_vectors& src, const cygym_actions& dst, const int n_envs,
                                               const int32_t* ienv, const uint64_t seed, const int64_t env_id_base, unsigned long long* st,
                                               const MlpView& vw, const uint8_t* lds_flags, const int lds_pitch) {
#ifdef CG_STAMPS
#define MSTAMP(k) do { if (st && threadIdx.x == 0) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); st[(size_t)blockIdx.x * 16 + (k)] = _t; } } while (0)
#else
#define MSTAMP(k) do { } while (0)
#endif
  MSTAMP(0);
  extern __shared__ __align__(16) uint8_t smem[];
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;   // (wave: an SGPR, so are the tile / k-slice indices)
  const int r = lane & 15, kk = lane >> 4;
  // HEAD_OPL = 0: action vectors wider than 512 (more than ~490 devices) are produced and decoded in chunks of 512 outputs
  // (STREAM); otherwise the whole vector is HEAD_OPL * 64 wide and decoded from registers in one go.
  constexpr bool STREAM = HEAD_OPL == 0;
  constexpr int OPLc = STREAM ? 8 : HEAD_OPL;
  constexpr int n_out_p = OPLc * WAVE;            // width of the LDS action-vector tile (= the whole padded vector unless STREAM)
  const int K = ml.K, nt = src.n_types;
  const int n_out = nt + src.n_devices + src.n_exploits + src.n_apps;
  const int n_out_pr = STREAM ? ((n_out + 63) & ~63) : n_out_p;   // the whole padded vector: what the packed last layer holds
  const MlpPlan pl = mlp_plan(K, ml.n_hidden, ml.width, n_out_p);
  const int kt = pl.kt, hp = pl.hp;
  float* At = reinterpret_cast<float*>(smem);
  float* outs = At;                       // (the observation tile is dead by the time the last layer writes)
  float* part = At + pl.region_a;         // [16 waves][16][16]
  float* hin = part + 16 * 256;            // hidden tiles [16][hp], swizzled like the observation tile
  float* hout = hin + 16 * hp;
  const int row0 = blockIdx.x * 16;
  const int G0 = (K + 15) >> 4;           // k-groups of layer 0
  // per-layer pointers (constant indices only: a dynamically indexed kernarg struct would be copied to scratch)
  const float* wl[CG_MLP_MAX_HIDDEN];
  const float* bl[CG_MLP_MAX_HIDDEN];
  const float* w_head = ml.w_head;
  const float* b_head = ml.b_head;
  {
    const size_t grp = ml.n_groups > 1 ? (size_t)((row0 / ml.rows_per_group) % ml.n_groups) : 0;   // a population of actors: this workgroup's 16 rows belong to ONE of them
    int kin = G0;
#pragma unroll
    for (int l = 0; l < CG_MLP_MAX_HIDDEN; ++l) {
      // (no nullptr constants in these selects: the compiler then keeps the pointers in the global address space -- global_load,
      //  not flat_load, whose completion cannot be counted in order)
      const int wd = l < ml.n_hidden ? ml.width[l] : 0;
      wl[l] = ml.w[l] + grp * (size_t)(wd >> 4) * kin * 256;
      bl[l] = ml.b[l] + (ml.b[l] ? grp * wd : 0);
      kin = l < ml.n_hidden ? wd >> 4 : kin;
    }
    w_head += grp * (size_t)(n_out_pr >> 4) * kin * 256;
    b_head += b_head ? grp * n_out : 0;
  }
  // What the decode of this wave's row will need from global memory (row id, rng tick, type-map entry per lane) is requested
  // after layer 0 -- late enough not to delay the observation requests (the memory counters retire in order), early enough
  // to have landed by the decode.
  int row = -1, tmap = lane;
  uint32_t tick = 0;
  int w_last = ml.width[0];   // width of the last hidden layer
#pragma unroll
  for (int l = 1; l < CG_MLP_MAX_HIDDEN; ++l) w_last = l < ml.n_hidden ? ml.width[l] : w_last;
  const int Gh = w_last >> 4;   // k-groups of the last layer
  const int n_tiles_tot = n_out_pr >> 4;                                   // output tiles of the last layer
  const int n_tiles_out = n_tiles_tot < 32 ? n_tiles_tot : 32;             // ... of its first (or only) chunk: at most two per wave
  const float4* whp = reinterpret_cast<const float4*>(w_head) + lane;
  float4 hb0[MLP_NB_SMALL], hb1[MLP_NB_SMALL];   // first weight batches of the last layer's tiles `wave` and `wave + 16`
  // ---------------- layer 0: observation tile from HBM through LDS ----------------
  {
    const int N = ml.width[0], n_tiles = N >> 4, ksplit = 16 / n_tiles;
    const int t = wave % n_tiles, q = wave / n_tiles;
    const bool active = q < ksplit;
    const float4* wp = reinterpret_cast<const float4*>(wl[0]) + (size_t)t * G0 * WAVE + lane;
    const float* a_row = At + r * kt;
    MlpAcc acc;
    acc.zero();
    // Each thread copies MLP_STAGE * 16 / 1024 / VW vectors of VW floats per stage: vector i = tid + 1024 * hh of the stage is
    // row i / (MLP_STAGE / VW) of the tile, vector i % (MLP_STAGE / VW) of the stage's columns (consecutive lanes, consecutive
    // addresses).  VW = 4 / 2 / 1 by what the rows' alignment allows: base address and row stride (a dense attacker view of
    // 4 M + 6 floats: 2).
    // VW = 0: the role view built on chip (obs_role) -- its own instantiation, so that neither path carries the other's registers.
    if constexpr (VW != 0) {
      constexpr int VWc = VW > 0 ? VW : 1;
      constexpr int VPS = MLP_STAGE / VWc, NV = VPS * 16 / MLP_THREADS;   // vectors per row and stage; vectors per thread and stage
      typedef float vec_t __attribute__((ext_vector_type(VWc)));
      // (branch-free requests: a row or column outside the source is read from a valid address and replaced by zeros)
      uint32_t rbase[NV];   // element offset of the row (32 bits: the host refuses views of 2^32 floats or more)
      bool rok[NV];
  #pragma unroll
      for (int hh = 0; hh < NV; ++hh) {
        const int sr = row0 + (tid + MLP_THREADS * hh) / VPS;
        bool ok = sr < src.n;
        long orow = ok ? sr : 0;
        if (ml.obs_by_env) {
          orow = (ok && src.rows) ? src.rows[sr] : orow;
          ok = ok && orow >= 0 && orow < n_envs;
          orow = ok ? orow : 0;
        }
        rbase[hh] = (uint32_t)((size_t)orow * ml.obs_stride);
        rok[hh] = ok;
      }
      for (int kc0 = 0; kc0 < K; kc0 += MLP_KT) {
        const int rem64 = (K - kc0 + 63) & ~63;
        const int kcur = rem64 < kt ? rem64 : kt;   // columns of this tile that hold data (a multiple of 64)
        // VW == 1 (rows that are only 4-byte aligned: odd strides) holds eight scalars per thread and stage: with all three
        // stages requested up front it spilled 10-11 VGPRs.  It copies a stage when it consumes it, four scalars at a time.
        constexpr bool PRE = VW != 1;
        vec_t x[PRE ? MLP_STAGES : 1][PRE ? NV : 1];
        const int gofs = kc0 >> 4, gw = G0 - gofs, gk = kcur >> 4;
        float4 b0[MLP_NB];
        // Request order = arrival order (the memory counters retire in order): stage 0 of the tile, the weights of stage 0,
        // then the rest of the tile.
  #pragma unroll
        for (int s = 0; s < MLP_STAGES; ++s) {
          if constexpr (PRE) {
  #pragma unroll
          for (int hh = 0; hh < NV; ++hh) {
            const int col = kc0 + MLP_STAGE * s + VW * ((tid + MLP_THREADS * hh) % VPS);
            const bool in = col < K;   // (a vector that straddles K reads into the row's padding: obs_stride % VW == 0; zeroed below)
            x[s][hh] = *reinterpret_cast<const vec_t*>(ml.obs + (size_t)rbase[hh] + (in ? col : 0));   // (zeroed when stored: no wait here)
          }
          }
          if (s == 0) {
            __builtin_amdgcn_sched_barrier(0);
            mlp_b_load<MLP_NB>(b0, wp, gofs, G0 - 1, q, ksplit);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        MSTAMP(1);
  #pragma unroll
        for (int s = 0; s < MLP_STAGES; ++s) {
          if (MLP_STAGE * s < kcur) {   // (uniform)
            const int gl1 = (MLP_STAGE / 16) * (s + 1);
            int g1 = gl1 < gk ? gl1 : gk;
            g1 = active ? (g1 < gw ? g1 : gw) : 0;   // (no groups for an idle wave)
            const int g0 = (MLP_STAGE / 16) * s + q;
            float4 b[MLP_NB];
            if (s == 0) {
  #pragma unroll
              for (int u = 0; u < MLP_NB; ++u) b[u] = b0[u];
            } else
            mlp_b_load<MLP_NB>(b, wp, gofs, G0 - 1, g0, ksplit);   // the stage's first weights fly under the stores and the barrier
            if constexpr (PRE) {
  #pragma unroll
            for (int hh = 0; hh < NV; ++hh) {
              const int i = tid + MLP_THREADS * hh, trow = i / VPS, c = MLP_STAGE * s + VW * (i % VPS);
              vec_t v = x[s][hh];
  #pragma unroll
              for (int j = 0; j < VW; ++j) v[j] = (kc0 + c + j < K && rok[hh]) ? v[j] : 0.f;
              *reinterpret_cast<vec_t*>(At + trow * kt + ((((c >> 2) ^ trow) << 2) | (c & 3))) = v;
            }
            } else {
              constexpr int HB = NV >= 4 ? 4 : NV;   // scalars in flight per thread
  #pragma unroll
              for (int h0 = 0; h0 < NV; h0 += HB) {
                float y[HB];
  #pragma unroll
                for (int h = 0; h < HB; ++h) {
                  const int col = kc0 + MLP_STAGE * s + ((tid + MLP_THREADS * (h0 + h)) % VPS);
                  y[h] = ml.obs[(size_t)rbase[h0 + h] + (col < K ? col : 0)];
                }
  #pragma unroll
                for (int h = 0; h < HB; ++h) {
                  const int i = tid + MLP_THREADS * (h0 + h), trow = i / VPS, c = MLP_STAGE * s + (i % VPS);
                  At[trow * kt + ((((c >> 2) ^ trow) << 2) | (c & 3))] = (kc0 + c < K && rok[h0 + h]) ? y[h] : 0.f;
                }
              }
            }
            __syncthreads();
            MSTAMP(2 + 2 * s);
            mlp_mfma_batch<MLP_NB>(acc, a_row, r, kk, b, g0, g1, ksplit);
            mlp_mfma_groups<MLP_NB>(acc, a_row, r, kk, wp, gofs, G0 - 1, g0 + MLP_NB * ksplit, g1, ksplit);
            MSTAMP(3 + 2 * s);
          }
        }
        if (kc0 + MLP_KT < K) __syncthreads();   // the next tile overwrites this one
      }
    } else {
      // ---- the role view of the 16 envs, built in LDS from their flag planes ----
      MSTAMP(1);
      const int M = vw.M;
      for (int kc0 = 0; kc0 < K; kc0 += MLP_KT) {
        const int rem64 = (K - kc0 + 63) & ~63;
        const int kcur = rem64 < kt ? rem64 : kt;
        const int c4max = kt >> 2;
        if (vw.role == 1) {   // defender view: one lane per device PAIR, three 16-byte stores (M even)
          const int p0 = kc0 / 12;
          uint32_t f2[2];
          float2 o[2], v[2], an[2];
          bool okp[2];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {   // every request of both pairs first (branch-free: clamped addresses), then the arithmetic
            const int i = tid + MLP_THREADS * hh, trow = i >> 7, p = p0 + (i & 127), sr = row0 + trow;
            long env = sr < src.n ? (src.rows ? src.rows[sr] : sr) : -1;
            if (env >= n_envs) env = -1;
            okp[hh] = env >= 0 && 2 * p < M;
            const int pc = 2 * p < M ? p : 0;
            const size_t ec = env >= 0 ? (size_t)env : 0;
            if constexpr (FROM_LDS) f2[hh] = reinterpret_cast<const uint16_t*>(lds_flags + (size_t)trow * lds_pitch)[pc];
            else f2[hh] = reinterpret_cast<const uint16_t*>(vw.live + ec * 4 * M)[pc];
            o[hh] = reinterpret_cast<const float2*>(vw.osv)[pc];
            v[hh] = reinterpret_cast<const float2*>(vw.ver)[pc];
            an[hh] = vw.ano_dyn ? reinterpret_cast<const float2*>(vw.ano_dyn + ec * M)[pc] : reinterpret_cast<const float2*>(vw.ano)[pc];
          }
          if constexpr (FROM_LDS) __syncthreads();   // (every flag byte is in a register before the tile overwrites the planes)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int i = tid + MLP_THREADS * hh, trow = i >> 7, pl = i & 127;
            const uint32_t fa = f2[hh] & 0xFFu, fb = f2[hh] >> 8;
            const bool ha = (fa & CG_F_NYA) || !(fa & CG_F_OWNED), hb = (fb & CG_F_NYA) || !(fb & CG_F_OWNED);
            const float ka = (float)((fa >> 2) & 1u), kb = (float)((fb >> 2) & 1u);
            float4 o0 = ha ? make_float4(-1.f, -1.f, -1.f, -1.f) : make_float4(o[hh].x, v[hh].x, -1.f, an[hh].x);
            float4 o1 = make_float4(ha ? -1.f : ka, ha ? -1.f : 0.f, hb ? -1.f : o[hh].y, hb ? -1.f : v[hh].y);
            float4 o2 = hb ? make_float4(-1.f, -1.f, -1.f, -1.f) : make_float4(-1.f, an[hh].y, kb, 0.f);
            if (!okp[hh]) { o0 = make_float4(0.f, 0.f, 0.f, 0.f); o1 = o0; o2 = o0; }
            float* arow = At + trow * kt;
            const int c4 = 3 * pl;
            if (c4 + 0 < c4max) *reinterpret_cast<float4*>(arow + (((c4 + 0) ^ trow) << 2)) = o0;
            if (c4 + 1 < c4max) *reinterpret_cast<float4*>(arow + (((c4 + 1) ^ trow) << 2)) = o1;
            if (c4 + 2 < c4max) *reinterpret_cast<float4*>(arow + (((c4 + 2) ^ trow) << 2)) = o2;
          }
        } else {   // attacker view: one 16-byte store per device, then the exploit availability bits
          const int d0 = kc0 >> 2;
          uint32_t fl[6];
          float ov[6], vv[6];
          int st_[6];   // 0: zeros (no env), 1: device, 2: availability bits
#pragma unroll
          for (int hh = 0; hh < 6; ++hh) {   // requests first (clamped addresses), arithmetic after
            const int i = tid + MLP_THREADS * hh, trow = i / (MLP_KT / 4), dl = i - trow * (MLP_KT / 4), d = d0 + dl, sr = row0 + trow;
            long env = sr < src.n ? (src.rows ? src.rows[sr] : sr) : -1;
            if (env >= n_envs) env = -1;
            st_[hh] = env < 0 ? 0 : d < M ? 1 : 2;
            const int dc = d < M ? d : 0;
            if constexpr (FROM_LDS) fl[hh] = (lds_flags + (size_t)trow * lds_pitch)[dc];
            else fl[hh] = vw.live[(env >= 0 ? (size_t)env : 0) * 4 * M + dc];
            ov[hh] = vw.osv[dc];
            vv[hh] = vw.ver[dc];
          }
          if constexpr (FROM_LDS) __syncthreads();
#pragma unroll
          for (int hh = 0; hh < 6; ++hh) {
            const int i = tid + MLP_THREADS * hh, trow = i / (MLP_KT / 4), dl = i - trow * (MLP_KT / 4), d = d0 + dl;
            const uint32_t f = fl[hh];
            const bool vis = (f & CG_F_KNOWN) && !(f & CG_F_NYA) && (f & CG_F_OWNED);
            float4 o = vis ? make_float4(ov[hh], vv[hh], (float)(f & 1u), 1.f) : make_float4(-1.f, -1.f, -1.f, -1.f);
            if (st_[hh] == 2) {   // MaxExploits availability bits (CyberDefenseEnv.py:226-233)
              const int e0 = 4 * (d - M);
              o.x = (e0 + 0 < vw.max_exploits && e0 + 0 < vw.X) ? 1.f : 0.f; o.y = (e0 + 1 < vw.max_exploits && e0 + 1 < vw.X) ? 1.f : 0.f;
              o.z = (e0 + 2 < vw.max_exploits && e0 + 2 < vw.X) ? 1.f : 0.f; o.w = (e0 + 3 < vw.max_exploits && e0 + 3 < vw.X) ? 1.f : 0.f;
            }
            if (st_[hh] == 0) o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (dl < c4max) *reinterpret_cast<float4*>(At + trow * kt + ((dl ^ trow) << 2)) = o;
          }
        }
        __syncthreads();
        MSTAMP(2);
        const int gofs = kc0 >> 4, gw = G0 - gofs, gk = kcur >> 4;
        const int g1 = active ? (gk < gw ? gk : gw) : 0;
        mlp_mfma_groups<MLP_NB_ROLE>(acc, a_row, r, kk, wp, gofs, G0 - 1, q, g1, ksplit);
        MSTAMP(7);
#ifdef CG_STAMPS
        if (st && lane == 0 && (wave & 3) == 3) { unsigned long long _t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t) :: "memory"); st[(size_t)blockIdx.x * 16 + 12 + (wave >> 2)] = _t; }
#endif
        if (kc0 + MLP_KT < K) __syncthreads();   // the next tile overwrites this one
      }
    }
    {
      const int srow = row0 + wave;
      row = srow < src.n ? (src.rows ? src.rows[srow] : srow) : -1;
      if (row >= n_envs) row = -1;
      if (src.epsilon_thr && row >= 0) tick = (uint32_t)ienv[(size_t)row * CG_I_COUNT + CG_I_RNG_TICK];
      if (src.type_map && lane < nt) tmap = src.type_map[lane];
      const int t0 = wave < n_tiles_out ? wave : 0, t1 = wave + 16 < n_tiles_out ? wave + 16 : 0;   // (a tile this wave does not have: any valid address)
      mlp_b_load<MLP_NB_SMALL>(hb0, whp + (size_t)t0 * Gh * WAVE, 0, Gh - 1, 0, 1);
      mlp_b_load<MLP_NB_SMALL>(hb1, whp + (size_t)t1 * Gh * WAVE, 0, Gh - 1, 0, 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (active) {
      const cg_floatx4 accs = acc.sum();
#pragma unroll
      for (int v = 0; v < 4; ++v) part[(wave << 8) + ((4 * kk + v) << 4) + r] = accs[v];   // D fragment: rows 4 * (lane / 16) + v, column lane % 16
    }
    float bv[4];
    mlp_bias_prefetch(bv, bl[0], N);
    __syncthreads();
    MSTAMP(8);
    mlp_finish_hidden(part, bv, N, n_tiles, ksplit, hin, hp);
    __syncthreads();
    MSTAMP(9);
  }
  // ---------------- further hidden layers: A fragments from the hidden tile ----------------
#pragma unroll
  for (int l = 1; l < CG_MLP_MAX_HIDDEN; ++l) {
    if (l < ml.n_hidden) {   // (uniform)
      const int Gin = ml.width[l - 1] >> 4, N = ml.width[l], n_tiles = N >> 4, ksplit = 16 / n_tiles;
      const int t = wave % n_tiles, q = wave / n_tiles;
      if (q < ksplit) {
        MlpAcc acc4;
        acc4.zero();
        mlp_mfma_groups<MLP_NB_SMALL>(acc4, hin + r * hp, r, kk, reinterpret_cast<const float4*>(wl[l]) + (size_t)t * Gin * WAVE + lane, 0, Gin - 1, q, Gin, ksplit);
        const cg_floatx4 acc = acc4.sum();
#pragma unroll
        for (int v = 0; v < 4; ++v) part[(wave << 8) + ((4 * kk + v) << 4) + r] = acc[v];
      }
      float bv[4];
      mlp_bias_prefetch(bv, bl[l], N);
      __syncthreads();
      mlp_finish_hidden(part, bv, N, n_tiles, ksplit, hout, hp);
      __syncthreads();
      float* sw = hin; hin = hout; hout = sw;
    }
  }
  // ---------------- last layer -> outs [16][n_out_p] ----------------
  if constexpr (!STREAM) {
    float bias_r[OPLc];   // (requested here: the loads fly under the last product)
  #pragma unroll
    for (int i = 0; i < OPLc; ++i) { const int j = lane + i * WAVE; bias_r[i] = (b_head && j < n_out) ? b_head[j] : 0.f; }
    {
      const float* a_row = hin + r * hp;
      if (wave < n_tiles_out) {
        MlpAcc acc4;
        acc4.zero();
        mlp_mfma_batch<MLP_NB_SMALL>(acc4, a_row, r, kk, hb0, 0, Gh, 1);
        mlp_mfma_groups<MLP_NB_SMALL>(acc4, a_row, r, kk, whp + (size_t)wave * Gh * WAVE, 0, Gh - 1, MLP_NB_SMALL, Gh, 1);
        const cg_floatx4 acc = acc4.sum();
  #pragma unroll
        for (int v = 0; v < 4; ++v) outs[(4 * kk + v) * n_out_p + wave * 16 + r] = acc[v];
      }
      if (wave + 16 < n_tiles_out) {
        MlpAcc acc4;
        acc4.zero();
        mlp_mfma_batch<MLP_NB_SMALL>(acc4, a_row, r, kk, hb1, 0, Gh, 1);
        mlp_mfma_groups<MLP_NB_SMALL>(acc4, a_row, r, kk, whp + (size_t)(wave + 16) * Gh * WAVE, 0, Gh - 1, MLP_NB_SMALL, Gh, 1);
        const cg_floatx4 acc = acc4.sum();
  #pragma unroll
        for (int v = 0; v < 4; ++v) outs[(4 * kk + v) * n_out_p + (wave + 16) * 16 + r] = acc[v];
      }
    }
    __syncthreads();
    MSTAMP(10);
    if (row < 0) return;
    head_decode_row<OPLc>(outs + wave * n_out_p, bias_r, ml.tanh_out, row, tick, tmap, src, dst, lane, seed, env_id_base);
    MSTAMP(11);
  } else {
    // ---- STREAM: chunks of 512 outputs through the same LDS tile; a wave keeps its row's running arg-maxima and list length ----
    const int M = src.n_devices, G = dst.max_groups, L = dst.max_devs;
    const int lo1 = nt + M, lo2 = nt + M + src.n_exploits;
    uint32_t bh0 = 0u, bl0 = 0u, bh1 = 0u, bl1 = 0u, bh2 = 0u, bl2 = 0u;   // (order bits, ~index) of types / exploit / app so far, per lane
    int base = 0;
    int16_t* out = const_cast<int16_t*>(dst.dev_idx) + (size_t)(row >= 0 ? row : 0) * L;
    const float* a_row = hin + r * hp;
    const int n_chunks = (n_tiles_tot + 31) >> 5;
    for (int c = 0; c < n_chunks; ++c) {
      const int tb = c << 5, te = tb + 32 < n_tiles_tot ? tb + 32 : n_tiles_tot;
      float bias_r[OPLc];
#pragma unroll
      for (int i = 0; i < OPLc; ++i) { const int j = (c << 9) + lane + i * WAVE; bias_r[i] = (b_head && j < n_out) ? b_head[j] : 0.f; }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int t = tb + wave + 16 * half;
        if (t < te) {   // (uniform)
          MlpAcc acc4;
          acc4.zero();
          if (c == 0) {
            mlp_mfma_batch<MLP_NB_SMALL>(acc4, a_row, r, kk, half ? hb1 : hb0, 0, Gh, 1);
            mlp_mfma_groups<MLP_NB_SMALL>(acc4, a_row, r, kk, whp + (size_t)t * Gh * WAVE, 0, Gh - 1, MLP_NB_SMALL, Gh, 1);
          } else {
            mlp_mfma_groups<MLP_NB_SMALL>(acc4, a_row, r, kk, whp + (size_t)t * Gh * WAVE, 0, Gh - 1, 0, Gh, 1);
          }
          const cg_floatx4 acc = acc4.sum();
#pragma unroll
          for (int v = 0; v < 4; ++v) outs[(4 * kk + v) * n_out_p + (t - tb) * 16 + r] = acc[v];
        }
      }
      __syncthreads();
      if (row >= 0) {
#pragma unroll
        for (int i = 0; i < OPLc; ++i) {
          const int j = (c << 9) + lane + i * WAVE;
          if ((c << 9) + i * WAVE >= n_out) break;   // (uniform: past the vector)
          const float x = outs[wave * n_out_p + lane + i * WAVE] + bias_r[i];
          const float v = ml.tanh_out ? tanhf(x) : x;
          const uint32_t ob = float_order_bits(v);
          if (j < nt && ob > bh0) { bh0 = ob; bl0 = ~(uint32_t)j; }
          if (j >= lo1 && j < lo2 && ob > bh1) { bh1 = ob; bl1 = ~(uint32_t)(j - lo1); }
          if (j >= lo2 && j < n_out && ob > bh2) { bh2 = ob; bl2 = ~(uint32_t)(j - lo2); }
          const int d = j - nt;
          const bool on = d >= 0 && d < M && v > 0.f;
          const uint64_t m = __ballot(on);
          const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (on && pos < L) out[pos] = (int16_t)d;
          base += __popcll(m);
        }
      }
      if (c + 1 < n_chunks) __syncthreads();   // the next chunk overwrites the tile
    }
    MSTAMP(10);
    if (row < 0) return;
    auto wave_best = [&](uint32_t h, uint32_t l) -> int {
      dpp_pair_max(h, l);
      const uint32_t rl = (uint32_t)__builtin_amdgcn_readlane((int)l, 63), rh = (uint32_t)__builtin_amdgcn_readlane((int)h, 63);
      return rh == 0u ? 0 : (int)~rl;
    };
    int at = nt > 0 ? wave_best(bh0, bl0) : 0;
    if (src.epsilon_thr && nt > 0) {   // epsilon-greedy (do_agent.py:972-973)
      const cg_u32x4 rr = cg_philox4x32_10((uint32_t)(env_id_base + row), tick, CG_SITE_EPS_TYPE, 0u, (uint32_t)seed, (uint32_t)(seed >> 32));
      if ((uint64_t)rr.v[0] < src.epsilon_thr) at = (int)cg_index(rr.v[1], (uint32_t)nt);
    }
    if (nt > 0) at = nt <= WAVE ? __shfl(tmap, at) : (src.type_map ? src.type_map[at] : at);
    const int cnt = base < L ? base : L;
    for (int q = cnt + lane; q < L; q += WAVE) out[q] = 0;
    const int ex = src.n_exploits > 0 ? wave_best(bh1, bl1) : 0;
    const int app = src.n_apps > 0 ? wave_best(bh2, bl2) : 0;
    if (lane == 0) {
      const_cast<int32_t*>(dst.atype)[(size_t)row * G] = at;
      const_cast<int32_t*>(dst.exploit)[(size_t)row * G * CG_MAX_EXPLOITS] = ex;
      const_cast<int32_t*>(dst.n_exploit)[(size_t)row * G] = 1;
      const_cast<int32_t*>(dst.app)[(size_t)row * G] = app;
      const_cast<int32_t*>(dst.dev_cnt)[(size_t)row * G] = cnt;
      if (base > L && src.status) atomicOr(src.status, CG_DECODE_TRUNCATED);
    }
    MSTAMP(11);
  }
#undef MSTAMP
}

template <int HEAD_OPL, int VW>
__global__ __launch_bounds__(MLP_THREADS) void actor_mlp_kernel(cygym_actor_mlp ml, cygym_action_vectors src, cygym_actions dst, int n_envs,
                                                                const int32_t* ienv, uint64_t seed, int64_t env_id_base, unsigned long long* st,
                                                                MlpView vw) {
  actor_mlp_body<HEAD_OPL, VW, false>(ml, src, dst, n_envs, ienv, seed, env_id_base, st, vw, nullptr, 0);
}
