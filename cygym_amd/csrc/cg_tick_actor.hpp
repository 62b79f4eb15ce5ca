// cg_tick_actor.hpp -- tick_actor_kernel: cygym_step and the NEXT actor's cygym_actor_mlp_decode as ONE launch (cygym_step_actor).
// At the `target` shape (256 devices, a batch of at most 16 envs per CU) both kernels run as one 16-wave workgroup per CU over the
// same 16 envs, so the tick's workgroup can go straight on: its envs' flag planes are still in LDS, the actor builds the next
// role's view from them, runs its layers and writes those envs' next actions.  Saves a launch ramp, a graph node and the
// actor's view requests per tick of a closed loop.  The tick is the text of step_kernel (cg_tick_body.inc) in a force-inlined
// lambda: step_kernel itself is not touched.  Compiled in its own unit (cg_inst_actor.hip).
#ifndef CG_TICK_ACTOR_HPP
#define CG_TICK_ACTOR_HPP
// (the tick's shape parameters are template parameters so that the body's `if constexpr`s stay dependent, as in step_kernel)
template <int HEAD_OPL, int WPB = 16, int MT = 256, bool FUSED = false, bool XE = false, bool WIDE = true>
__global__ __launch_bounds__(16 * WAVE, 4) void tick_actor_kernel(const KParams P0, cygym_actor_mlp ml, cygym_action_vectors src, cygym_actions dst,
                                                                   MlpView vw) {
  {
    auto tick = [&]() __attribute__((always_inline)) {
#include "cg_tick_body.inc"
    };
    tick();
  }
  __syncthreads();   // every env of the workgroup has finished its tick (its planes and scalars are written back; the planes are still in LDS)
  extern __shared__ __align__(16) uint8_t smem_ta[];
  actor_mlp_body<HEAD_OPL, 0, true>(ml, src, dst, P0.n_envs, P0.b.ienv, P0.c.seed, P0.c.env_id_base, nullptr, vw, smem_ta + P0.shared_lds, P0.wave_lds);
}
#endif  // CG_TICK_ACTOR_HPP
