// cg_inst_actor.hip -- the instantiation unit of tick_actor_kernel (cg_tick_actor.hpp): the tick at 256 devices followed by the
// next actor's network, for the two output widths a 256-device action vector can have (n_out in 257..320 and 321..384).
#include "cg_device.hpp"
namespace cygym_k {
#include "cg_decode.hpp"
#include "cg_tick_actor.hpp"
template __global__ void tick_actor_kernel<5>(const KParams, cygym_actor_mlp, cygym_action_vectors, cygym_actions, MlpView);
template __global__ void tick_actor_kernel<6>(const KParams, cygym_actor_mlp, cygym_action_vectors, cygym_actions, MlpView);
}  // namespace cygym_k
