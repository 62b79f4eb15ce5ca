// cg_inst.hip -- one instantiation unit of the tick kernels: compiled once per group (-DCG_INST_GROUP=0..7, see
// CG_STEP_KERNELS in cg_device.hpp) so that the ~80 step_kernel variants build in parallel.
#if defined(CG_INST_GROUP) && CG_INST_GROUP < 4
#define CG_CBY_GLOBAL 0   // groups 0-3 hold the compile-time sizes (64 / 256 devices): comp_by is always staged in LDS there,
#endif                    // the global-memory arms of its accessors (cg_env.hpp) are not even compiled
#include "cg_device.hpp"
#ifndef CG_INST_GROUP
#error "compile with -DCG_INST_GROUP=<0..7>"
#endif
namespace cygym_k {
#define CG_INST(W, M, F, X, WD, G) CG_INST_IF(W, M, F, X, WD, G)
#define CG_INST_IF(W, M, F, X, WD, G) CG_INST_##G(W, M, F, X, WD)
#define CG_DO_INST(W, M, F, X, WD) template __global__ void step_kernel<W, M, F, X, WD>(const KParams);
#define CG_NO_INST(W, M, F, X, WD)
#if CG_INST_GROUP == 0
#define CG_INST_0 CG_DO_INST
#else
#define CG_INST_0 CG_NO_INST
#endif
#if CG_INST_GROUP == 1
#define CG_INST_1 CG_DO_INST
#else
#define CG_INST_1 CG_NO_INST
#endif
#if CG_INST_GROUP == 2
#define CG_INST_2 CG_DO_INST
#else
#define CG_INST_2 CG_NO_INST
#endif
#if CG_INST_GROUP == 3
#define CG_INST_3 CG_DO_INST
#else
#define CG_INST_3 CG_NO_INST
#endif
#if CG_INST_GROUP == 4
#define CG_INST_4 CG_DO_INST
#else
#define CG_INST_4 CG_NO_INST
#endif
#if CG_INST_GROUP == 5
#define CG_INST_5 CG_DO_INST
#else
#define CG_INST_5 CG_NO_INST
#endif
#if CG_INST_GROUP == 6
#define CG_INST_6 CG_DO_INST
#else
#define CG_INST_6 CG_NO_INST
#endif
#if CG_INST_GROUP == 7
#define CG_INST_7 CG_DO_INST
#else
#define CG_INST_7 CG_NO_INST
#endif
CG_STEP_KERNELS(CG_INST)
}  // namespace cygym_k
