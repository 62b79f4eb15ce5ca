// cg_device.hpp -- everything device-side of the tick path, shared by the translation units of libcygym_hip.so:
// the C-ABI file (cygym_hip.hip: host code + the small auxiliary kernels) and the instantiation units
// (cg_inst.hip compiled once per CG_INST_GROUP: the step_kernel<WPB, MT, FUSED, XE, WIDE> variants, ~80 of them, which
// cygym_amd/build.py compiles in parallel).  A NAMED namespace: the instantiations are referenced across units.
#ifndef CG_DEVICE_HPP
#define CG_DEVICE_HPP
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <new>
#include "cygym_abi.h"

#ifndef CG_FUSED_LB
#define CG_FUSED_LB 4   // rollout kernels: 4 waves per SIMD (128 VGPRs); see the tick-loop note in cg_tick.hpp
#endif
#ifndef CG_LB
// Full-feature per-tick kernels: 4 waves per SIMD (128 VGPRs).  At 6 (80 VGPRs) they spilled 5-11 VGPRs on top of
// ~150 SGPRs kept in VGPR lanes, and with the parameter block read through the laundered kernarg pointer that
// combination miscompiled: a spilled SGPR pair (an f64 env accumulator) came back clobbered after the divergent
// block / unblock code at run-time sizes (caught by every full-feature fixture test).  No VGPR spills, no problem.
#define CG_LB 4
#endif
#define CG_E_STAR_OK 0x80  // kernel-private: star edges verified for the current owned set

namespace cygym_k {
#include "cg_params.hpp"
#include "cg_wave.hpp"
#include "cg_env.hpp"
#include "cg_extra_edges.hpp"
#include "cg_defender.hpp"
#include "cg_attacker.hpp"
#include "cg_arrivals.hpp"
#include "cg_evolve.hpp"
#include "cg_tick.hpp"
#ifdef CG_MAIN_UNIT
#include "cg_aux_kernels.hpp"   // plain (non-template) kernels: defined in the C-ABI unit only
#endif
}  // namespace cygym_k

// Development aid (cygym_amd/build.py build_to(dev_mt=...), tools/devbuild.py): -DCG_DEV_MT=<0|64|256> keeps only the
// kernels of that device-count class (build.py then compiles only their instantiation groups).  Handles of any other
// class fail at cygym_create.
#ifdef CG_DEV_MT
#define CG_HAS_MT(m) ((m) == CG_DEV_MT)
#else
#define CG_HAS_MT(m) 1
#endif

// The instantiation table: CG_STEP_KERNELS(X) expands X(WPB, MT, FUSED, XE, WIDE, GROUP) for every variant the library
// ships.  GROUP is the translation unit (cg_inst.hip -DCG_INST_GROUP=g) that holds the code.
#define CG_SHAPES_CT(X, MT, F, XE, G) X(16, MT, F, XE, false, G) X(8, MT, F, XE, false, G) X(4, MT, F, XE, false, G) X(2, MT, F, XE, false, G) X(1, MT, F, XE, false, G)
#define CG_SHAPES_RT(X, F, XE, G) CG_SHAPES_CT(X, 0, F, XE, G) X(12, 0, F, XE, false, G) X(6, 0, F, XE, false, G) X(5, 0, F, XE, false, G) X(3, 0, F, XE, false, G)
#define CG_STEP_KERNELS(X)                                                                  \
  CG_SHAPES_CT(X, 256, false, false, 0) CG_SHAPES_CT(X, 256, false, true, 0) X(16, 256, false, false, true, 0) \
  CG_SHAPES_CT(X, 256, true, false, 1) CG_SHAPES_CT(X, 256, true, true, 1)                   \
  CG_SHAPES_CT(X, 64, false, false, 2) CG_SHAPES_CT(X, 64, false, true, 2)                   \
  CG_SHAPES_CT(X, 64, true, false, 3) CG_SHAPES_CT(X, 64, true, true, 3)                     \
  CG_SHAPES_RT(X, false, false, 4) CG_SHAPES_RT(X, false, true, 5)                           \
  CG_SHAPES_RT(X, true, false, 6) CG_SHAPES_RT(X, true, true, 7)
#define CG_INST_GROUPS 8
#endif  // CG_DEVICE_HPP
