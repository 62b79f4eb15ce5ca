"""ctypes binding of libcygym_hip.so (the product's only compute path).

There is NO CPU fallback: if the shared library is missing or a call fails, this
module raises.  The CPU oracle under oracle/ is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.environ.get("CYGYM_SO") or os.path.join(HERE, "libcygym_hip.so")   # CYGYM_SO: A/B a different build

EXPORTS = [
    "cygym_version", "cygym_sizeof", "cygym_last_error", "cygym_create", "cygym_destroy", "cygym_set_config", "cygym_bind", "cygym_derive",
    "cygym_set_snapshot", "cygym_reset", "cygym_randomize", "cygym_step", "cygym_step_range", "cygym_rollout", "cygym_observe",
    "cygym_gen_actions", "cygym_write_actions", "cygym_decode_actions", "cygym_actor_head_decode", "cygym_actor_mlp_decode", "cygym_step_actor", "cygym_group_actions", "cygym_sample_group_actions", "cygym_fit_forests",
    "cygym_timer_start", "cygym_timer_stop", "cygym_launch_plan",
]


class CygymError(RuntimeError):
    """`code`: the library's return code (include/cygym_abi.h: CYGYM_EINVAL -1, CYGYM_EHIP -2, CYGYM_EUNSUPPORTED -3,
    CYGYM_ENOTBOUND -4), None for errors raised on the Python side."""
    code = None


EUNSUPPORTED = -3


_lib = None


def load():
    """Load the HIP library; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise CygymError(
            f"{SO} is missing: build it with `python -m cygym_amd.build` (hipcc --offload-arch=gfx950). "
            "cygym_amd has no CPU fallback.")
    L = C.CDLL(SO)
    H = C.c_void_p
    L.cygym_version.restype = C.c_int
    L.cygym_last_error.restype = C.c_char_p
    L.cygym_last_error.argtypes = [H]
    L.cygym_create.argtypes = [C.POINTER(abi.Topology), C.POINTER(abi.Config), C.c_int32, C.c_int32, C.POINTER(H)]
    L.cygym_destroy.argtypes = [H]
    L.cygym_destroy.restype = None
    L.cygym_set_config.argtypes = [H, C.POINTER(abi.Config)]
    L.cygym_bind.argtypes = [H, C.POINTER(abi.Buffers)]
    L.cygym_set_snapshot.argtypes = [H, C.POINTER(abi.Buffers)]
    L.cygym_derive.argtypes = [H, C.POINTER(abi.Buffers), C.c_void_p]
    L.cygym_reset.argtypes = [H, C.POINTER(abi.Buffers), C.c_void_p, C.c_int32, C.c_void_p]
    L.cygym_randomize.argtypes = [H, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.cygym_step.argtypes = [H, C.POINTER(abi.Actions), C.POINTER(abi.Outputs), C.c_void_p]
    L.cygym_step_range.argtypes = [H, C.c_int32, C.c_int32, C.POINTER(abi.Actions), C.POINTER(abi.Outputs), C.c_void_p]
    L.cygym_rollout.argtypes = [H, C.c_int32, C.POINTER(abi.Actions), C.POINTER(abi.Outputs), C.c_void_p]
    L.cygym_observe.argtypes = [H, C.c_int32, C.c_void_p, C.c_void_p]
    L.cygym_write_actions.argtypes = [H, C.POINTER(abi.ActionRows), C.POINTER(abi.Actions), C.c_void_p]
    L.cygym_decode_actions.argtypes = [H, C.POINTER(abi.ActionVectors), C.POINTER(abi.Actions), C.c_void_p]
    L.cygym_actor_head_decode.argtypes = [H, C.POINTER(abi.ActorHead), C.POINTER(abi.ActionVectors), C.POINTER(abi.Actions), C.c_void_p]
    L.cygym_actor_mlp_decode.argtypes = [H, C.POINTER(abi.ActorMlp), C.POINTER(abi.ActionVectors), C.POINTER(abi.Actions), C.c_void_p]
    L.cygym_step_actor.argtypes = [H, C.POINTER(abi.Actions), C.POINTER(abi.Outputs), C.POINTER(abi.ActorMlp), C.POINTER(abi.ActionVectors),
                                   C.POINTER(abi.Actions), C.c_void_p]
    L.cygym_group_actions.argtypes = [H, C.POINTER(abi.DeviceTypes), C.POINTER(abi.Actions), C.c_void_p]
    L.cygym_sample_group_actions.argtypes = [H, C.POINTER(abi.DeviceLogits), C.POINTER(abi.Actions), C.c_void_p]
    L.cygym_fit_forests.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.cygym_gen_actions.argtypes = [H, C.c_int32] + [C.c_void_p] * 8 + [C.c_int32, C.c_void_p]
    L.cygym_timer_start.argtypes = [H, C.c_void_p]
    L.cygym_timer_stop.argtypes = [H, C.c_void_p, C.POINTER(C.c_float)]
    L.cygym_launch_plan.argtypes = [H, C.POINTER(C.c_int32)]
    if L.cygym_version() != abi.ABI_VERSION:
        raise CygymError(f"ABI mismatch: library {L.cygym_version()} vs python {abi.ABI_VERSION}")
    L.cygym_sizeof.argtypes = [C.c_int32]
    for which, st in enumerate((abi.Topology, abi.Config, abi.Buffers, abi.Actions, abi.Outputs, abi.ActionRows, abi.ActionVectors, abi.ActorHead, abi.ActorMlp, abi.DeviceTypes, abi.DeviceLogits)):
        if L.cygym_sizeof(which) != C.sizeof(st):
            raise CygymError(f"ABI struct {st.__name__}: library {L.cygym_sizeof(which)} bytes vs python {C.sizeof(st)}")
    _lib = L
    return L


def check(rc: int, handle=None, what: str = ""):
    if rc != 0:
        msg = load().cygym_last_error(handle)
        err = CygymError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
        err.code = int(rc)
        raise err
