"""ctypes driver for the CPU oracle (oracle/libcygym_oracle.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Holds env state in numpy arrays with the same struct-of-arrays
layout the HIP library uses (include/cygym_abi.h) and steps it with the scalar C
restatement.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from cygym_amd import abi
from cygym_amd import spec as S

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "libcygym_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "cygym_oracle.c")
    deps = [src, os.path.join(HERE, "..", "include", "cygym_abi.h"), os.path.join(HERE, "..", "include", "cygym_spec.h")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in deps):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-std=c11", "-I" + os.path.join(HERE, "..", "include"),
                               "-shared", "-o", SO, src, "-lm"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(SO)
        L.cgo_step.argtypes = [C.POINTER(abi.Topology), C.POINTER(abi.Config), C.POINTER(abi.Buffers),
                               C.POINTER(abi.Actions), C.POINTER(abi.Outputs), C.POINTER(abi.Buffers),
                               C.c_int32, C.c_int32]
        L.cgo_reset.argtypes = [C.POINTER(abi.Topology), C.POINTER(abi.Config), C.POINTER(abi.Buffers),
                                C.POINTER(abi.Buffers), C.c_void_p, C.c_int32]
        L.cgo_randomize.argtypes = [C.POINTER(abi.Topology), C.POINTER(abi.Config), C.POINTER(abi.Buffers),
                                    C.c_void_p, C.c_int32]
        L.cgo_observe.argtypes = [C.POINTER(abi.Topology), C.POINTER(abi.Config), C.POINTER(abi.Buffers),
                                  C.c_int32, C.c_void_p, C.c_int32]
        _lib = L
    return _lib


OTHER_SHAPES = {"blocked": ("EW", np.uint32), "ring": ("R", np.uint16), "ienv": ("I", np.int32), "fenv": ("D", np.float64)}
STATE_KEYS = abi.STATE_PLANES + tuple(OTHER_SHAPES)


def alloc_state(n, M, EW, K=0, detector=False, anomaly=False):
    """Struct-of-arrays state.  `live` / `stash` are the [N][4][M] buffers of the ABI; the
    per-plane entries (flags, busy, ..., st_comp_by) are numpy VIEWS into them."""
    dims = {"EW": (EW,), "R": (S.LOG_RING, 2), "I": (S.I_COUNT,), "D": (S.D_COUNT,)}
    st = {"live": np.zeros((n, 4, M), np.uint8), "stash": np.zeros((n, 4, M), np.uint8)}
    for i, k in enumerate(abi.LIVE_PLANES):
        st[k] = st["live"][:, i]
    for i, k in enumerate(abi.STASH_PLANES):
        st[k] = st["stash"][:, i]
    for k, (d, dt) in OTHER_SHAPES.items():
        st[k] = np.zeros((n,) + dims[d], dt)
    st["blocked_in"] = np.zeros((n, EW), np.uint32)   # derived, library-side only: the oracle never reads it
    st["extra"] = np.zeros((n, abi.x_words(K)), np.uint32)   # edges added by evolve_network (K = max_extra)
    # trained-detector mode: the env's flattened isolation forest and the long comm-log history it is fitted on
    st["forest"] = np.zeros((n, S.FOREST_WORDS if detector else 0), np.uint32)
    st["hist"] = np.full((n, S.HIST_RING if detector else 0, 2), 0xFFFF, np.uint16)
    st["anomaly"] = np.zeros((n, M if anomaly else 0), np.float32)   # per-env Device.anomaly_score (slow scan path only)
    st["ring"][:] = 0xFFFF
    return st


def state_struct(st) -> abi.Buffers:
    b = abi.Buffers()
    for k in abi.BUFFER_FIELDS:
        a = st[k]
        assert a.flags["C_CONTIGUOUS"], k
        setattr(b, k, a.ctypes.data if a.size else None)
    b.n_envs = st["live"].shape[0]
    return b


def copy_state(st):
    out = {k: st[k].copy() for k in abi.BUFFER_FIELDS}
    for i, k in enumerate(abi.LIVE_PLANES):
        out[k] = out["live"][:, i]
    for i, k in enumerate(abi.STASH_PLANES):
        out[k] = out["stash"][:, i]
    return out


def alloc_actions(n, G, L):
    return dict(mode=np.zeros(n, np.int32), n_groups=np.zeros(n, np.int32), atype=np.zeros((n, G), np.int32),
                n_exploit=np.zeros((n, G), np.int32), exploit=np.full((n, G, S.MAX_EXPLOITS), -1, np.int32),
                app=np.full((n, G), -1, np.int32), dev_cnt=np.zeros((n, G), np.int32),
                dev_idx=np.zeros((n, L), np.int16))


def actions_struct(act) -> abi.Actions:
    a = abi.Actions()
    for k in ("mode", "n_groups", "atype", "n_exploit", "exploit", "app", "dev_cnt", "dev_idx"):
        assert act[k].flags["C_CONTIGUOUS"], k
        setattr(a, k, act[k].ctypes.data)
    a.max_groups = act["atype"].shape[1]
    a.max_devs = act["dev_idx"].shape[1]
    return a


class OracleBatch:
    """N envs over one shared topology, stepped by the C oracle."""

    def __init__(self, topo: abi.TopologyArrays, cfg: abi.EnvConfig, n_envs: int, detector: bool = False):
        self.topo = topo.normalised()
        if detector and self.topo.det_apl is None:
            from cygym_amd import detector as D
            self.topo.det_apl = D.apl_table()
        self.slow_scan = not cfg.fast_scan       # the per-log scan path: long history + per-env anomaly scores
        self.detector = bool(detector) or self.slow_scan
        if self.detector and self.topo.det_apl is None:
            from cygym_amd import detector as D
            self.topo.det_apl = D.apl_table()
        self.topo.validate()
        self.cfg = cfg
        self.N = n_envs
        self.M = self.topo.M
        self.state = alloc_state(n_envs, self.M, self.topo.EW, self.topo.max_extra, self.detector, self.slow_scan)
        self.snapshot = None
        self.obs = np.zeros((n_envs, self.M, 6), np.float32)
        self.raw = np.zeros(n_envs, np.float64)
        self.shaped = np.zeros(n_envs, np.float64)
        self.done = np.zeros(n_envs, np.uint8)
        self._t = self.topo.to_c()

    def load_state(self, init: dict, broadcast=True):
        """init: dict of arrays with leading dim 1 or N (planes may be wider ints)."""
        for k in STATE_KEYS:
            src = np.asarray(init[k])
            dt = self.state[k].dtype
            if k == "blocked" and (src.shape[-1] != self.topo.EW or src.dtype not in (np.uint32, np.int32)):
                src = abi.pack_blocked(src, self.topo.EW)
            if k == "ring":
                src = np.where(src.astype(np.int64) < 0, 0xFFFF, src)
            self.state[k][...] = src.astype(dt) if src.shape[0] == self.N else np.broadcast_to(src.astype(dt), self.state[k].shape)
        self.state["extra"][...] = 0
        if "extra" in init and self.state["extra"].size:
            src = np.asarray(init["extra"], np.uint32)
            self.state["extra"][...] = src if src.shape[0] == self.N else np.broadcast_to(src, self.state["extra"].shape)
        for k, fill in (("forest", 0), ("hist", 0xFFFF)):
            self.state[k][...] = fill
            if k in init and self.state[k].size:
                src = np.asarray(init[k]).astype(self.state[k].dtype)
                self.state[k][...] = src if src.shape[0] == self.N else np.broadcast_to(src, self.state[k].shape)
        if self.state["anomaly"].size:
            src = np.asarray(init["anomaly"], np.float32) if "anomaly" in init else self.topo.anomaly[None]
            self.state["anomaly"][...] = src if src.shape[0] == self.N else np.broadcast_to(src, self.state["anomaly"].shape)
        self.snapshot = copy_state(self.state)

    def install_forest(self, env: int, words):
        """What the host does after Detector.train: write the flattened forest (header words 3, 4 -- the
        request the tick recorded -- stay) and clear CG_E_DET_PENDING."""
        f = self.state["forest"][env]
        w = np.asarray(words, np.uint32)
        f[0:3] = w[0:3]
        f[5] = f[3]
        f[7] = w[7]
        f[S.FOREST_HDR:] = w[S.FOREST_HDR:]
        self.state["ienv"][env, S.I_FLAGS] &= ~S.E_DET_PENDING

    def step(self, act: dict, begin=0, end=None):
        c = self.cfg.to_c()
        b = state_struct(self.state)
        a = actions_struct(act)
        o = abi.Outputs()
        o.obs, o.raw, o.shaped, o.done = (self.obs.ctypes.data, self.raw.ctypes.data,
                                          self.shaped.ctypes.data, self.done.ctypes.data)
        snap = state_struct(self.snapshot) if self.snapshot is not None else None
        rc = lib().cgo_step(C.byref(self._t), C.byref(c), C.byref(b), C.byref(a), C.byref(o),
                            C.byref(snap) if snap is not None else None, begin, self.N if end is None else end)
        if rc != 0:
            raise RuntimeError(f"cgo_step failed: {rc}")
        return self.obs, self.raw, self.shaped, self.done

    def randomize(self, env_ids=None):
        c = self.cfg.to_c()
        b = state_struct(self.state)
        if env_ids is None:
            rc = lib().cgo_randomize(C.byref(self._t), C.byref(c), C.byref(b), None, self.N)
        else:
            ids = np.ascontiguousarray(env_ids, np.int32)
            rc = lib().cgo_randomize(C.byref(self._t), C.byref(c), C.byref(b), ids.ctypes.data, len(ids))
        assert rc == 0

    def reset(self, env_ids=None):
        c = self.cfg.to_c()
        b = state_struct(self.state)
        s = state_struct(self.snapshot)
        if env_ids is None:
            rc = lib().cgo_reset(C.byref(self._t), C.byref(c), C.byref(b), C.byref(s), None, self.N)
        else:
            ids = np.ascontiguousarray(env_ids, np.int32)
            rc = lib().cgo_reset(C.byref(self._t), C.byref(c), C.byref(b), C.byref(s), ids.ctypes.data, len(ids))
        assert rc == 0

    def observe(self, role: int):
        c = self.cfg.to_c()
        b = state_struct(self.state)
        width = 6 * self.M if role in (0, 1) else 4 * self.M + self.cfg.max_exploits
        out = np.zeros((self.N, width), np.float32)
        rc = lib().cgo_observe(C.byref(self._t), C.byref(c), C.byref(b), role, out.ctypes.data, self.N)
        assert rc == 0
        return out
