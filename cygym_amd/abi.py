"""ctypes mirror of include/cygym_abi.h (struct layouts + config helpers).

Pure layout code: no compute.  Used by the product loader (cygym_amd/_lib.py) and,
in tests only, to drive the CPU oracle through the same structs.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import rng as R
from . import spec as S

ABI_VERSION = 4

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
i16p = C.POINTER(C.c_int16)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)


class Topology(C.Structure):
    _fields_ = [
        ("n_devices", C.c_int32), ("n_exploits", C.c_int32), ("n_edges", C.c_int32), ("max_extra_edges", C.c_int32),
        ("dstatic", C.c_void_p), ("vuln", C.c_void_p), ("napps", C.c_void_p),
        ("os_val", C.c_void_p), ("version", C.c_void_p), ("anomaly", C.c_void_p),
        ("out_ptr", C.c_void_p), ("out_col", C.c_void_p), ("in_ptr", C.c_void_p),
        ("in_col", C.c_void_p), ("in_eid", C.c_void_p), ("det_apl", C.c_void_p),
    ]


class Config(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("env_id_base", C.c_int64),
        ("num_of_device", C.c_int32), ("min_network_size", C.c_int32), ("max_exploits", C.c_int32),
        ("evolve_period", C.c_int32), ("workload_cap", C.c_int32), ("workload_period_base", C.c_int32),
        ("workload_period_max", C.c_int32), ("scaling_vulnerability", C.c_int32), ("fast_scan", C.c_int32),
        ("n_att_actions", C.c_int32), ("n_def_actions", C.c_int32), ("zero_day", C.c_int32),
        ("zero_day_owned_mask", C.c_int32), ("default_high", C.c_int32), ("baseline", C.c_int32),
        ("auto_reset", C.c_int32), ("episode_limit", C.c_int32), ("turbo", C.c_int32),
        ("work_scale", C.c_double), ("comp_scale", C.c_double), ("def_scale", C.c_double), ("gamma", C.c_double),
        ("p_add_thr", C.c_uint64), ("p_attacker_thr", C.c_uint64),
        ("poisson_thr", C.c_uint64 * S.POISSON_TABLE), ("tri_thr", C.c_uint64 * S.TRI_TABLE),
        ("turbo_fraction_clients", C.c_double), ("turbo_fraction_servers", C.c_double),
        ("turbo_max_clients", C.c_int32), ("turbo_max_servers", C.c_int32), ("turbo_ramp_steps", C.c_int32),
        ("turbo_train_max_logs", C.c_int32), ("turbo_train_stride", C.c_int32), ("reserved2", C.c_int32),
    ]


class Buffers(C.Structure):
    _fields_ = [
        ("live", C.c_void_p), ("stash", C.c_void_p),
        ("blocked", C.c_void_p), ("blocked_in", C.c_void_p), ("ring", C.c_void_p), ("ienv", C.c_void_p),
        ("fenv", C.c_void_p), ("extra", C.c_void_p), ("forest", C.c_void_p), ("hist", C.c_void_p), ("anomaly", C.c_void_p),
        ("n_envs", C.c_int32), ("reserved", C.c_int32),
    ]


class Actions(C.Structure):
    _fields_ = [
        ("mode", C.c_void_p), ("n_groups", C.c_void_p), ("atype", C.c_void_p), ("n_exploit", C.c_void_p),
        ("exploit", C.c_void_p), ("app", C.c_void_p), ("dev_cnt", C.c_void_p), ("dev_idx", C.c_void_p),
        ("max_groups", C.c_int32), ("max_devs", C.c_int32),
    ]


class Outputs(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("raw", C.c_void_p), ("shaped", C.c_void_p), ("done", C.c_void_p),
                ("obs_def", C.c_void_p), ("obs_att", C.c_void_p), ("ret", C.c_void_p), ("alive", C.c_void_p),
                ("status", C.c_void_p)]


class ActionRows(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("atype", C.c_void_p), ("exploit", C.c_void_p), ("app", C.c_void_p),
                ("dev_mask", C.c_void_p), ("dev_idx", C.c_void_p), ("dev_cnt", C.c_void_p),
                ("n", C.c_int32), ("reserved", C.c_int32)]


class ActionVectors(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("vec", C.c_void_p), ("type_map", C.c_void_p), ("stride", C.c_int32),
                ("n_types", C.c_int32), ("n_devices", C.c_int32), ("n_exploits", C.c_int32), ("n_apps", C.c_int32),
                ("n", C.c_int32), ("epsilon_thr", C.c_uint64), ("status", C.c_void_p)]


class ActorHead(C.Structure):
    _fields_ = [("hidden", C.c_void_p), ("weight_t", C.c_void_p), ("bias", C.c_void_p), ("H", C.c_int32),
                ("hidden_stride", C.c_int32), ("tanh_out", C.c_int32), ("weight_pitch", C.c_int32),
                ("n_groups", C.c_int32), ("rows_per_group", C.c_int32)]


MLP_MAX_HIDDEN = 3


class ActorMlp(C.Structure):
    _fields_ = [("obs", C.c_void_p), ("w", C.c_void_p * MLP_MAX_HIDDEN), ("b", C.c_void_p * MLP_MAX_HIDDEN),
                ("w_head", C.c_void_p), ("b_head", C.c_void_p), ("obs_stride", C.c_int32), ("K", C.c_int32),
                ("n_hidden", C.c_int32), ("width", C.c_int32 * MLP_MAX_HIDDEN), ("tanh_out", C.c_int32),
                ("obs_by_env", C.c_int32), ("obs_role", C.c_int32), ("reserved", C.c_int32), ("n_groups", C.c_int32),
                ("rows_per_group", C.c_int32)]


class DeviceTypes(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("types", C.c_void_p), ("visible", C.c_void_p), ("exploit", C.c_void_p), ("app", C.c_void_p),
                ("n", C.c_int32), ("n_types", C.c_int32), ("noop", C.c_int32), ("role", C.c_int32), ("single_mask", C.c_uint32),
                ("reserved", C.c_int32), ("status", C.c_void_p)]


class DeviceLogits(C.Structure):
    _fields_ = [("rows", C.c_void_p), ("logits", C.c_void_p), ("exp_logits", C.c_void_p), ("app_logits", C.c_void_p),
                ("types_out", C.c_void_p), ("exp_out", C.c_void_p), ("app_out", C.c_void_p), ("logp_out", C.c_void_p),
                ("n", C.c_int32), ("n_types", C.c_int32), ("n_exp", C.c_int32), ("n_app", C.c_int32), ("noop", C.c_int32),
                ("role", C.c_int32), ("single_mask", C.c_uint32), ("greedy", C.c_int32), ("status", C.c_void_p)]


DECODE_TRUNCATED = 0x10000

BASELINES = {"Nash": 0, "No Defense": 1, "Preset": 2, "No Attack": 3}

LIVE_PLANES = ("flags", "busy", "wl", "comp_by")          # order inside Buffers.live  [N][4][M]
STASH_PLANES = ("st_flags", "st_busy", "st_wl", "st_comp_by")  # order inside Buffers.stash [N][4][M]
STATE_PLANES = LIVE_PLANES + STASH_PLANES
BUFFER_FIELDS = ("live", "stash", "blocked", "blocked_in", "ring", "ienv", "fenv", "extra", "forest", "hist", "anomaly")


@dataclass
class EnvConfig:
    """Scalar knobs of the reference env object (volt_typhoon_env.py:32-117,
    CyberDefenseEnv.py:19-62), with the reference's defaults."""
    seed: int = 0
    env_id_base: int = 0
    num_of_device: int = 3
    min_network_size: int = 2
    max_exploits: int = 6
    evolve_period: int = 2
    workload_cap: int = -1
    workload_period_base: int = 50
    workload_period_max: int = 200
    scaling_vulnerability: int = 1
    fast_scan: int = 1
    n_att_actions: int = 5
    n_def_actions: int = 14
    zero_day: int = 0
    zero_day_owned_mask: int = 0
    default_high: int = 3
    baseline: str = "Nash"
    auto_reset: int = 0
    episode_limit: int = 1000
    work_scale: float = 1.0
    comp_scale: float = 50.0
    def_scale: float = 1.0
    gamma: float = 0.99
    lambda_events: float = 0.7
    p_add: float = 0.1
    p_attacker: float = 0.0
    tri_mode: float = 2.0
    tri_high: float = 5.0
    turbo: int = 0                          # volt_typhoon_env.py:92 and its knobs :97-109
    turbo_fraction_clients: float = 0.05
    turbo_fraction_servers: float = 0.02
    turbo_max_clients: int = 200
    turbo_max_servers: int = 40
    turbo_ramp_steps: int = 200
    turbo_train_max_logs: int = 256
    turbo_train_stride: int = 2

    def to_c(self) -> Config:
        c = Config()
        for name, _ in Config._fields_:
            if name in ("baseline", "p_add_thr", "p_attacker_thr", "poisson_thr", "tri_thr", "reserved2"):
                continue
            setattr(c, name, getattr(self, name))
        c.baseline = BASELINES[self.baseline] if isinstance(self.baseline, str) else int(self.baseline)
        c.p_add_thr = R.bernoulli_threshold(self.p_add)
        c.p_attacker_thr = R.bernoulli_threshold(self.p_attacker)
        for i, t in enumerate(R.poisson_table(self.lambda_events, S.POISSON_TABLE)):
            c.poisson_thr[i] = t
        for i, t in enumerate(R.triangular_ceil_table(self.tri_mode, self.tri_high, S.TRI_TABLE)):
            c.tri_thr[i] = t
        return c


@dataclass
class TopologyArrays:
    """Host-side (numpy) shared topology + static per-device columns."""
    M: int
    X: int
    dstatic: np.ndarray
    vuln: np.ndarray
    napps: np.ndarray
    os_val: np.ndarray
    version: np.ndarray
    anomaly: np.ndarray
    out_ptr: np.ndarray
    out_col: np.ndarray
    in_ptr: np.ndarray
    in_col: np.ndarray
    in_eid: np.ndarray
    max_extra: int = 0          # K: capacity of the per-env list of edges evolve_network may add
    det_apl: np.ndarray | None = None   # f64 [DET_APL_N] leaf-term table of the trained detector (detector.apl_table())
    _keep: list = field(default_factory=list, repr=False)

    @property
    def E(self) -> int:
        return int(self.out_col.shape[0])

    @property
    def EW(self) -> int:
        return max(1, (self.E + 31) // 32)

    @property
    def XW(self) -> int:
        """words of one env's extra-edge list (cygym_spec.h CG_X_WORDS)"""
        return x_words(self.max_extra)

    def normalised(self) -> "TopologyArrays":
        def a(x, dt):
            return np.ascontiguousarray(np.asarray(x, dtype=dt))
        return TopologyArrays(
            int(self.M), int(self.X), a(self.dstatic, np.uint8), a(self.vuln, np.uint8), a(self.napps, np.uint8),
            a(self.os_val, np.float32), a(self.version, np.float32), a(self.anomaly, np.float32),
            a(self.out_ptr, np.int32), a(self.out_col, np.int32), a(self.in_ptr, np.int32),
            a(self.in_col, np.int32), a(self.in_eid, np.int32), int(self.max_extra),
            None if self.det_apl is None else a(self.det_apl, np.float64))

    def validate(self):
        M, E = self.M, self.E
        if not (1 <= M <= 2048):
            raise ValueError(f"n_devices must be in [1, 2048], got {M}")
        if not (0 <= self.X <= S.MAX_EXPLOITS):
            raise ValueError("n_exploits out of range")
        for name in ("dstatic", "vuln", "napps", "os_val", "version", "anomaly"):
            if getattr(self, name).shape != (M,):
                raise ValueError(f"{name} must have shape ({M},)")
        for p, c in ((self.out_ptr, self.out_col), (self.in_ptr, self.in_col)):
            if p.shape != (M + 1,) or p[0] != 0 or p[-1] != E or np.any(np.diff(p) < 0):
                raise ValueError("malformed CSR row pointer")
            if c.shape != (E,) or (E and (c.min() < 0 or c.max() >= M)):
                raise ValueError("malformed CSR column array")
        if not (0 <= self.max_extra <= 4096):
            raise ValueError("max_extra must be in [0, 4096]")
        if self.in_eid.shape != (E,):
            raise ValueError("in_eid must have shape (E,)")
        if self.det_apl is not None and self.det_apl.shape != (S.DET_APL_N,):
            raise ValueError(f"det_apl must have shape ({S.DET_APL_N},)")
        if E:
            if self.in_eid.min() < 0 or self.in_eid.max() >= E:
                raise ValueError("in_eid out of range")
            # every in-entry (u -> v) must point at an out slot of u whose column is v
            src = np.repeat(np.arange(M), np.diff(self.out_ptr))
            dst_of_in = np.repeat(np.arange(M), np.diff(self.in_ptr))
            if np.any(src[self.in_eid] != self.in_col) or np.any(self.out_col[self.in_eid] != dst_of_in):
                raise ValueError("in_eid does not match the out-CSR")

    def to_c(self) -> Topology:
        t = Topology()
        t.n_devices, t.n_exploits, t.n_edges = self.M, self.X, self.E
        t.max_extra_edges = int(self.max_extra)
        for name in ("dstatic", "vuln", "napps", "os_val", "version", "anomaly",
                     "out_ptr", "out_col", "in_ptr", "in_col", "in_eid"):
            arr = getattr(self, name)
            if arr.size == 0:  # keep a valid pointer for empty edge sets
                arr = np.zeros(1, arr.dtype)
                self._keep.append(arr)
            setattr(t, name, arr.ctypes.data)
        t.det_apl = None if self.det_apl is None else self.det_apl.ctypes.data
        return t


def x_words(K: int) -> int:
    return int(K) + (int(K) + 31) // 32


def pack_extra(edges, blocked, K: int) -> np.ndarray:
    """(u, v) pairs + 0/1 blocked flags -> one env's extra-edge list u32 [x_words(K)] (keys ascending)."""
    row = np.zeros(x_words(K), np.uint32)
    keys = sorted(((int(u) << 16) | int(v), int(b)) for (u, v), b in zip(edges, blocked))
    if len(keys) > K:
        raise ValueError(f"{len(keys)} extra edges do not fit max_extra={K}")
    for j, (k, b) in enumerate(keys):
        row[j] = k
        if b:
            row[K + (j >> 5)] |= np.uint32(1 << (j & 31))
    return row


def unpack_extra(row: np.ndarray, n: int, K: int):
    """-> [(u, v, blocked)] of the first n live entries"""
    row = np.asarray(row, np.uint32)
    return [(int(row[j]) >> 16, int(row[j]) & 0xFFFF, int((int(row[K + (j >> 5)]) >> (j & 31)) & 1)) for j in range(n)]


def build_in_csr(M: int, out_ptr: np.ndarray, out_col: np.ndarray):
    """In-CSR (rows sorted by source id, the order python-igraph reports) with the
    out-slot of every entry."""
    E = len(out_col)
    src = np.repeat(np.arange(M, dtype=np.int64), np.diff(out_ptr))
    order = np.lexsort((np.arange(E), src, out_col))  # by (dst, src, slot)
    in_col = src[order].astype(np.int32)
    in_eid = order.astype(np.int32)
    counts = np.bincount(out_col, minlength=M) if E else np.zeros(M, np.int64)
    in_ptr = np.zeros(M + 1, np.int32)
    in_ptr[1:] = np.cumsum(counts)
    return in_ptr, in_col, in_eid


def pack_blocked(bits: np.ndarray, EW: int) -> np.ndarray:
    """[..., E] 0/1 -> [..., EW] uint32 (bit k of word k>>5)."""
    bits = np.asarray(bits, np.uint8)
    lead = bits.shape[:-1]
    E = bits.shape[-1]
    pad = np.zeros(lead + (EW * 32,), np.uint8)
    pad[..., :E] = bits
    w = pad.reshape(lead + (EW, 32)).astype(np.uint32)
    return (w << np.arange(32, dtype=np.uint32)).sum(axis=-1, dtype=np.uint64).astype(np.uint32)


def unpack_blocked(words: np.ndarray, E: int) -> np.ndarray:
    words = np.asarray(words, np.uint32)
    bits = (words[..., :, None] >> np.arange(32, dtype=np.uint32)) & np.uint32(1)
    return bits.reshape(words.shape[:-1] + (-1,))[..., :E].astype(np.uint8)
