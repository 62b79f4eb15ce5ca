"""BatchedCyberDefenseEnv: N independent CyGym environments stepped at once on one
MI355X by the hand-written HIP kernels in libcygym_hip.so.

The Python host only owns memory (torch tensors in HBM) and marshals pointers
through the C ABI (include/cygym_abi.h); all per-tick work happens in the kernels.
This is the batched surface described in SURVEY.md section 8b; the per-env view
that mirrors the reference's `Volt_Typhoon_CyberDefenseEnv` method surface lives
in cygym_amd/env_view.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, abi
from . import spec as S

_BUF_DTYPES = {"live": torch.uint8, "stash": torch.uint8, "blocked": torch.int32, "blocked_in": torch.int32,
               "ring": torch.int16,
               "ienv": torch.int32, "fenv": torch.float64, "extra": torch.int32}
_STATE_KEYS = abi.STATE_PLANES + ("blocked", "ring", "ienv", "fenv")
_NP_VIEW = {"blocked": np.uint32, "ring": np.uint16, "extra": np.uint32}


def _alloc_state(n, M, EW, device, K=0):
    """`live` / `stash` are the [N][4][M] buffers of the ABI; flags/busy/... are VIEWS into them.
    `extra` is the per-env list of edges evolve_network added (K = topo.max_extra entries)."""
    dims = {"live": (4, M), "stash": (4, M), "blocked": (EW,), "blocked_in": (EW,), "ring": (S.LOG_RING, 2),
            "ienv": (S.I_COUNT,), "fenv": (S.D_COUNT,), "extra": (abi.x_words(K),)}
    st = {k: torch.zeros((n,) + dims[k], dtype=dt, device=device) for k, dt in _BUF_DTYPES.items()}
    for i, k in enumerate(abi.LIVE_PLANES):
        st[k] = st["live"][:, i]
    for i, k in enumerate(abi.STASH_PLANES):
        st[k] = st["stash"][:, i]
    return st


def _buffers_struct(st) -> abi.Buffers:
    b = abi.Buffers()
    for k in abi.BUFFER_FIELDS:
        t = st[k]
        assert t.is_contiguous()
        setattr(b, k, t.data_ptr() if t.numel() else None)
    b.n_envs = st["live"].shape[0]
    return b


def initial_state_numpy(topo: abi.TopologyArrays, *, flags, busy=None, wl=None, comp_by=None, blocked=None):
    """Assemble a single-env initial state dict (numpy) from the live planes."""
    M, EW = topo.M, topo.EW
    z = lambda: np.zeros((1, M), np.uint8)  # noqa: E731
    st = {k: z() for k in ("flags", "busy", "wl", "comp_by", "st_flags", "st_busy", "st_wl", "st_comp_by")}
    st["flags"][0] = flags
    if busy is not None: st["busy"][0] = busy
    if wl is not None: st["wl"][0] = wl
    if comp_by is not None: st["comp_by"][0] = comp_by
    st["blocked"] = np.zeros((1, EW), np.uint32) if blocked is None else abi.pack_blocked(np.asarray(blocked)[None], EW)
    st["ring"] = np.full((1, S.LOG_RING, 2), 0xFFFF, np.uint16)
    st["ienv"] = np.zeros((1, S.I_COUNT), np.int32)
    st["fenv"] = np.zeros((1, S.D_COUNT), np.float64)
    return st


class BatchedCyberDefenseEnv:
    """N envs over one shared topology on one GPU.

    Parameters
    ----------
    topo : abi.TopologyArrays     shared topology + static per-device columns
    cfg  : abi.EnvConfig          scalar knobs (reference attribute names)
    n_envs : int                  envs in this shard
    init_state : dict             numpy planes with leading dim 1 (broadcast) or n_envs
    device : torch device         e.g. "cuda:0"
    max_groups, max_devs          capacity of the action tensors (G, L)
    """

    def __init__(self, topo: abi.TopologyArrays, cfg: abi.EnvConfig, n_envs: int, init_state: dict,
                 device="cuda:0", max_groups: int = 1, max_devs: int | None = None):
        self.lib = _lib.load()   # raises when libcygym_hip.so is missing: no fallback
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.CygymError("BatchedCyberDefenseEnv runs on a ROCm GPU only (device must be cuda:N)")
        self.topo = topo.normalised()
        self.topo.validate()
        self.cfg = cfg
        self.N, self.M, self.EW = int(n_envs), self.topo.M, self.topo.EW
        self.G = int(max_groups)
        self.L = int(max_devs if max_devs is not None else max(1, self.M))
        self._h = C.c_void_p()
        t = self.topo.to_c()
        c = cfg.to_c()
        with torch.cuda.device(self.device):
            rc = self.lib.cygym_create(C.byref(t), C.byref(c), self.N, self.device.index or 0, C.byref(self._h))
        _lib.check(rc, None, "cygym_create")
        self.K = self.topo.max_extra
        self.state = _alloc_state(self.N, self.M, self.EW, self.device, self.K)
        lead = int(np.asarray(init_state["flags"]).shape[0])
        if lead not in (1, self.N):
            raise ValueError("init_state must have leading dimension 1 or n_envs")
        self.snapshot = _alloc_state(lead, self.M, self.EW, self.device, self.K)
        self._load(self.snapshot, init_state)
        _lib.check(self.lib.cygym_bind(self._h, C.byref(_buffers_struct(self.state))), self._h, "cygym_bind")
        self._snap_struct = _buffers_struct(self.snapshot)
        self._derive(self.snapshot)
        _lib.check(self.lib.cygym_set_snapshot(self._h, C.byref(self._snap_struct)), self._h, "cygym_set_snapshot")
        dev = self.device
        self.act = dict(
            mode=torch.zeros(self.N, dtype=torch.int32, device=dev),
            n_groups=torch.zeros(self.N, dtype=torch.int32, device=dev),
            atype=torch.zeros((self.N, self.G), dtype=torch.int32, device=dev),
            n_exploit=torch.zeros((self.N, self.G), dtype=torch.int32, device=dev),
            exploit=torch.full((self.N, self.G, S.MAX_EXPLOITS), -1, dtype=torch.int32, device=dev),
            app=torch.full((self.N, self.G), -1, dtype=torch.int32, device=dev),
            dev_cnt=torch.zeros((self.N, self.G), dtype=torch.int32, device=dev),
            dev_idx=torch.zeros((self.N, self.L), dtype=torch.int16, device=dev),
        )
        self.obs = torch.zeros((self.N, self.M, 6), dtype=torch.float32, device=dev)
        self.raw = torch.zeros(self.N, dtype=torch.float64, device=dev)
        self.shaped = torch.zeros(self.N, dtype=torch.float64, device=dev)
        self.done = torch.zeros(self.N, dtype=torch.uint8, device=dev)
        self._out = abi.Outputs()
        self._out.obs, self._out.raw = self.obs.data_ptr(), self.raw.data_ptr()
        self._out.shaped, self._out.done = self.shaped.data_ptr(), self.done.data_ptr()
        # first load is a verbatim copy of the snapshot (reset() keeps the live RNG tick)
        for k in abi.BUFFER_FIELDS:
            self.state[k].copy_(self.snapshot[k].expand_as(self.state[k]))

    # ------------------------------------------------------------------
    def _load(self, dst, src):
        for k in _STATE_KEYS:
            a = np.asarray(src[k])
            if k == "blocked":
                if a.shape[-1] != self.EW or a.dtype not in (np.uint32, np.int32):
                    a = abi.pack_blocked(a, self.EW)
                a = a.astype(np.uint32).view(np.int32)
            elif k == "ring":
                a = np.where(a < 0, 0xFFFF, a).astype(np.uint16).view(np.int16)
            elif k == "fenv":
                a = a.astype(np.float64)
            elif k == "ienv":
                a = a.astype(np.int32)
            else:
                if a.max(initial=0) > 255 or a.min(initial=0) < 0:
                    raise ValueError(f"{k} does not fit a byte plane")
                a = a.astype(np.uint8)
            dst[k].copy_(torch.from_numpy(np.ascontiguousarray(a)).reshape(dst[k].shape))
        dst["extra"].zero_()
        if "extra" in src and dst["extra"].numel():
            a = np.ascontiguousarray(np.asarray(src["extra"]).astype(np.uint32)).view(np.int32)
            dst["extra"].copy_(torch.from_numpy(a).reshape(dst["extra"].shape))

    def _derive(self, st):
        """Fill the library-maintained derived buffers (blocked_in) of a state dict."""
        _lib.check(self.lib.cygym_derive(self._h, C.byref(_buffers_struct(st)), self._stream()), self._h, "cygym_derive")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.cygym_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def set_config(self, cfg: abi.EnvConfig):
        """Attribute writes on the reference env object (env.base_line = ..., env.comp_scale = ...)."""
        self.cfg = cfg
        c = cfg.to_c()
        _lib.check(self.lib.cygym_set_config(self._h, C.byref(c)), self._h, "cygym_set_config")

    def load_state(self, state: dict):
        """Overwrite the live state (numpy planes, leading dim N or 1)."""
        lead = int(np.asarray(state["flags"]).shape[0])
        if lead == self.N:
            self._load(self.state, state)
            self._derive(self.state)
        else:
            tmp = _alloc_state(1, self.M, self.EW, self.device, self.K)
            self._load(tmp, state)
            self._derive(tmp)
            for k in abi.BUFFER_FIELDS:
                self.state[k].copy_(tmp[k].expand_as(self.state[k]))

    def reset(self, env_ids=None):
        """reset(from_init=True) (volt_typhoon_env.py:1904): restore the initial snapshot."""
        if env_ids is None:
            rc = self.lib.cygym_reset(self._h, None, None, self.N, self._stream())
        else:
            ids = torch.as_tensor(env_ids, dtype=torch.int32, device=self.device).contiguous()
            rc = self.lib.cygym_reset(self._h, None, C.c_void_p(ids.data_ptr()), int(ids.numel()), self._stream())
            torch.cuda.current_stream(self.device).synchronize()  # keep `ids` alive until consumed
        _lib.check(rc, self._h, "cygym_reset")

    def randomize(self, env_ids=None):
        """randomize_compromise_and_ownership() (volt_typhoon_env.py:330) for the given envs."""
        if env_ids is None:
            rc = self.lib.cygym_randomize(self._h, None, self.N, self._stream())
        else:
            ids = torch.as_tensor(env_ids, dtype=torch.int32, device=self.device).contiguous()
            rc = self.lib.cygym_randomize(self._h, C.c_void_p(ids.data_ptr()), int(ids.numel()), self._stream())
            torch.cuda.current_stream(self.device).synchronize()
        _lib.check(rc, self._h, "cygym_randomize")

    def actions_struct(self, act=None) -> abi.Actions:
        act = self.act if act is None else act
        a = abi.Actions()
        for k in ("mode", "n_groups", "atype", "n_exploit", "exploit", "app", "dev_cnt", "dev_idx"):
            t = act[k]
            if not t.is_contiguous() or t.device != self.device:
                raise ValueError(f"action tensor {k} must be contiguous on {self.device}")
            setattr(a, k, t.data_ptr())
        a.max_groups = int(act["atype"].shape[1]) if act["atype"].dim() > 1 else 1
        a.max_devs = int(act["dev_idx"].shape[1])
        return a

    def step(self, act=None):
        """One tick for every env.  `act`: dict of device tensors shaped like self.act (default: self.act).
        Returns (obs [N,M,6] f32, raw [N] f64, shaped [N] f64, done [N] u8) -- views of reused buffers."""
        a = self.actions_struct(act)
        _lib.check(self.lib.cygym_step(self._h, C.byref(a), C.byref(self._out), self._stream()), self._h, "cygym_step")
        return self.obs, self.raw, self.shaped, self.done

    def alloc_rollout(self, n_ticks: int):
        """Action and output tensors with a leading tick dimension for rollout()."""
        T = int(n_ticks)
        act = {k: torch.zeros((T,) + tuple(v.shape), dtype=v.dtype, device=self.device) for k, v in self.act.items()}
        act["exploit"].fill_(-1)
        act["app"].fill_(-1)
        out = dict(obs=torch.zeros((T, self.N, self.M, 6), dtype=torch.float32, device=self.device),
                   raw=torch.zeros((T, self.N), dtype=torch.float64, device=self.device),
                   shaped=torch.zeros((T, self.N), dtype=torch.float64, device=self.device),
                   done=torch.zeros((T, self.N), dtype=torch.uint8, device=self.device))
        return act, out

    def rollout(self, act: dict, out: dict):
        """T consecutive ticks in ONE launch (cygym_rollout): `act` / `out` carry a leading tick dimension
        (see alloc_rollout).  Open-loop: every tick's action is staged beforehand.  Same results as T step()
        calls; an env's state stays on chip between its ticks and envs never wait for each other."""
        T = int(act["mode"].shape[0])
        a = abi.Actions()
        for k in ("mode", "n_groups", "atype", "n_exploit", "exploit", "app", "dev_cnt", "dev_idx"):
            t = act[k]
            if not t.is_contiguous() or t.device != self.device or t.shape[0] != T or t.shape[1] != self.N:
                raise ValueError(f"rollout tensor {k} must be contiguous [T, N, ...] on {self.device}")
            setattr(a, k, t.data_ptr())
        a.max_groups = int(act["atype"].shape[2])
        a.max_devs = int(act["dev_idx"].shape[2])
        o = abi.Outputs()
        for k in ("obs", "raw", "shaped", "done"):
            t = out[k]
            if not t.is_contiguous() or t.shape[0] != T or t.shape[1] != self.N:
                raise ValueError(f"rollout output {k} must be contiguous [T, N, ...]")
            setattr(o, k, t.data_ptr())
        _lib.check(self.lib.cygym_rollout(self._h, T, C.byref(a), C.byref(o), self._stream()), self._h, "cygym_rollout")
        return out

    def gen_actions_rollout(self, tick0: int, act: dict):
        """Fill a [T, N, ...] action dict with the synthetic script for ticks tick0 .. tick0+T-1."""
        for t in range(act["mode"].shape[0]):
            self.gen_actions(tick0 + t, {k: v[t] for k, v in act.items()})
        return act

    def set_actions_numpy(self, act_np: dict):
        for k, t in self.act.items():
            a = np.asarray(act_np[k])
            t.copy_(torch.from_numpy(np.ascontiguousarray(a.astype(_np_dtype(t)))).reshape(t.shape))

    def gen_actions(self, tick: int, act=None):
        """Fill `act` (default self.act) with the synthetic bench script for `tick` (on device)."""
        act = self.act if act is None else act
        if act["atype"].shape[1] != 1:
            raise ValueError("the synthetic script is single-action (max_groups == 1)")
        p = lambda k: C.c_void_p(act[k].data_ptr())  # noqa: E731
        rc = self.lib.cygym_gen_actions(self._h, int(tick), p("mode"), p("n_groups"), p("atype"), p("n_exploit"),
                                        p("exploit"), p("app"), p("dev_cnt"), p("dev_idx"),
                                        int(act["dev_idx"].shape[1]), self._stream())
        _lib.check(rc, self._h, "cygym_gen_actions")
        return act

    def observe(self, role: int) -> torch.Tensor:
        """role 0: _get_state, 1: _get_defender_state, 2: _get_attacker_state (CyberDefenseEnv.py:146-257)."""
        width = 6 * self.M if role in (0, 1) else 4 * self.M + self.cfg.max_exploits
        out = torch.empty((self.N, width), dtype=torch.float32, device=self.device)
        _lib.check(self.lib.cygym_observe(self._h, int(role), C.c_void_p(out.data_ptr()), self._stream()),
                   self._h, "cygym_observe")
        return out

    def timer_start(self):
        _lib.check(self.lib.cygym_timer_start(self._h, self._stream()), self._h, "cygym_timer_start")

    def timer_stop(self) -> float:
        ms = C.c_float()
        _lib.check(self.lib.cygym_timer_stop(self._h, self._stream(), C.byref(ms)), self._h, "cygym_timer_stop")
        return float(ms.value)

    # ------------------------------------------------------------------
    def state_numpy(self) -> dict:
        torch.cuda.synchronize(self.device)
        out = {}
        for k in abi.BUFFER_FIELDS:
            a = self.state[k].cpu().numpy()
            if k in _NP_VIEW:
                a = a.view(_NP_VIEW[k])
            out[k] = a
        for i, k in enumerate(abi.LIVE_PLANES):
            out[k] = out["live"][:, i]
        for i, k in enumerate(abi.STASH_PLANES):
            out[k] = out["stash"][:, i]
        return out

    def counters(self) -> dict:
        """The cumulative `info` counters of the reference (volt_typhoon_env.py:1272-1285) as [N] tensors."""
        ie, fe = self.state["ienv"], self.state["fenv"]
        return {
            "step_count": ie[:, S.I_STEP_NUM], "revert_count": ie[:, S.I_REVERT_CNT],
            "checkpoint_count": ie[:, S.I_CKPT_CNT], "defensive_cost": fe[:, S.D_DEF_COST],
            "clearning_cost": fe[:, S.D_CLEAN_COST], "Scan_count": ie[:, S.I_SCAN_CNT],
            "work_done": ie[:, S.I_WORK_DONE], "Compromised_devices": ie[:, S.I_COMP_CNT],
            "Edges Blocked": ie[:, S.I_EDGES_BLOCKED], "Edges Added": ie[:, S.I_EDGES_ADDED],
        }


def _np_dtype(t: torch.Tensor):
    return {torch.int32: np.int32, torch.int16: np.int16, torch.uint8: np.uint8,
            torch.float32: np.float32, torch.float64: np.float64}[t.dtype]
