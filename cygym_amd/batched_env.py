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
               "ienv": torch.int32, "fenv": torch.float64, "extra": torch.int32, "forest": torch.int32, "hist": torch.int16,
               "anomaly": torch.float32}
_STATE_KEYS = abi.STATE_PLANES + ("blocked", "ring", "ienv", "fenv")
_NP_VIEW = {"blocked": np.uint32, "ring": np.uint16, "extra": np.uint32, "forest": np.uint32, "hist": np.uint16}


def _alloc_state(n, M, EW, device, K=0, detector=False, anomaly=False):
    """`live` / `stash` are the [N][4][M] buffers of the ABI; flags/busy/... are VIEWS into them.
    `extra` is the per-env list of edges evolve_network added (K = topo.max_extra entries); `forest` / `hist`
    (trained-detector mode: the env's flattened isolation forest and the comm-log history it is fitted on) have
    zero width unless asked for."""
    dims = {"live": (4, M), "stash": (4, M), "blocked": (EW,), "blocked_in": (EW,), "ring": (S.LOG_RING, 2),
            "ienv": (S.I_COUNT,), "fenv": (S.D_COUNT,), "extra": (abi.x_words(K),),
            "forest": (S.FOREST_WORDS if detector else 0,), "hist": (S.HIST_RING if detector else 0, 2),
            "anomaly": (M if anomaly else 0,)}   # per-env Device.anomaly_score: only the per-log scan path (fast_scan=False) writes it
    st = {k: torch.zeros((n,) + dims[k], dtype=dt, device=device) for k, dt in _BUF_DTYPES.items()}
    st["hist"].fill_(-1)
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
                 device="cuda:0", max_groups: int = 1, max_devs: int | None = None, detector: bool = False):
        self.lib = _lib.load()   # raises when libcygym_hip.so is missing: no fallback
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.CygymError("BatchedCyberDefenseEnv runs on a ROCm GPU only (device must be cuda:N)")
        self.topo = topo.normalised()
        # detector=True: trained-detector mode is available (defender action 10 -> service_detectors()): binds the
        # per-env forest (4 KB) and history ring (8 KB) and selects the full-feature kernels
        self.slow_scan = not cfg.fast_scan      # the per-log scan path (volt_typhoon_env.py:1030-1050) reads the long history
        self.detector = bool(detector) or self.slow_scan   # ... and writes per-env anomaly scores: both buffers come with it
        if self.detector and self.topo.det_apl is None:
            from . import detector as D
            self.topo.det_apl = D.apl_table()
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
        self._epoch = 0          # bumped by every launch through this object (_stream): per-env views cache their counter rows against it
        self._dirty_views = []   # views holding counter writes that have not reached the device yet (env_view.py)
        self.state = _alloc_state(self.N, self.M, self.EW, self.device, self.K, self.detector, self.slow_scan)
        self._scratch = None   # cygym_randomize's shuffle keys, allocated on first use
        self._act_cache = {}   # id(action dict) -> (data pointers, shapes, validated C struct)
        lead = int(np.asarray(init_state["flags"]).shape[0])
        if lead not in (1, self.N):
            raise ValueError("init_state must have leading dimension 1 or n_envs")
        self.snapshot = _alloc_state(lead, self.M, self.EW, self.device, self.K, self.detector, self.slow_scan)
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
        # one status word per batch: the kernels OR in the sticky / pending bits of the envs they tick
        # (cygym_outputs.status); take_status() reads and clears it
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.ret = self.alive = None   # episode-return accumulators of step(returns=True), see reset_returns()
        self.role_obs = {}     # "defender" / "attacker" -> [N, W] float32 role view written by step(view=...)
        self._outs = {}        # (view, full_obs) -> abi.Outputs
        self._out = self._outputs(None, True)
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
        if dst["anomaly"].numel():
            a = np.asarray(src["anomaly"], np.float32) if "anomaly" in src else self.topo.anomaly[None]
            dst["anomaly"].copy_(torch.from_numpy(np.ascontiguousarray(a)).reshape((-1, self.M)).expand_as(dst["anomaly"]))
        dst["forest"].zero_()
        dst["hist"].fill_(-1)
        for k, udt, sdt in (("forest", np.uint32, np.int32), ("hist", np.uint16, np.int16)):
            if k in src and dst[k].numel():
                a = np.ascontiguousarray(np.asarray(src[k]).astype(udt)).view(sdt)
                dst[k].copy_(torch.from_numpy(a).reshape((-1,) + tuple(dst[k].shape[1:])).expand_as(dst[k]))

    def _derive(self, st):
        """Fill the library-maintained derived buffers (blocked_in) of a state dict."""
        _lib.check(self.lib.cygym_derive(self._h, C.byref(_buffers_struct(st)), self._stream()), self._h, "cygym_derive")

    def _stream(self):
        """The stream argument of a library call.  Every launch passes through here, so this is also where the per-env
        views' pending counter writes are uploaded (before the launch) and their cached counter rows expire."""
        if self._dirty_views:
            self._flush_views()
        self._epoch += 1
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # `state` is what every consumer (and foreign code) reads the device tensors through: a view's pending counter writes
    # (env_view.py: `env.step_num = 0` ... held back so that a run of them is ONE upload) are flushed on access.
    @property
    def state(self):
        if self._dirty_views:
            self._flush_views()
        self._epoch += 1   # the caller may write through what it gets: cached counter rows expire
        return self._state

    @state.setter
    def state(self, st):
        self._state = st

    def _flush_views(self):
        views, self._dirty_views = self._dirty_views, []
        for v in views:
            v._flush()

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
            tmp = _alloc_state(1, self.M, self.EW, self.device, self.K, self.detector, self.slow_scan)
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
        if self._scratch is None:   # caller-owned scratch of cygym_randomize: u32 [N][ceil(M/64)*64]
            self._scratch = torch.empty((self.N, (self.M + 63) // 64 * 64), dtype=torch.int32, device=self.device)
        sc = C.c_void_p(self._scratch.data_ptr())
        if env_ids is None:
            rc = self.lib.cygym_randomize(self._h, None, self.N, sc, self._stream())
        else:
            ids = torch.as_tensor(env_ids, dtype=torch.int32, device=self.device).contiguous()
            if ids.numel() > self.N:
                raise ValueError("more env ids than envs")
            rc = self.lib.cygym_randomize(self._h, C.c_void_p(ids.data_ptr()), int(ids.numel()), sc, self._stream())
            torch.cuda.current_stream(self.device).synchronize()
        _lib.check(rc, self._h, "cygym_randomize")

    _ACT_DTYPES = {"mode": torch.int32, "n_groups": torch.int32, "atype": torch.int32, "n_exploit": torch.int32,
                   "exploit": torch.int32, "app": torch.int32, "dev_cnt": torch.int32, "dev_idx": torch.int16}

    def _check_actions(self, act, lead):
        """The kernel indexes these tensors by raw pointer: dtype, device, contiguity and every dimension are
        checked here, so a malformed dict is a Python error and never an out-of-bounds access on the GPU.
        `lead`: leading dimensions, (N,) for step() or (T, N) for rollout().  Returns (G, L)."""
        nl = len(lead)
        G = int(act["atype"].shape[nl]) if act["atype"].dim() > nl else 0
        L = int(act["dev_idx"].shape[nl]) if act["dev_idx"].dim() > nl else 0
        want = {"mode": (), "n_groups": (), "atype": (G,), "n_exploit": (G,), "exploit": (G, S.MAX_EXPLOITS),
                "app": (G,), "dev_cnt": (G,), "dev_idx": (L,)}
        for k, tail in want.items():
            t = act[k]
            if t.dtype != self._ACT_DTYPES[k] or t.device != self.device or not t.is_contiguous():
                raise ValueError(f"action tensor {k} must be a contiguous {self._ACT_DTYPES[k]} tensor on {self.device}")
            if tuple(t.shape) != tuple(lead) + tail:
                raise ValueError(f"action tensor {k} has shape {tuple(t.shape)}, expected {tuple(lead) + tail}")
        if G < 1 or L < 1:
            raise ValueError("action tensors need max_groups >= 1 and max_devs >= 1")
        return G, L

    def actions_struct(self, act=None) -> abi.Actions:
        """The C struct of an action dict.  Validation (dtype / shape / device of eight tensors) costs ~8 us of host
        time -- as much as the launch itself -- so the struct of a dict that was validated before is reused as long
        as the dict still holds the very same storages (a per-tick loop steps the same few dicts over and over)."""
        act = self.act if act is None else act
        ptrs = tuple(act[k].data_ptr() for k in self._ACT_DTYPES)
        sig = tuple((act[k].shape, act[k].dtype) for k in self._ACT_DTYPES)
        hit = self._act_cache.get(id(act))
        if hit is not None and hit[0] == ptrs and hit[1] == sig:
            return hit[2]
        G, L = self._check_actions(act, (self.N,))
        a = abi.Actions()
        for k, p in zip(self._ACT_DTYPES, ptrs):
            setattr(a, k, p)
        a.max_groups, a.max_devs = G, L
        if len(self._act_cache) > 4096:
            self._act_cache.clear()
        self._act_cache[id(act)] = (ptrs, sig, a)
        return a

    def role_width(self, role: str) -> int:
        if role == "defender":
            return 6 * self.M
        if role == "attacker":
            return 4 * self.M + self.cfg.max_exploits
        raise ValueError("role must be 'attacker' or 'defender'")

    def _outputs(self, view, full_obs, returns=False) -> abi.Outputs:
        """The cygym_outputs struct for one (role view, full observation, return accumulation) combination; cached."""
        key = (view, bool(full_obs), bool(returns))
        o = self._outs.get(key)
        if o is None:
            o = abi.Outputs()
            if returns:
                if self.ret is None:
                    self.reset_returns()
                o.ret, o.alive = self.ret.data_ptr(), self.alive.data_ptr()
            o.obs = self.obs.data_ptr() if full_obs else None
            o.raw, o.shaped, o.done = self.raw.data_ptr(), self.shaped.data_ptr(), self.done.data_ptr()
            o.status = self.status.data_ptr()
            if view is not None:
                if view not in self.role_obs:
                    self.role_obs[view] = torch.zeros((self.N, self.role_width(view)), dtype=torch.float32, device=self.device)
                setattr(o, "obs_def" if view == "defender" else "obs_att", self.role_obs[view].data_ptr())
            self._outs[key] = o
        return o

    def reset_returns(self):
        """Start a new rollout of every env: zero the episode-return accumulators ([N, 2] f64: defender, attacker reward
        sums) and mark every env alive (step(returns=True) adds to them until the env's first done, do_agent.py:266-274)."""
        if self.ret is None:
            self.ret = torch.zeros((self.N, 2), dtype=torch.float64, device=self.device)
            self.alive = torch.ones(self.N, dtype=torch.uint8, device=self.device)
        else:
            self.ret.zero_()
            self.alive.fill_(1)

    def step(self, act=None, view: str | None = None, full_obs: bool = True, returns: bool = False):
        """One tick for every env.  `act`: dict of device tensors shaped like self.act (default: self.act).
        Returns (obs [N,M,6] f32, raw [N] f64, shaped [N] f64, done [N] u8) -- views of reused buffers.

        view = "defender" / "attacker": the tick also writes that role's view of the state it leaves behind into
        self.role_obs[view] ([N, 6M] / [N, 4M + MaxExploits]) -- what `_get_defender_state()` / `_get_attacker_state()`
        return before the role's next action -- so a closed loop needs no observe() launch between ticks.
        full_obs=False: the [N, M, 6] full observation is not written (`obs` then holds an older tick's).
        returns=True: the tick also adds its raw reward to self.ret[:, role] for every env that has not reported done
        since reset_returns() (the `def_total` / `att_total` sums of the reference's loop), in the kernel."""
        a = self.actions_struct(act)
        o = self._out if (view is None and full_obs and not returns) else self._outputs(view, full_obs, returns)
        _lib.check(self.lib.cygym_step(self._h, C.byref(a), C.byref(o), self._stream()), self._h, "cygym_step")
        return self.obs, self.raw, self.shaped, self.done

    def write_actions(self, rows, a: dict, act=None):
        """Scatter one strategy's chosen actions into rows `rows` (int env ids, device tensor; None = all rows in order)
        of the action tensors `act` (default self.act), group 0: ONE launch (cygym_write_actions).  `a`: device
        tensors atype [n], exploit [n] (one index, -1 = none), app [n], and either dev_mask [n, M] (bool / uint8,
        compacted in the kernel to the ascending id list, first max_devs) or dev_idx [n, L] + dev_cnt [n]."""
        act = self.act if act is None else act
        dst = self.actions_struct(act)
        i32 = lambda t: t if (t.dtype == torch.int32 and t.is_contiguous()) else t.to(torch.int32).contiguous()  # noqa: E731
        keep = [i32(a["atype"]), i32(a["exploit"]), i32(a["app"])]
        n = int(keep[0].shape[0])
        src = abi.ActionRows()
        src.atype, src.exploit, src.app = (t.data_ptr() for t in keep)
        if rows is not None:
            r = i32(rows)
            if int(r.shape[0]) != n:
                raise ValueError("rows and action tensors differ in length")
            keep.append(r)
            src.rows = r.data_ptr()
        elif n > self.N:
            raise ValueError("more action rows than envs")
        if "dev_mask" in a:
            m = a["dev_mask"]
            if m.dtype not in (torch.bool, torch.uint8):
                m = m != 0
            m = m.contiguous()
            if tuple(m.shape) != (n, self.M):
                raise ValueError(f"dev_mask must have shape {(n, self.M)}")
            keep.append(m)
            src.dev_mask = m.data_ptr()
        else:
            di = a["dev_idx"]
            di = di if (di.dtype == torch.int16 and di.is_contiguous()) else di.to(torch.int16).contiguous()
            if tuple(di.shape) != (n, dst.max_devs):
                raise ValueError(f"dev_idx must have shape {(n, dst.max_devs)}")
            dc = i32(a["dev_cnt"])
            keep += [di, dc]
            src.dev_idx, src.dev_cnt = di.data_ptr(), dc.data_ptr()
        for t in keep:
            if t.device != self.device:
                raise ValueError("action rows must live on the batch's device")
        src.n = n
        _lib.check(self.lib.cygym_write_actions(self._h, C.byref(src), C.byref(dst), self._stream()), self._h, "cygym_write_actions")
        # (`keep` may die here: the caching allocator only reuses the blocks for work enqueued later on this stream)

    def decode_actions(self, rows, vec: torch.Tensor, n_types: int, n_exploits: int | None = None, n_apps: int = 0,
                       type_map: torch.Tensor | None = None, act=None, epsilon: float = 0.0):
        """DoubleOracle.decode_action (do_agent.py:935-998, plain branch) for a batch, fused with the scatter into the
        action tensors: `vec` [n, >= n_types + M + n_exploits + n_apps] float32 actor outputs (type logits | device
        values | exploit values | app values) -> rows `rows` of `act` (group 0): atype = argmax (through `type_map`
        [n_types] int32 when given), device list = ascending ids with value > 0, exploit = [argmax], app = argmax.
        ONE launch (cygym_decode_actions).  A row that chooses more devices than max_devs holds is cut to the first
        max_devs ids and raises abi.DECODE_TRUNCATED in the batch's status word.  epsilon > 0: with that probability
        an env's action type is uniformly random instead (the reference's epsilon-greedy, do_agent.py:972-973; the draw
        is addressed by the env's current rng tick, site CG_SITE_EPS_TYPE)."""
        act = self.act if act is None else act
        dst = self.actions_struct(act)
        if vec.dtype != torch.float32 or vec.dim() != 2 or vec.device != self.device or vec.stride(1) != 1:
            raise ValueError("vec must be a [n, width] float32 tensor on the batch's device with unit inner stride")
        n_exploits = self.cfg.max_exploits if n_exploits is None else int(n_exploits)
        n = int(vec.shape[0])
        if int(vec.shape[1]) < n_types + self.M + n_exploits + n_apps:
            raise ValueError("vec rows are narrower than n_types + n_devices + n_exploits + n_apps")
        src = abi.ActionVectors()
        src.vec, src.stride = vec.data_ptr(), int(vec.stride(0))
        src.n_types, src.n_devices, src.n_exploits, src.n_apps, src.n = int(n_types), self.M, n_exploits, int(n_apps), n
        src.status = self.status.data_ptr()
        if epsilon > 0.0:
            from . import rng as R
            src.epsilon_thr = R.bernoulli_threshold(float(epsilon))
        keep = [vec]
        if rows is not None:
            r = rows if (rows.dtype == torch.int32 and rows.is_contiguous()) else rows.to(torch.int32).contiguous()
            if int(r.shape[0]) != n or r.device != self.device:
                raise ValueError("rows must be a device tensor as long as vec")
            keep.append(r)
            src.rows = r.data_ptr()
        elif n > self.N:
            raise ValueError("more action rows than envs")
        if type_map is not None:
            tm = type_map if (type_map.dtype == torch.int32 and type_map.is_contiguous()) else type_map.to(torch.int32).contiguous()
            if int(tm.numel()) != int(n_types) or tm.device != self.device:
                raise ValueError("type_map must hold n_types int32 entries on the batch's device")
            keep.append(tm)
            src.type_map = tm.data_ptr()
        _lib.check(self.lib.cygym_decode_actions(self._h, C.byref(src), C.byref(dst), self._stream()), self._h, "cygym_decode_actions")

    def actor_head_decode(self, rows, hidden: torch.Tensor, weight_t: torch.Tensor, bias, n_types: int,
                          n_exploits: int | None = None, n_apps: int = 0, type_map=None, act=None, epsilon: float = 0.0,
                          tanh: bool = False, n_groups: int = 1):
        """The actor's LAST linear layer fused with decode_actions (cygym_actor_head_decode): action vector of row r =
        act(hidden[r] @ weight_t + bias), weight_t = head_weights(nn.Linear.weight) ([H, n_out rounded up to 64], k-major), decoded from registers
        -- the [n, n_out] vectors never reach HBM.  Limits: H <= 256, n_out = n_types + M + n_exploits + n_apps <= 512.
        n_groups = S > 1: a population of S same-shaped actors in one launch -- row r is multiplied with the matrix of actor
        r // (n / S); weight_t [S, H, pitch], bias [S, n_out], n / S a multiple of 16."""
        act = self.act if act is None else act
        dst = self.actions_struct(act)
        n_exploits = self.cfg.max_exploits if n_exploits is None else int(n_exploits)
        n_out = int(n_types) + self.M + n_exploits + int(n_apps)
        H = int(hidden.shape[1])
        ok = lambda t: t.dtype == torch.float32 and t.device == self.device  # noqa: E731
        if not ok(hidden) or hidden.dim() != 2 or hidden.stride(1) != 1:
            raise ValueError("hidden must be a [n, H] float32 tensor on the batch's device with unit inner stride")
        pitch = (n_out + 63) // 64 * 64
        S_ = int(n_groups)
        wshape, bshape = ((H, pitch), (n_out,)) if S_ <= 1 else ((S_, H, pitch), (S_, n_out))
        if not ok(weight_t) or tuple(weight_t.shape) != wshape or not weight_t.is_contiguous():
            raise ValueError(f"weight_t must be a contiguous float32 {list(wshape)} tensor (see head_weights())")
        if bias is not None and (not ok(bias) or tuple(bias.shape) != bshape or not bias.is_contiguous()):
            raise ValueError(f"bias must be a contiguous float32 {list(bshape)} tensor")
        n = int(hidden.shape[0])
        if S_ > 1 and (n % S_ or (n // S_) % 16):
            raise ValueError("a population launch needs the same number of rows per actor, a multiple of 16")
        hd = abi.ActorHead()
        hd.hidden, hd.weight_t, hd.bias = hidden.data_ptr(), weight_t.data_ptr(), (bias.data_ptr() if bias is not None else None)
        hd.H, hd.hidden_stride, hd.tanh_out, hd.weight_pitch = H, int(hidden.stride(0)), int(bool(tanh)), pitch
        hd.n_groups, hd.rows_per_group = (S_, n // S_) if S_ > 1 else (1, 0)
        src = abi.ActionVectors()
        src.n_types, src.n_devices, src.n_exploits, src.n_apps, src.n = int(n_types), self.M, n_exploits, int(n_apps), n
        src.status = self.status.data_ptr()
        if epsilon > 0.0:
            from . import rng as R
            src.epsilon_thr = R.bernoulli_threshold(float(epsilon))
        keep = [hidden]
        if rows is not None:
            r = rows if (rows.dtype == torch.int32 and rows.is_contiguous()) else rows.to(torch.int32).contiguous()
            if int(r.shape[0]) != n or r.device != self.device:
                raise ValueError("rows must be a device tensor as long as hidden")
            keep.append(r)
            src.rows = r.data_ptr()
        elif n > self.N:
            raise ValueError("more action rows than envs")
        if type_map is not None:
            tm = type_map if (type_map.dtype == torch.int32 and type_map.is_contiguous()) else type_map.to(torch.int32).contiguous()
            if int(tm.numel()) != int(n_types) or tm.device != self.device:
                raise ValueError("type_map must hold n_types int32 entries on the batch's device")
            keep.append(tm)
            src.type_map = tm.data_ptr()
        _lib.check(self.lib.cygym_actor_head_decode(self._h, C.byref(hd), C.byref(src), C.byref(dst), self._stream()),
                   self._h, "cygym_actor_head_decode")

    @staticmethod
    def head_weights(weight: torch.Tensor) -> torch.Tensor:
        """nn.Linear.weight [n_out, H] -> the k-major, row-padded copy actor_head_decode reads: [H, n_out rounded up to 64]."""
        n_out, H = weight.shape
        out = torch.zeros((H, (n_out + 63) // 64 * 64), dtype=torch.float32, device=weight.device)
        out[:, :n_out] = weight.detach().t()
        return out

    @staticmethod
    def pack_linear(weight: torch.Tensor, pad_out_to: int = 16) -> torch.Tensor:
        """nn.Linear.weight [N, K] -> the fragment-ordered copy cygym_actor_mlp_decode reads (cygym_abi.h):
        [ceil(N / 16)][ceil(K / 16)][64][4] with packed[t][g][lane][i] = W[16 t + lane % 16][16 g + 4 (lane // 16) + i],
        zeros outside; N rounded up to a multiple of `pad_out_to` (64 for the last layer)."""
        N, K = weight.shape
        Np, Kp = (N + pad_out_to - 1) // pad_out_to * pad_out_to, (K + 15) // 16 * 16
        w = torch.zeros((Np, Kp), dtype=torch.float32, device=weight.device)
        w[:N, :K] = weight.detach()
        # [t, c, g, kk, i] -> [t, g, kk, c, i]  (lane = 16 kk + c)
        return w.reshape(Np // 16, 16, Kp // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)

    def actor_mlp_decode(self, rows, obs: torch.Tensor, hidden_layers, head, n_types: int, n_exploits: int | None = None,
                         n_apps: int = 0, type_map=None, act=None, epsilon: float = 0.0, tanh: bool = False, n_groups: int = 1,
                         obs_by_env: bool = False, obs_role: str | None = None, rows_per_group: int | None = None, step: dict | None = None):
        """The WHOLE actor (Linear-ReLU stack + last Linear layer, do_agent.py:357-370) fused with decode_actions
        (cygym_actor_mlp_decode): ONE launch per acting role -- hidden activations and action vectors never reach HBM.
          obs            [n, K] float32 (unit inner stride); with obs_by_env the batch's [N, K] role view, read at rows `rows`
          hidden_layers  [(packed weights, bias, width), ...] 1 to 3 of them: pack_linear(nn.Linear.weight), widths multiples of
                         16 up to 256
          head           (pack_linear(last.weight, 64), bias): n_out = n_types + M + n_exploits + n_apps <= 8192
        n_groups = S > 1: a population of S same-shaped actors (packed tensors / biases of all actors concatenated, actor
        after actor), row r belongs to actor r // (n / S), n / S a multiple of 16.
        obs_role = "defender" / "attacker": `obs` is not read (pass None) -- the kernel builds the role's view of env rows[r]
        (or r) on chip from the batch's CURRENT state (flag plane + static columns: 256 bytes per env instead of a 6 KB view;
        _get_defender_state / _get_attacker_state, CyberDefenseEnv.py:194-257), so the tick need not write role views. M even.
        rows_per_group: with n_groups = S, row r belongs to actor (r // rows_per_group) % S (default n // S: actor after actor);
        the grid layouts of rollout_grid in env order are (nA * n_mc, nD) for the defender and (n_mc, nA) for the attacker.
        step = {"act": tensors of the tick to run FIRST, "view", "full_obs", "returns" as in step()}: cygym_step_actor -- the
        tick and this actor (on the state the tick leaves behind) as ONE launch; needs can_step_actor(...)."""
        act = self.act if act is None else act
        dst = self.actions_struct(act)
        n_exploits = self.cfg.max_exploits if n_exploits is None else int(n_exploits)
        n_out = int(n_types) + self.M + n_exploits + int(n_apps)
        ok = lambda t: t.dtype == torch.float32 and t.device == self.device  # noqa: E731
        if not 1 <= len(hidden_layers) <= abi.MLP_MAX_HIDDEN:
            raise ValueError(f"1 to {abi.MLP_MAX_HIDDEN} hidden layers")
        S_ = max(1, int(n_groups))
        ml = abi.ActorMlp()
        if obs_role is not None:
            if obs_role not in ("defender", "attacker") or self.M % 2:
                raise ValueError("obs_role is 'defender' or 'attacker', on batches with an even device count")
            K = self.role_width(obs_role)
            ml.obs, ml.obs_stride, ml.K, ml.obs_role = None, K, K, (1 if obs_role == "defender" else 2)
            obs_by_env = True
        else:
            if not ok(obs) or obs.dim() != 2 or obs.stride(1) != 1:
                raise ValueError("obs must be a [n, K] float32 tensor on the batch's device with unit inner stride")
            K = int(obs.shape[1])
            ml.obs, ml.obs_stride, ml.K = obs.data_ptr(), int(obs.stride(0)), K
        ml.n_hidden, ml.tanh_out, ml.obs_by_env = len(hidden_layers), int(bool(tanh)), int(bool(obs_by_env))
        kin = (K + 15) // 16
        for l, (w, b, width) in enumerate(hidden_layers):
            width = int(width)
            if width % 16 or not 16 <= width <= 256:
                raise ValueError("hidden widths must be multiples of 16 up to 256")
            if not ok(w) or not w.is_contiguous() or w.numel() != S_ * (width // 16) * kin * 256:
                raise ValueError(f"hidden layer {l}: packed weights have the wrong size (see pack_linear)")
            if b is not None and (not ok(b) or not b.is_contiguous() or b.numel() != S_ * width):
                raise ValueError(f"hidden layer {l}: bias must hold {S_ * width} float32 values")
            ml.w[l], ml.b[l], ml.width[l] = w.data_ptr(), (b.data_ptr() if b is not None else None), width
            kin = width // 16
        wh, bh = head
        n_out_p = (n_out + 63) // 64 * 64
        if not ok(wh) or not wh.is_contiguous() or wh.numel() != S_ * (n_out_p // 16) * kin * 256:
            raise ValueError("head: packed weights have the wrong size (pack_linear(weight, 64))")
        if bh is not None and (not ok(bh) or not bh.is_contiguous() or bh.numel() != S_ * n_out):
            raise ValueError(f"head: bias must hold {S_ * n_out} float32 values")
        ml.w_head, ml.b_head = wh.data_ptr(), (bh.data_ptr() if bh is not None else None)
        n = int(rows.shape[0]) if (obs_by_env and rows is not None) else (self.N if obs_role is not None else int(obs.shape[0]))
        rpg = (n // S_ if rows_per_group is None else int(rows_per_group)) if S_ > 1 else 0
        if S_ > 1 and (rpg < 16 or rpg % 16 or (rows_per_group is None and n % S_)):
            raise ValueError("a population launch needs the same number of rows per actor, a multiple of 16")
        ml.n_groups, ml.rows_per_group = (S_, rpg) if S_ > 1 else (1, 0)
        src = abi.ActionVectors()
        src.n_types, src.n_devices, src.n_exploits, src.n_apps, src.n = int(n_types), self.M, n_exploits, int(n_apps), n
        src.status = self.status.data_ptr()
        if epsilon > 0.0:
            from . import rng as R
            src.epsilon_thr = R.bernoulli_threshold(float(epsilon))
        keep = [obs]
        if rows is not None:
            r = rows if (rows.dtype == torch.int32 and rows.is_contiguous()) else rows.to(torch.int32).contiguous()
            if int(r.shape[0]) != n or r.device != self.device:
                raise ValueError("rows must be a device tensor with one entry per source row")
            keep.append(r)
            src.rows = r.data_ptr()
        elif n > self.N:
            raise ValueError("more action rows than envs")
        if obs_role is None and obs_by_env and int(obs.shape[0]) < self.N:
            raise ValueError("obs_by_env needs the batch's [N, K] role view")
        if type_map is not None:
            tm = type_map if (type_map.dtype == torch.int32 and type_map.is_contiguous()) else type_map.to(torch.int32).contiguous()
            if int(tm.numel()) != int(n_types) or tm.device != self.device:
                raise ValueError("type_map must hold n_types int32 entries on the batch's device")
            keep.append(tm)
            src.type_map = tm.data_ptr()
        if step is not None:
            a = self.actions_struct(step.get("act"))
            view, full_obs, returns = step.get("view"), bool(step.get("full_obs", False)), bool(step.get("returns", False))
            o = self._out if (view is None and full_obs and not returns) else self._outputs(view, full_obs, returns)
            _lib.check(self.lib.cygym_step_actor(self._h, C.byref(a), C.byref(o), C.byref(ml), C.byref(src), C.byref(dst), self._stream()),
                       self._h, "cygym_step_actor")
            return
        _lib.check(self.lib.cygym_actor_mlp_decode(self._h, C.byref(ml), C.byref(src), C.byref(dst), self._stream()),
                   self._h, "cygym_actor_mlp_decode")

    def can_step_actor(self, n_out: int) -> bool:
        """May a tick and the next actor run as ONE launch (cygym_step_actor)?  Where both kernels share their launch shape: 256
        devices, a fixed topology without detector buffers, a multiple of 16 envs and at most 16 envs per CU, 257..384 outputs."""
        cus = torch.cuda.get_device_properties(self.device).multi_processor_count
        return (self.M == 256 and not getattr(self, "detector", False) and not self.slow_scan and int(self.topo.max_extra) == 0
                and self.N % 16 == 0 and self.N <= 16 * cus and 257 <= int(n_out) <= 384)

    def group_actions(self, rows, types: torch.Tensor, exploit=None, app=None, role: str = "defender", n_types: int | None = None,
                      noop: int | None = None, single_types=(11, 12), visible: torch.Tensor | None = None, act=None):
        """The grouping of per-device decisions into `env.step(groups)` (IPPO.py:560-572 / MAPPO.py) for a batch, ONE launch
        (cygym_group_actions): `types` [n, M] (any integer dtype) = the action type every device sampled; for each type in
        ascending order except `noop` (default: 8 defender / 3 attacker) the visible devices that sampled it become the group
        (type, [exploit[r]], ascending ids, app[r]) -- one uniformly random device for a type in `single_types` (the Philox draw
        addressed by the env's rng tick, site CG_SITE_GROUP_PICK) -- and a row without groups steps [(noop, [0], [], 0)].
        `visible` [n, M] overrides the role's visibility mask (build_visibility_mask, IPPO.py:74-96), which the kernel
        otherwise reads off the flag plane.  Writes n_groups and the groups of rows `rows` of `act`; needs max_groups >=
        the number of groups a row can have and max_devs >= M (else the row is cut and abi.DECODE_TRUNCATED raised)."""
        act = self.act if act is None else act
        dst = self.actions_struct(act)
        if role not in ("defender", "attacker"):
            raise ValueError("role must be 'attacker' or 'defender'")
        n_types = (14 if role == "defender" else 3) if n_types is None else int(n_types)   # get_num_action_types (volt_typhoon_env.py:514-520): the attacker's no-op (3) lies outside its range
        noop = (8 if role == "defender" else 3) if noop is None else int(noop)
        if types.dim() != 2 or int(types.shape[1]) != self.M or types.device != self.device:
            raise ValueError("types must be an [n, M] integer tensor on the batch's device")
        t8 = types if (types.dtype == torch.uint8 and types.is_contiguous()) else types.to(torch.uint8).contiguous()
        n = int(t8.shape[0])
        src = abi.DeviceTypes()
        src.types, src.n, src.n_types, src.noop, src.role = t8.data_ptr(), n, n_types, noop, (1 if role == "defender" else 2)
        src.single_mask = sum(1 << int(t) for t in single_types if 0 <= int(t) < 32)
        src.status = self.status.data_ptr()
        keep = [t8]

        def i32(x, what):
            x = x if (x.dtype == torch.int32 and x.is_contiguous()) else x.to(torch.int32).contiguous()
            if int(x.numel()) != n or x.device != self.device:
                raise ValueError(f"{what} must hold one entry per row on the batch's device")
            keep.append(x)
            return x.data_ptr()

        if rows is not None:
            src.rows = i32(rows, "rows")
        elif n > self.N:
            raise ValueError("more rows than envs")
        if exploit is not None:
            src.exploit = i32(exploit, "exploit")
        if app is not None:
            src.app = i32(app, "app")
        if visible is not None:
            v8 = visible if (visible.dtype == torch.uint8 and visible.is_contiguous()) else (visible != 0).to(torch.uint8).contiguous()
            if tuple(v8.shape) != (n, self.M) or v8.device != self.device:
                raise ValueError("visible must be [n, M] on the batch's device")
            keep.append(v8)
            src.visible = v8.data_ptr()
        _lib.check(self.lib.cygym_group_actions(self._h, C.byref(src), C.byref(dst), self._stream()), self._h, "cygym_group_actions")

    def sample_group_actions(self, rows, logits: torch.Tensor, exp_logits=None, app_logits=None, role: str = "defender", noop: int | None = None,
                             single_types=(11, 12), greedy: bool = False, act=None):
        """Sampling AND grouping of a per-device actor's decisions in ONE launch (cygym_sample_group_actions; IPPO.py:524-572 for
        a batch): `logits` [n, M, K] float32 -> one Categorical sample per VISIBLE device (the role's mask, read off the flag
        plane), one for the exploit (`exp_logits` [n, E]) and one for the app (`app_logits` [n, A]), the sum of their
        log-probabilities, and the groups written into rows `rows` of `act` like group_actions.  Samples walk the inverse CDF
        with addressed Philox draws (env, rng tick, CG_SITE_SAMPLE, device / head); greedy=True takes the arg-max instead.
        Returns (types [n, M] uint8 -- 0 where invisible --, exploit [n] int32, app [n] int32, logp [n] float32)."""
        act = self.act if act is None else act
        dst = self.actions_struct(act)
        if role not in ("defender", "attacker"):
            raise ValueError("role must be 'attacker' or 'defender'")
        ok = lambda t: t.dtype == torch.float32 and t.device == self.device and t.is_contiguous()  # noqa: E731
        if not ok(logits) or logits.dim() != 3 or int(logits.shape[1]) != self.M or not 1 <= int(logits.shape[2]) <= 32:
            raise ValueError("logits must be a contiguous [n, M, K <= 32] float32 tensor on the batch's device")
        n, K = int(logits.shape[0]), int(logits.shape[2])
        src = abi.DeviceLogits()
        types = torch.empty((n, self.M), dtype=torch.uint8, device=self.device)
        exp_o = torch.empty((n,), dtype=torch.int32, device=self.device)
        app_o = torch.empty((n,), dtype=torch.int32, device=self.device)
        logp = torch.empty((n,), dtype=torch.float32, device=self.device)
        src.logits, src.types_out, src.exp_out, src.app_out, src.logp_out = logits.data_ptr(), types.data_ptr(), exp_o.data_ptr(), app_o.data_ptr(), logp.data_ptr()
        src.n, src.n_types, src.role, src.greedy = n, K, (1 if role == "defender" else 2), int(bool(greedy))
        src.noop = (8 if role == "defender" else 3) if noop is None else int(noop)
        src.single_mask = sum(1 << int(t) for t in single_types if 0 <= int(t) < 32)
        src.status = self.status.data_ptr()
        keep = [logits]
        for name, t in (("exp", exp_logits), ("app", app_logits)):
            if t is None or int(t.shape[-1]) == 0:
                continue
            if not ok(t) or t.dim() != 2 or int(t.shape[0]) != n or int(t.shape[1]) > 32:
                raise ValueError(f"{name}_logits must be a contiguous [n, <= 32] float32 tensor on the batch's device")
            setattr(src, name + "_logits", t.data_ptr())
            setattr(src, "n_" + name, int(t.shape[1]))
            keep.append(t)
        if rows is not None:
            r = rows if (rows.dtype == torch.int32 and rows.is_contiguous()) else rows.to(torch.int32).contiguous()
            if int(r.numel()) != n or r.device != self.device:
                raise ValueError("rows must hold one env id per row on the batch's device")
            keep.append(r)
            src.rows = r.data_ptr()
        elif n > self.N:
            raise ValueError("more rows than envs")
        _lib.check(self.lib.cygym_sample_group_actions(self._h, C.byref(src), C.byref(dst), self._stream()), self._h, "cygym_sample_group_actions")
        return types, exp_o, app_o, logp

    def take_status(self) -> int:
        """Read and clear the batch's status word: the OR of CG_E_TOPO_OVF | CG_E_BUSY_SAT | CG_E_DET_PENDING |
        CG_E_UNPINNED over the envs ticked since the last call (one 4-byte device-to-host copy; synchronises)."""
        v = int(self.status.item()) & 0xFFFFFFFF
        if v:
            self.status.zero_()
        return v

    def prime_view(self, role: str):
        """Fill self.role_obs[role] with the role's view of the CURRENT state (one cygym_observe launch): the first
        observation of a closed loop; every later one is written by step(view=...) itself."""
        self._outputs(role, False)
        self.role_obs[role].copy_(self.observe(1 if role == "defender" else 2))
        return self.role_obs[role]

    def step_range(self, begin: int, n: int, act=None, view: str | None = None, full_obs: bool = True, returns: bool = False):
        """One tick for the envs [begin, begin + n) only, on the current stream (cygym_step_range).  The action and
        output tensors keep their [N] leading dimension.  A closed-loop driver pipelines sub-batches this way: each
        sub-batch on its own stream, so that its policy evaluation and the tail of its slowest env overlap the other
        sub-batches' ticks (see bench.py, leg `per_tick_stepping`)."""
        a = self.actions_struct(act)
        o = self._out if (view is None and full_obs and not returns) else self._outputs(view, full_obs, returns)
        _lib.check(self.lib.cygym_step_range(self._h, int(begin), int(n), C.byref(a), C.byref(o), self._stream()),
                   self._h, "cygym_step_range")
        return self.obs, self.raw, self.shaped, self.done

    # ---- trained-detector mode: Detector.train is a host callback (cygym_amd/detector.py) ----
    def install_forest(self, env: int, words):
        """Write one env's flattened forest (keeping the request the tick recorded in header words 3, 4) and clear
        CG_E_DET_PENDING.  `words`: u32 [FOREST_WORDS] from detector.fit_forest / flatten_forest."""
        if not self.detector:
            raise _lib.CygymError("this batch was created without detector=True")
        w = torch.from_numpy(np.ascontiguousarray(np.asarray(words, np.uint32)).view(np.int32)).to(self.device)
        f = self.state["forest"][env]
        f[0:3] = w[0:3]
        f[5] = f[3]
        f[7] = w[7]
        f[S.FOREST_HDR:] = w[S.FOREST_HDR:]
        self.state["ienv"][env, S.I_FLAGS] &= ~S.E_DET_PENDING

    def pending_detectors(self, env_ids=None) -> torch.Tensor:
        """Env ids (int64 device tensor, ascending) whose Detector.train request is still unanswered
        (CG_E_DET_PENDING), optionally among `env_ids` only."""
        flags = self.state["ienv"][:, S.I_FLAGS]
        if env_ids is None:
            return torch.nonzero(flags & S.E_DET_PENDING).flatten()
        ids = torch.as_tensor(env_ids, dtype=torch.int64, device=self.device).flatten()
        if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= self.N):
            raise IndexError("env id out of range")
        return ids[(flags[ids] & S.E_DET_PENDING) != 0]

    def service_detectors(self, env_ids=None) -> int:
        """Answer every pending Detector.train (defender action 10 on a non-empty log, volt_typhoon_env.py:945-962):
        fit scikit-learn's IsolationForest on the last <= 2000 entries of the env's history ring -- the numpy stream
        it draws from seeded by the Philox draw addressed (env, request tick, CG_SITE_DET_FIT) -- and install the
        flattened trees.  Call it after a tick that may have carried action 10 and before the next scan (a scan
        that finds the request still pending answers all-"D" and raises CG_E_UNPINNED).  `env_ids`: only these envs
        (a per-env view services its own request, not the whole batch's).

        ONE gather of the pending envs' request headers, log totals and history rings (three device-to-host copies
        whatever the number of requests), the fits, ONE scatter of the forests back.  Synchronises; returns the
        number of forests fitted."""
        if not self.detector:
            return 0
        from . import detector as D
        pend = self.pending_detectors(env_ids)
        n = int(pend.numel())                      # (the one synchronisation of a call that finds nothing to do)
        if n == 0:
            return 0
        st = self.state
        ids = pend.cpu().numpy()
        hdr = st["forest"][pend, :S.FOREST_HDR].cpu().numpy().view(np.uint32)
        total = st["ienv"][pend, S.I_LOG_TOTAL].cpu().numpy().astype(np.int64)
        hist = st["hist"][pend].cpu().numpy().view(np.uint16)
        req_tick, req_total, n_fits = hdr[:, 3].astype(np.int64), hdr[:, 4].astype(np.int64), hdr[:, 6].astype(np.int64)
        gone = total - np.maximum(0, req_total - S.TRAIN_WINDOW) > S.HIST_RING
        if gone.any():
            e = int(ids[np.argmax(gone)])
            raise _lib.CygymError(f"env {e}: the training window of its request has left the history ring "
                                  "(service_detectors() must run before 48 more log entries arrive)")
        cfg = self.cfg
        words = D.fit_forests(
            [D.training_window(hist[j], int(req_total[j]), bool(cfg.turbo), cfg.turbo_train_max_logs, cfg.turbo_train_stride)
             for j in range(n)],
            [D.fit_seed(cfg.seed, cfg.env_id_base + int(ids[j]), int(req_tick[j])) for j in range(n)],
            [int(v) for v in n_fits])
        w = torch.from_numpy(np.ascontiguousarray(words).view(np.int32)).to(self.device)
        f = st["forest"]
        cur = f[pend]                                # [n, FOREST_WORDS]: keep the request the tick recorded (words 3, 4, 6)
        cur[:, 0:3] = w[:, 0:3]
        cur[:, 5] = cur[:, 3]
        cur[:, 7] = w[:, 7]
        cur[:, S.FOREST_HDR:] = w[:, S.FOREST_HDR:]
        f[pend] = cur
        st["ienv"][pend, S.I_FLAGS] &= ~S.E_DET_PENDING
        return n

    def unpinned_envs(self) -> int:
        """How many envs carry the sticky CG_E_UNPINNED bit: a scan ran in trained-detector mode without a current
        forest (action 10 never serviced, or no forest buffer bound), so their results are not the reference's."""
        return int(((self.state["ienv"][:, S.I_FLAGS] & S.E_UNPINNED) != 0).sum())

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

    def _training_ticks(self, act: dict) -> list:
        """Ticks of a [T, N, ...] script in which some env carries defender action 10 (Detector.train,
        volt_typhoon_env.py:945-962) in a group the tick will read.  One small reduction + one device-to-host copy."""
        G = act["atype"].shape[2]
        ng = act["n_groups"]
        used = torch.arange(G, device=self.device)[None, None, :] < ng.clamp(min=1)[:, :, None]   # step(): group 0 only
        hit = ((act["atype"] == 10) & used).any(dim=2) & ((act["mode"] & 0xFF) == S.MODE_DEFENDER) & (ng >= 0)
        return torch.nonzero(hit.any(dim=1)).flatten().tolist()

    def rollout(self, act: dict, out: dict, check: bool = True):
        """T consecutive ticks in ONE launch (cygym_rollout): `act` / `out` carry a leading tick dimension
        (see alloc_rollout).  Open-loop: every tick's action is staged beforehand.  Same results as T step()
        calls; an env's state stays on chip between its ticks and envs never wait for each other.

        `out` may also hold "obs_def" / "obs_att" ([T, N, 6M] / [T, N, 4M + MaxExploits] float32): the role views of
        the state each tick leaves behind (cygym_outputs); "obs" may be None (not written).

        Detector.train is a host callback, and the reference trains synchronously inside the tick (:961): on a batch
        created with detector=True a script that carries defender action 10 is therefore CUT after every such tick --
        launch, service_detectors(), next launch -- so that later scans see the forests the reference would have.
        On a batch without detector buffers nothing can be fitted: the launch runs, and if a scan then ran in
        trained mode without a forest (CG_E_UNPINNED) this raises instead of returning all-"D" results silently.
        check=False skips that final status read (the call then stays asynchronous; poll take_status() yourself)."""
        T = int(act["mode"].shape[0])
        G, L = self._check_actions(act, (T, self.N))
        odt = {"obs": (torch.float32, (self.M, 6)), "raw": (torch.float64, ()), "shaped": (torch.float64, ()), "done": (torch.uint8, ()),
               "obs_def": (torch.float32, (6 * self.M,)), "obs_att": (torch.float32, (4 * self.M + self.cfg.max_exploits,))}
        optional = ("obs", "obs_def", "obs_att")
        for k, (dt, tail) in odt.items():
            t = out.get(k)
            if t is None and k in optional:
                continue
            if t is None or t.dtype != dt or t.device != self.device or not t.is_contiguous() or tuple(t.shape) != (T, self.N) + tail:
                raise ValueError(f"rollout output {k} must be a contiguous {dt} tensor of shape {(T, self.N) + tail} on {self.device}")
        cuts = [t + 1 for t in self._training_ticks(act)] if self.detector else []
        bounds = sorted(set(c for c in cuts if c < T) | {T})
        t0 = 0
        for t1 in bounds:
            a = abi.Actions()
            for k in self._ACT_DTYPES:
                setattr(a, k, act[k][t0:t1].data_ptr())
            a.max_groups, a.max_devs = G, L
            o = abi.Outputs()
            for k in odt:
                if out.get(k) is not None:
                    setattr(o, k, out[k][t0:t1].data_ptr())
            o.status = self.status.data_ptr()
            _lib.check(self.lib.cygym_rollout(self._h, t1 - t0, C.byref(a), C.byref(o), self._stream()), self._h, "cygym_rollout")
            if t1 in cuts:
                self.service_detectors()
            t0 = t1
        if check and (self.take_status() & S.E_UNPINNED):
            raise _lib.CygymError(
                f"{self.unpinned_envs()} env(s) ran a scan in trained-detector mode without a current forest (defender action 10 "
                "earlier in the script): create the batch with detector=True so that the trainings can be serviced")
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

    def visibility_mask(self, role: str) -> torch.Tensor:
        """IPPO / MAPPO's `build_visibility_mask(env, role)` (IPPO.py:74-96) for every env at once, on the device:
        [N, M] float32, 1 where the device is visible to the role -- attacker: Known_to_attacker and attacker_owned and
        not Not_yet_added; defender: attacker_owned and not Not_yet_added.  Pure tensor ops on the flag plane (no
        launch of this library, no host round trip), for closed-loop policies that mask their per-device heads."""
        f = self.state["flags"]
        if role == "attacker":
            want = S.F_KNOWN | S.F_OWNED
        elif role == "defender":
            want = S.F_OWNED
        else:
            raise ValueError("role must be 'attacker' or 'defender'")
        return ((f & (want | S.F_NYA)) == want).to(torch.float32)

    def launch_plan(self) -> dict:
        """How the tick kernels of this batch are launched (cygym_launch_plan, include/cygym_abi.h)."""
        out = (C.c_int32 * 8)()
        _lib.check(self.lib.cygym_launch_plan(self._h, out), self._h, "cygym_launch_plan")
        keys = ("waves_per_workgroup", "waves_per_workgroup_rollout", "lds_bytes_per_wave", "lds_bytes_shared",
                "comp_by_in_global", "lists_in_global", "reserved", "wide")
        return dict(zip(keys, (int(v) for v in out)))

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
