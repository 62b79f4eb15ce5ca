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
               "ienv": torch.int32, "fenv": torch.float64, "extra": torch.int32, "forest": torch.int32, "hist": torch.int16}
_STATE_KEYS = abi.STATE_PLANES + ("blocked", "ring", "ienv", "fenv")
_NP_VIEW = {"blocked": np.uint32, "ring": np.uint16, "extra": np.uint32, "forest": np.uint32, "hist": np.uint16}


def _alloc_state(n, M, EW, device, K=0, detector=False):
    """`live` / `stash` are the [N][4][M] buffers of the ABI; flags/busy/... are VIEWS into them.
    `extra` is the per-env list of edges evolve_network added (K = topo.max_extra entries); `forest` / `hist`
    (trained-detector mode: the env's flattened isolation forest and the comm-log history it is fitted on) have
    zero width unless asked for."""
    dims = {"live": (4, M), "stash": (4, M), "blocked": (EW,), "blocked_in": (EW,), "ring": (S.LOG_RING, 2),
            "ienv": (S.I_COUNT,), "fenv": (S.D_COUNT,), "extra": (abi.x_words(K),),
            "forest": (S.FOREST_WORDS if detector else 0,), "hist": (S.HIST_RING if detector else 0, 2)}
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
        self.detector = bool(detector)
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
        self.state = _alloc_state(self.N, self.M, self.EW, self.device, self.K, self.detector)
        self._scratch = None   # cygym_randomize's shuffle keys, allocated on first use
        self._act_cache = {}   # id(action dict) -> (data pointers, shapes, validated C struct)
        lead = int(np.asarray(init_state["flags"]).shape[0])
        if lead not in (1, self.N):
            raise ValueError("init_state must have leading dimension 1 or n_envs")
        self.snapshot = _alloc_state(lead, self.M, self.EW, self.device, self.K, self.detector)
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
            tmp = _alloc_state(1, self.M, self.EW, self.device, self.K, self.detector)
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

    def step(self, act=None):
        """One tick for every env.  `act`: dict of device tensors shaped like self.act (default: self.act).
        Returns (obs [N,M,6] f32, raw [N] f64, shaped [N] f64, done [N] u8) -- views of reused buffers."""
        a = self.actions_struct(act)
        _lib.check(self.lib.cygym_step(self._h, C.byref(a), C.byref(self._out), self._stream()), self._h, "cygym_step")
        return self.obs, self.raw, self.shaped, self.done

    def step_range(self, begin: int, n: int, act=None):
        """One tick for the envs [begin, begin + n) only, on the current stream (cygym_step_range).  The action and
        output tensors keep their [N] leading dimension.  A closed-loop driver pipelines sub-batches this way: each
        sub-batch on its own stream, so that its policy evaluation and the tail of its slowest env overlap the other
        sub-batches' ticks (see bench.py, leg `per_tick_stepping`)."""
        a = self.actions_struct(act)
        _lib.check(self.lib.cygym_step_range(self._h, int(begin), int(n), C.byref(a), C.byref(self._out), self._stream()),
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
        f[S.FOREST_HDR:] = w[S.FOREST_HDR:]
        self.state["ienv"][env, S.I_FLAGS] &= ~S.E_DET_PENDING

    def service_detectors(self) -> int:
        """Answer every pending Detector.train (defender action 10 on a non-empty log, volt_typhoon_env.py:945-962):
        fit scikit-learn's IsolationForest on the last <= 2000 entries of the env's history ring -- the numpy stream
        it draws from seeded by the Philox draw addressed (env, request tick, CG_SITE_DET_FIT) -- and install the
        flattened trees.  Call it after a tick that may have carried action 10 and before the next scan (a scan
        that finds the request still pending answers all-"D" and raises CG_E_UNPINNED).  Synchronises; returns the
        number of forests fitted."""
        if not self.detector:
            return 0
        from . import detector as D
        flags = self.state["ienv"][:, S.I_FLAGS]
        pend = torch.nonzero(flags & S.E_DET_PENDING).flatten().tolist()
        for e in pend:
            hdr = self.state["forest"][e, :S.FOREST_HDR].cpu().numpy().view(np.uint32)
            req_tick, req_total = int(hdr[3]), int(hdr[4])
            total = int(self.state["ienv"][e, S.I_LOG_TOTAL])
            if total - max(0, req_total - S.TRAIN_WINDOW) > S.HIST_RING:
                raise _lib.CygymError(f"env {e}: the training window of tick {req_tick} has left the history ring "
                                      "(service_detectors() must run before 48 more log entries arrive)")
            hist = self.state["hist"][e].cpu().numpy().view(np.uint16)
            rows = D.training_window(hist, req_total, bool(self.cfg.turbo), self.cfg.turbo_train_max_logs, self.cfg.turbo_train_stride)
            self.install_forest(e, D.fit_forest(rows, D.fit_seed(self.cfg.seed, self.cfg.env_id_base + e, req_tick),
                                                n_fits=int(hdr[6])))
        return len(pend)

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
        G, L = self._check_actions(act, (T, self.N))
        a = abi.Actions()
        for k in self._ACT_DTYPES:
            setattr(a, k, act[k].data_ptr())
        a.max_groups, a.max_devs = G, L
        o = abi.Outputs()
        odt = {"obs": (torch.float32, (self.M, 6)), "raw": (torch.float64, ()), "shaped": (torch.float64, ()), "done": (torch.uint8, ())}
        for k, (dt, tail) in odt.items():
            t = out[k]
            if t.dtype != dt or t.device != self.device or not t.is_contiguous() or tuple(t.shape) != (T, self.N) + tail:
                raise ValueError(f"rollout output {k} must be a contiguous {dt} tensor of shape {(T, self.N) + tail} on {self.device}")
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
