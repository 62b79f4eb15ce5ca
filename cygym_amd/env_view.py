"""CyberDefenseEnvView: the reference's per-env Gym-style surface over one slot of a
BatchedCyberDefenseEnv, so the reference's rollout loops can drive the HIP kernels
unchanged.

Mirrors `Volt_Typhoon_CyberDefenseEnv` (volt_typhoon_env.py) / `CyberDefenseEnv`
(CyberDefenseEnv.py): same method names, argument meaning, return shapes and error
behaviour.  Every tick runs in the HIP kernel; this class only marshals.

    env = CyberDefenseEnvView(batch, index=0)
    env.mode = "defender"
    state, raw, shaped, done, info, logs = env.step((1, [0], [3, 7], 0))

Batch-wide vs per-env: scalar knobs (`comp_scale`, `work_scale`, ...) and the Philox seed live
in the handle's config and therefore apply to every env of the batch; counters, flags, `mode`
and `base_line` are per env -- `env.base_line = ...` (the reference's loops assign it per env and
turn, do_agent.py:218-221) travels in THIS env's mode word (CG_MODE_BASELINE), so two views of one
batch driven by two reference loops do not disturb each other.

Host traffic: the counters (`step_num`, `work_done`, ...) of an env are read as ONE row copy that stays
valid until the batch launches again, and a run of counter writes (the reference zeroes twelve of them
before a rollout, do_agent.py:192-196, each behind a `hasattr`) is held back and uploaded as one row
before the next launch or the next access to `batch.state`.
"""
from __future__ import annotations

import dataclasses
import random

import numpy as np
import torch

from . import abi, host_logic as HL
from . import spec as S


_NP_OF = {torch.int32: np.int32, torch.int16: np.int16, torch.uint8: np.uint8}


class Discrete:
    """Stand-in for gym.spaces.Discrete (n, sample, contains)."""

    def __init__(self, n):
        self.n = int(n)

    def sample(self):
        return random.randrange(self.n)

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return f"Discrete({self.n})"


class Box:
    def __init__(self, low, high, shape, dtype=np.float32):
        self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype


_COUNTER_COLS = {
    "step_num": S.I_STEP_NUM, "defender_step": S.I_DEF_STEP, "attacker_step": S.I_ATT_STEP,
    "work_done": S.I_WORK_DONE, "checkpoint_count": S.I_CKPT_CNT, "revert_count": S.I_REVERT_CNT,
    "scan_cnt": S.I_SCAN_CNT, "compromised_devices_cnt": S.I_COMP_CNT, "edges_blocked": S.I_EDGES_BLOCKED,
    "edges_added": S.I_EDGES_ADDED,
}
_FLOAT_COLS = {"defensive_cost": S.D_DEF_COST, "clearning_cost": S.D_CLEAN_COST}
_BASELINE_NAMES = {v: k for k, v in abi.BASELINES.items()}
_CONFIG_ATTRS = {"comp_scale", "work_scale", "def_scale", "lambda_events", "p_add", "p_attacker",
                 "workload_cap", "zero_day", "fast_scan", "scaling_vulnerability"}


class CyberDefenseEnvView:
    def __init__(self, batch, index: int = 0):
        object.__setattr__(self, "_b", batch)
        object.__setattr__(self, "_i", int(index))
        if not (0 <= self._i < batch.N):
            raise IndexError("env index out of range")
        object.__setattr__(self, "_base_line", None)    # None: the batch's configured baseline
        object.__setattr__(self, "_rows", None)         # (batch epoch, ienv row int64, fenv row float64) as last fetched
        object.__setattr__(self, "_pend_i", {})         # counter writes not uploaded yet: column -> value
        object.__setattr__(self, "_pend_f", {})
        self.mode = None                     # CyberDefenseEnv.py:31
        self.state = self._get_state()
        self.debug = False
        self.tech = "DO"
        self.snapshot_path = None
        self.private_exploit_id = None
        self.prior_pi = None
        self.time_budget_deadline = None
        self.time_budget_exceeded = None
        self.time_budget_seconds = 2.592e12

    # ---- attribute surface -------------------------------------------------
    def __getattr__(self, name):
        b = object.__getattribute__(self, "_b")
        i = object.__getattribute__(self, "_i")
        if name in _COUNTER_COLS:
            col = _COUNTER_COLS[name]
            pend = object.__getattribute__(self, "_pend_i")
            return int(pend[col]) if col in pend else int(self._counter_rows()[0][col])
        if name in _FLOAT_COLS:
            col = _FLOAT_COLS[name]
            pend = object.__getattribute__(self, "_pend_f")
            return float(pend[col]) if col in pend else float(self._counter_rows()[1][col])
        if name == "base_line":
            own = object.__getattribute__(self, "_base_line")
            if own is None:
                own = b.cfg.baseline
            return own if isinstance(own, str) else _BASELINE_NAMES[int(own)]
        if name in _CONFIG_ATTRS:
            v = getattr(b.cfg, name)
            if name == "workload_cap":
                return None if v < 0 else v
            return bool(v) if name in ("zero_day", "fast_scan", "scaling_vulnerability") else v
        if name == "γ":
            return b.cfg.gamma
        if name == "Max_network_size":
            return b.M
        if name == "numOfDevice":
            return b.cfg.num_of_device
        if name == "Min_network_size":
            return b.cfg.min_network_size
        if name == "MaxExploits":
            return b.cfg.max_exploits
        if name == "defender_action_space":
            return Discrete(b.cfg.n_def_actions)
        if name == "attacker_action_space":
            return Discrete(b.cfg.n_att_actions)
        if name == "observation_space":
            return Box(low=0, high=1, shape=(b.M * 6,), dtype=np.float32)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        b = object.__getattribute__(self, "_b")
        i = object.__getattribute__(self, "_i")
        if name in _COUNTER_COLS:
            self._pend_i[_COUNTER_COLS[name]] = int(value)
            if self not in b._dirty_views:
                b._dirty_views.append(self)
        elif name in _FLOAT_COLS:
            self._pend_f[_FLOAT_COLS[name]] = float(value)
            if self not in b._dirty_views:
                b._dirty_views.append(self)
        elif name == "base_line":
            if value not in abi.BASELINES:
                raise ValueError(f"unknown base_line {value!r}")
            object.__setattr__(self, "_base_line", value)   # THIS env's: carried in its mode word every tick (_launch)
        elif name in _CONFIG_ATTRS:
            if name == "workload_cap":
                value = -1 if value is None else int(value)
            elif name in ("zero_day", "fast_scan", "scaling_vulnerability"):
                value = int(bool(value))
            b.set_config(dataclasses.replace(b.cfg, **{name: value}))
        elif name == "γ":
            b.set_config(dataclasses.replace(b.cfg, gamma=float(value)))
        elif name == "numOfDevice":
            b.set_config(dataclasses.replace(b.cfg, num_of_device=int(value)))
        elif name in ("Max_network_size", "MaxExploits"):
            if int(value) != getattr(self, name):
                raise ValueError(f"{name} is fixed by the topology of the batch")
        else:
            object.__setattr__(self, name, value)

    # ---- counters: one row copy per batch epoch, writes coalesced ---------------
    def _counter_rows(self):
        """(ienv row, fenv row) of this env as numpy arrays: ONE device-to-host copy, reused until the batch launches again
        (or its `state` is handed out): every launch bumps `batch._epoch`."""
        b, i = self._b, self._i
        rows = object.__getattribute__(self, "_rows")
        if rows is None or rows[0] != b._epoch:
            st = b._state     # (not `b.state`: that access would expire the row being fetched)
            pack = torch.cat([st["ienv"][i].double(), st["fenv"][i]]).cpu().numpy()
            rows = (b._epoch, pack[:S.I_COUNT].astype(np.int64), pack[S.I_COUNT:].copy())
            object.__setattr__(self, "_rows", rows)
        return rows[1], rows[2]

    def _flush(self):
        """Upload the pending counter writes of this env: one row copy per touched table (called by the batch before its
        next launch and whenever `batch.state` is handed out)."""
        pi, pf = self._pend_i, self._pend_f
        if not pi and not pf:
            return
        b, i = self._b, self._i
        ie, fe = self._counter_rows()
        if pi:
            for col, v in pi.items():
                ie[col] = v
            b._state["ienv"][i].copy_(torch.from_numpy(ie.astype(np.int32)))
            pi.clear()
        if pf:
            for col, v in pf.items():
                fe[col] = v
            b._state["fenv"][i].copy_(torch.from_numpy(fe.astype(np.float64)))
            pf.clear()

    # ---- small helpers of the reference -----------------------------------
    def get_num_action_types(self, mode=None):           # volt_typhoon_env.py:514-520
        if mode == "defender":
            return 14
        if mode == "attacker":
            return 3
        raise ValueError("Invalid mode: must be either 'defender' or 'attacker'")

    def get_device_indices(self):                        # :522
        return list(range(self._b.M))

    def get_num_exploit_indices(self):                   # :525
        return self._b.topo.X

    def get_num_app_indices(self):                       # :528-533 (unique app ids: every device has its own)
        return int(self._b.topo.napps.astype(np.int64).sum())

    def seed(self, seed=None):                           # CyberDefenseEnv.py:261-268
        """The reference seeds the process-global `random` / `numpy.random` streams its tick draws from.  Here the tick's
        draws are addressed Philox draws keyed by the batch's seed, so this sets THAT seed -- for
        every env of the batch, from the next tick on (an env keeps its own stream through its env id) -- together with the
        host streams `sample_action` uses.  seed=None draws one, like gym's seeding helper."""
        if seed is None:
            seed = random.SystemRandom().getrandbits(63)
        seed = int(seed)
        if seed < 0:
            raise ValueError("seed must be a non-negative integer")
        random.seed(seed)
        np.random.seed(seed & 0xFFFFFFFF)
        self._b.set_config(dataclasses.replace(self._b.cfg, seed=seed & 0xFFFFFFFFFFFFFFFF))
        return [seed]

    def set_exploit_seed(self, seed: int):               # CyberDefenseEnv.py:65-72
        object.__setattr__(self, "_exploit_seed", seed)
        object.__setattr__(self, "_rng", np.random.RandomState(seed))

    def sample_exploits(self):                           # CyberDefenseEnv.py:73-85
        if not hasattr(self, "_rng"):
            object.__setattr__(self, "_rng", np.random.RandomState())
        return self._rng.randint(low=0, high=self.get_num_exploit_indices(), size=(self._b.M,))

    def _flags(self) -> np.ndarray:
        return self._b.state["flags"][self._i].cpu().numpy()

    def sample_action(self):                             # CyberDefenseEnv.py:555-578
        if self.mode == "defender":
            at = self.defender_action_space.sample()
        elif self.mode == "attacker":
            at = self.attacker_action_space.sample()
        else:
            raise ValueError("Invalid mode")
        devs = random.sample(list(range(self._b.M)), k=random.randint(1, max(1, self.numOfDevice)))
        exploit_indices = np.array([random.randrange(self.MaxExploits)], dtype=int)
        n_apps = self.get_num_app_indices()
        app_index = random.randint(0, n_apps - 1) if n_apps > 0 else 0
        return (at, exploit_indices, devs, app_index)

    # ---- observations (CyberDefenseEnv.py:146-257) --------------------------
    def _obs(self, role):
        return self._b.observe(role)[self._i].cpu().numpy()

    def _get_state(self):
        return self._obs(0).astype(np.float64)

    def _get_defender_state(self):
        return self._obs(1).astype(np.float64)

    def _get_attacker_state(self):
        return self._obs(2).astype(np.float32)

    @property
    def simulator(self):
        """Read-only object facade (`env.simulator.subnet.net[i].isCompromised`, `.graph`, `.exploits`,
        `.logger.get_logs()`), rebuilt from the current state on every access."""
        from .facade import SimulatorView
        return SimulatorView(self)

    def _get_ordered_devices(self):                      # CyberDefenseEnv.py:95-102
        net = self.simulator.subnet.net
        return [net[i] for i in sorted(net.keys())][: self._b.M]

    # ---- episode control -----------------------------------------------------
    def initialize_environment(self):
        """The network was built when the batch was created; re-arm this env from the snapshot."""
        return self.reset(from_init=True)

    def reset(self, from_init=True, *args, **kwargs):   # volt_typhoon_env.py:1904
        self._b.reset([self._i])
        self.state = self._get_state()
        return self.state

    def randomize_compromise_and_ownership(self):        # :330-383
        self._b.randomize([self._i])

    # ---- the tick -------------------------------------------------------------
    def _launch(self, groups, grouped, partial=False):
        """Tick THIS env only: its row of the batch's action tensors is written (a few bytes, host to device) and
        the launch covers the one-env range [i, i + 1) (cygym_step_range) -- the cost of a view's step does not
        depend on how many envs the batch holds."""
        b, i = self._b, self._i
        row = {k: np.zeros((1,) + tuple(v.shape[1:]), dtype=_NP_OF[v.dtype]) for k, v in b.act.items()}
        row["exploit"][:] = -1
        row["app"][:] = -1
        HL.encode_into(row, 0, self.mode, groups, grouped, b.M)
        if partial:
            row["mode"][0] |= S.MODE_PARTIAL
        # env.base_line of THIS env, this tick (it persists: the view sends it with every tick)
        row["mode"][0] |= (abi.BASELINES[self.base_line] + 1) << S.MODE_BASELINE_SHIFT
        for k, v in b.act.items():
            v[i:i + 1].copy_(torch.from_numpy(row[k]))
        obs, raw, shaped, done = b.step_range(i, 1)
        if b.detector and any(int(g[0]) == 10 for g in groups):
            b.service_detectors([i])     # Detector.train is synchronous in the reference (volt_typhoon_env.py:961); only THIS env's request
        # ONE device-to-host copy for everything the 6-tuple and `info` need: the observation, the rewards, done and the
        # env's counter rows (all exactly representable in f64), instead of a copy / .item() per field
        st = b._state
        epoch = b._epoch     # (after service_detectors: it may have changed this env's flag word)
        pack = torch.cat([obs[i].reshape(-1).double(), raw[i:i + 1], shaped[i:i + 1], done[i:i + 1].double(),
                          st["ienv"][i].double(), st["fenv"][i]]).cpu().numpy()      # (.cpu() synchronises with the launch)
        n = 6 * b.M
        self.state = pack[:n].copy()
        self._last_ie = pack[n + 3: n + 3 + S.I_COUNT].astype(np.int64)
        self._last_fe = pack[n + 3 + S.I_COUNT:]
        object.__setattr__(self, "_rows", (epoch, self._last_ie.copy(), self._last_fe.copy()))   # counter reads until the next launch: no copy
        return float(pack[n]), float(pack[n + 1]), bool(pack[n + 2])

    def _info(self, action_taken, executed=None, grouped=False, partial=False):
        ie, fe = self._last_ie, self._last_fe      # the rows _launch fetched with the observation
        # step() builds info before step_num += 1 (:1272 vs :1308), step_grouped after (:751 vs :759)
        info = {
            "mode": self.mode, "step_count": int(ie[S.I_STEP_NUM]) - (0 if (grouped or partial) else 1),
            "revert_count": int(ie[S.I_REVERT_CNT]),
            "checkpoint_count": int(ie[S.I_CKPT_CNT]), "defensive_cost": float(fe[S.D_DEF_COST]),
            "clearning_cost": float(fe[S.D_CLEAN_COST]), "Scan_count": int(ie[S.I_SCAN_CNT]),
            "action_taken": action_taken, "work_done": int(ie[S.I_WORK_DONE]),
            "Compromised_devices": int(ie[S.I_COMP_CNT]), "Edges Blocked": int(ie[S.I_EDGES_BLOCKED]),
            "Edges Added": int(ie[S.I_EDGES_ADDED]),
        }
        if executed is not None:
            info["executed_atype"] = executed
        return info

    def _logs(self, total=None):
        """The tail of `simulator.logger.logs` (CDSimulator.py:667-676): the last 32 entries, or the last 2048 when the
        batch keeps the long history (detector=True).  `time_step` is `simulator.system_time`, which only the base
        class's `step` advances (CyberDefenseEnv.py:408) -- on this path it stays 0; every entry is of kind 'A'
        (volt_typhoon_env.py:1161)."""
        total = int(self._b.state["ienv"][self._i, S.I_LOG_TOTAL].item()) if total is None else int(total)
        if getattr(self._b, "detector", False) and self._b.state["hist"].numel() > 0:
            cap, ring = S.HIST_RING, self._b.state["hist"][self._i].cpu().numpy().view(np.uint16).reshape(S.HIST_RING, 2)
        else:
            cap, ring = S.LOG_RING, self._b.state["ring"][self._i].cpu().numpy().view(np.uint16).reshape(S.LOG_RING, 2)
        n = min(total, cap)
        out = []
        for j in range(total - n, total):
            f, t = ring[j % cap]
            out.append({"time_step": 0, "from_device": int(f), "to_device": int(t), "kind": "A"})
        return out

    def step(self, action, agent_cnt=None):              # volt_typhoon_env.py:818
        if HL.is_grouped(action):
            return self.step_grouped(action)          # the reference ignores agent_cnt here (:842-844)
        partial = agent_cnt is not None and agent_cnt != self._b.M   # :1207 / :1307
        if action is None:
            action = HL.default_action(self.mode, self.base_line, self._flags())
        cfg = self._b.cfg
        norm = HL.validate_single(self.mode, self.base_line, action, self._b.M, cfg.n_def_actions, cfg.n_att_actions)
        raw, shaped, done = self._launch([norm], grouped=False, partial=partial)
        executed = int(self._last_ie[S.I_LAST_ATYPE])
        return self.state, raw, shaped, done, self._info(action, executed, partial=partial), self._logs(self._last_ie[S.I_LOG_TOTAL])

    def step_grouped(self, groups):                      # :694-779
        assert isinstance(groups, (list, tuple)) and len(groups) > 0
        HL.mode_code(self.mode)
        norm = []
        for g in groups:
            at, ex, dv, app = g
            dv = HL._as_list(dv)
            for d in dv:                                  # :670-671 indexes every listed device
                if not (0 <= d < self._b.M):
                    raise KeyError(d)
            eff = 8 if (self.mode == "defender" and int(at) == 0) else int(at)
            if self.mode == "defender" and self.base_line == "Nash" and eff == 11 and len(dv) == 0:
                raise ValueError("Action 11 requires exactly one device index")
            norm.append((int(at), HL._as_list(ex), dv, HL.app_index_value(app)))
        raw, shaped, done = self._launch(norm, grouped=True)
        return self.state, raw, shaped, done, self._info(groups, grouped=True), self._logs(self._last_ie[S.I_LOG_TOTAL])
