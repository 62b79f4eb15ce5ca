"""Host-side logic of the reference's step() that is not per-tick device work:
default actions for `action=None`, argument validation, action encoding.

Pure numpy / Python (no GPU): tested on CPU against the reference fixtures.
"""
from __future__ import annotations

import numpy as np

from . import spec as S

DEFENDER, ATTACKER = "defender", "attacker"
PER_DEVICE_DEF = (1, 4, 5, 6, 7, 9, 12, 13)   # volt_typhoon_env.py:983-986


def mode_code(mode) -> int:
    if mode == DEFENDER:
        return S.MODE_DEFENDER
    if mode == ATTACKER:
        return S.MODE_ATTACKER
    raise ValueError("Invalid mode: must be either 'defender' or 'attacker'")


def default_action(mode: str, base_line: str, flags_row: np.ndarray):
    """The action the reference substitutes for `action=None` (volt_typhoon_env.py:847-874)."""
    f = np.asarray(flags_row)
    ids = np.arange(f.shape[0])
    if mode == DEFENDER:
        if base_line == "No Defense":
            sel = ((f & S.F_OWNED) == 0) & ((f & S.F_NYA) == 0)
            return (8, [0], [int(i) for i in ids[sel]], 0)
        return (7, [0], [], 0)            # "Preset" and everything else
    if mode == ATTACKER:
        if base_line == "No Attack":
            sel = ((f & S.F_KNOWN) != 0) & ((f & S.F_NYA) == 0)
            return (3, [0], [int(i) for i in ids[sel]], 0)
        return (2, [0], [], 0)
    raise ValueError("Invalid mode: must be either 'defender' or 'attacker'")


def is_grouped(action) -> bool:
    """step() dispatches to step_grouped for a non-empty list/tuple of list/tuples (:842-844)."""
    return isinstance(action, (list, tuple)) and len(action) > 0 and isinstance(action[0], (list, tuple))


def _as_list(x):
    if x is None:
        return []
    return [int(v) for v in np.asarray(x).reshape(-1)]


def app_index_value(app_index) -> int:
    """`isinstance(app_index, int)` gate of action 4 (:1015): numpy integers do not pass it."""
    if isinstance(app_index, int) and not isinstance(app_index, bool):
        return int(app_index)
    if isinstance(app_index, bool):
        return int(app_index)   # bool is an int subclass in Python
    return -1


def validate_single(mode: str, base_line: str, action, M: int, n_def: int, n_att: int):
    """Raise what the reference would raise for this action (KeyError for unknown devices,
    ValueError for action 11 without a device); return the normalised tuple."""
    at, ex, dv, app = action
    at = int(at)
    dv = _as_list(dv)
    eff = at
    if mode == DEFENDER:
        if not (0 <= eff < n_def):
            eff = 8
        if base_line != "Nash":
            eff = 8
        if eff == 11 and len(dv) == 0:
            raise ValueError("Action 11 requires exactly one device index")
        touch = dv if eff in PER_DEVICE_DEF else (dv[:1] if eff in (10, 11) else [])
        for d in touch:
            if not (0 <= d < M):
                raise KeyError(d)
    return at, _as_list(ex), dv, app_index_value(app)


def encode_into(act: dict, row: int, mode: str, groups, grouped: bool, M: int):
    """Write one env's action(s) into row `row` of numpy action arrays shaped like the C ABI."""
    G = act["atype"].shape[1]
    L = act["dev_idx"].shape[1]
    if len(groups) > G:
        raise ValueError(f"{len(groups)} groups exceed max_groups={G}")
    act["mode"][row] = mode_code(mode)
    act["n_groups"][row] = len(groups) if grouped else 0
    used = 0
    for g, (at, ex, dv, app) in enumerate(groups):
        ex = _as_list(ex)[: S.MAX_EXPLOITS]
        dv = _as_list(dv)
        if used + len(dv) > L:
            raise ValueError(f"device lists ({used + len(dv)} entries) exceed max_devs={L}")
        act["atype"][row, g] = int(at)
        act["n_exploit"][row, g] = len(ex)
        act["exploit"][row, g, :] = -1
        act["exploit"][row, g, : len(ex)] = ex
        act["app"][row, g] = app_index_value(app)
        act["dev_cnt"][row, g] = len(dv)
        act["dev_idx"][row, used: used + len(dv)] = dv
        used += len(dv)


def group_actions_np(types, visible, exploit, app, n_types, noop, single_types, picks=None):
    """numpy restatement of cygym_group_actions for ONE env (IPPO.py:560-572): the list of groups
    [(type, [exploit], [device ids], app), ...] `env.step(groups)` receives.  `picks[t]` = index of the device a
    single-device type keeps among its devices (the reference: random.choice(devs); the library: the addressed Philox draw)."""
    groups = []
    for t in range(int(n_types)):
        if t == noop:
            continue
        devs = [int(i) for i in np.nonzero((np.asarray(visible) != 0) & (np.asarray(types) == t))[0]]
        if not devs:
            continue
        if t in set(single_types):
            devs = [devs[int(picks[t]) if picks is not None else 0]]
        groups.append((t, [int(exploit)], devs, int(app)))
    return groups or [(int(noop), [0], [], 0)]
