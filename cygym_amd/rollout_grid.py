"""Batched rollout consumer: the |D| x |A| x N_MC grid of
independent episodes that the reference evaluates with a process pool
(`build_payoff_matrices` do_agent.py:1666-1753 -> `simulate_game` :1875-2089, one
`env.step` per tick per process) becomes ONE batch -- a cell (i, j, mc) is an env slot --
and, for open-loop strategies, ONE launch of cygym_rollout.

Open-loop strategies are the ones whose action at tick t does not depend on the
observation: the reference's baselines (`action=None` with base_line in {"No Defense",
"No Attack", "Preset"}) and fixed sequences (`strat.actions[t % len(strat.actions)]`,
do_agent.py:237-238 -- indexed with the GLOBAL tick t, not the role's turn number):
`payoff_grid`, one cygym_rollout launch.

Closed-loop strategies (anything that maps the role observation to an action every tick --
the reference's actor networks, do_agent.py:212-262) run through `simulate_grid`: per tick
one batched policy evaluation per distinct strategy (torch, on the device, on the rows of the
cells that play it), one fused scatter of the chosen actions into the batch's action tensors
(cygym_write_actions) and one cygym_step launch, which also emits the NEXT actor's role view
(cygym_outputs.obs_def / obs_att) -- no cygym_observe launch in between.  Nothing in that loop
touches the host unless a strategy can ask for Detector.train (defender action 10): the
reference trains synchronously inside that tick (volt_typhoon_env.py:961), so the loop then
reads the batch's 4-byte status word after defender ticks and services the requests before
the next tick.

`env.base_line` is per cell and per turn, as in the reference's loop (do_agent.py:218-221): a
baseline strategy assigns its name before its step and the assignment STAYS for the other
role's turns (volt_typhoon_env.py:913-914 then turns every defender action into a no-op,
:1126 skips the attacker).  It travels in the mode word of every tick (CG_MODE_BASELINE).
"""
from __future__ import annotations

import numpy as np
import torch

from . import abi
from . import host_logic as HL
from ._lib import CygymError as _CygymError, EUNSUPPORTED as _EUNSUPPORTED
from . import sharding
from . import spec as S

BASELINE_DEF = {"No Defense": (8, [0], [], 0), "Preset": (7, [0], [], 0)}
BASELINE_ATT = {"No Attack": (3, [0], [], 0), "Preset": (2, [0], [], 0)}


def _baseline_code(strategy, role):
    """abi.BASELINES code of a baseline-name strategy, -1 for everything else."""
    if not isinstance(strategy, str):
        return -1
    table = BASELINE_DEF if role == HL.DEFENDER else BASELINE_ATT
    if strategy not in table:
        raise ValueError(f"unknown {role} baseline {strategy!r}")
    return abi.BASELINES[strategy]


def _action_at(strategy, t, role):
    """strategy: a baseline name, or a list of reference-style 4-tuples cycled with the global tick t
    (`strat.actions[t % len(strat.actions)]`, do_agent.py:237-238)."""
    if isinstance(strategy, str):
        table = BASELINE_DEF if role == HL.DEFENDER else BASELINE_ATT
        if strategy not in table:
            raise ValueError(f"unknown {role} baseline {strategy!r}")
        # what step(None) substitutes (:847-874); the device lists of the reference's defaults only feed
        # no-op action types, so the empty list is equivalent
        return table[strategy]
    return strategy[t % len(strategy)]


def _cfg_baseline_code(batch) -> int:
    b = batch.cfg.baseline
    return abi.BASELINES[b] if isinstance(b, str) else int(b)


def baseline_schedule(def_strategies, att_strategies, cells: np.ndarray, n_mc: int, T: int, start_code: int) -> np.ndarray:
    """[T, n] int32: env.base_line (abi.BASELINES code) of every cell at every tick of the reference's loop.  It starts
    as `start_code`; on a role's turn a baseline strategy of that role overwrites it, and it persists otherwise."""
    nA = len(att_strategies)
    dcode = np.array([_baseline_code(s, HL.DEFENDER) for s in def_strategies], np.int32)[cells // (nA * n_mc)]
    acode = np.array([_baseline_code(s, HL.ATTACKER) for s in att_strategies], np.int32)[(cells // n_mc) % nA]
    out = np.zeros((T, len(cells)), np.int32)
    cur = np.full(len(cells), start_code, np.int32)
    for t in range(T):
        code = dcode if t % 2 == 0 else acode
        cur = np.where(code >= 0, code, cur)
        out[t] = cur
    return out


def payoff_grid(batch, def_strategies, att_strategies, n_mc: int, T: int, randomize: bool = True,
                group=None, n_total: int | None = None, cell_offset: int = 0):
    """Fill the batch with the cells [cell_offset, cell_offset + batch.N) of the row-major grid
    (i over defender strategies, j over attacker strategies, mc), run T alternating ticks
    (defender on even ticks, do_agent.py:207) in one fused launch and return
    (U_def [|D|,|A|], U_att [|D|,|A|]) = mean over mc of the per-role reward sums
    (`def_total += r` on defender turns, `att_total += r` on attacker turns, :266-270).

    A script that carries defender action 10 (Detector.train) is cut after those ticks on a batch created with
    detector=True (launch, service, launch: BatchedCyberDefenseEnv.rollout); without detector buffers a later
    scan raises instead of answering all-"D" silently.

    With torch.distributed initialised, ranks hold consecutive slices of the grid
    (cygym_amd.sharding) and the per-cell sums are all-gathered before averaging."""
    nD, nA = len(def_strategies), len(att_strategies)
    cells = nD * nA * n_mc
    n_total = cells if n_total is None else n_total
    if cell_offset + batch.N > cells:
        raise ValueError("batch holds more envs than grid cells")
    batch.reset()
    if randomize:
        batch.randomize()                                  # do_agent.py:189-190
    act, out = batch.alloc_rollout(T)
    host = {k: v.cpu().numpy() for k, v in act.items()}
    cell = np.arange(cell_offset, cell_offset + batch.N)
    rows_of = {HL.DEFENDER: [np.nonzero(cell // (nA * n_mc) == i)[0] for i in range(nD)],
               HL.ATTACKER: [np.nonzero((cell // n_mc) % nA == j)[0] for j in range(nA)]}
    one = {k: np.zeros((1,) + v.shape[2:], v.dtype) for k, v in host.items()}   # one env's row, encoded once per (strategy, tick)
    for t in range(T):
        role = HL.DEFENDER if t % 2 == 0 else HL.ATTACKER
        for k, strat in enumerate(def_strategies if role == HL.DEFENDER else att_strategies):
            r = rows_of[role][k]
            if r.size == 0:
                continue
            one["exploit"][:] = -1
            one["app"][:] = -1
            one["dev_idx"][:] = 0
            HL.encode_into(one, 0, role, [_action_at(strat, t, role)], False, batch.M)
            for key, v in host.items():
                v[t, r] = one[key][0]
    bl = baseline_schedule(def_strategies, att_strategies, cell, n_mc, T, _cfg_baseline_code(batch))
    host["mode"] |= ((bl + 1) << S.MODE_BASELINE_SHIFT).astype(host["mode"].dtype)
    for k, v in act.items():
        v.copy_(torch.from_numpy(host[k]))
    batch.rollout(act, out)
    raw = out["raw"]                                        # [T, N]
    def_sum = raw[0::2].sum(dim=0)
    att_sum = raw[1::2].sum(dim=0)
    both = torch.stack([def_sum, att_sum], dim=1)           # [N, 2]
    both = sharding.gather_by_env(both, n_total, group)     # no-op on one rank
    if both.shape[0] != cells:
        raise ValueError("gathered cells do not cover the grid")
    g = both.reshape(nD, nA, n_mc, 2).mean(dim=2)
    return g[..., 0].cpu().numpy(), g[..., 1].cpu().numpy()


# ------------------------------------------------------------------------------------------
# closed loop
# ------------------------------------------------------------------------------------------
class SequencePolicy:
    """A baseline name or a fixed action sequence as a (trivially) closed-loop policy: the same strategies
    `payoff_grid` accepts, so that mixed grids (baseline rows against neural columns) run through one loop.
    Called with the GLOBAL tick (the reference indexes `strat.actions[t % len]` with it)."""

    uses_global_tick = True

    def __init__(self, strategy, role):
        self.strategy, self.role = strategy, role
        seq = [_action_at(strategy, 0, role)] if isinstance(strategy, str) else list(strategy)
        self.action_types = sorted({int(a[0]) for a in seq})      # what this policy can emit (see simulate_grid)

    def __call__(self, obs, t, M, L):
        a = _action_at(self.strategy, t, self.role)
        n = obs.shape[0]
        dv = HL._as_list(a[2])[:L]
        dev_idx = torch.zeros((n, L), dtype=torch.int16, device=obs.device)
        if dv:
            dev_idx[:, : len(dv)] = torch.tensor(dv, dtype=torch.int16, device=obs.device)
        ex = HL._as_list(a[1])
        return {"atype": torch.full((n,), int(a[0]), dtype=torch.int32, device=obs.device),
                "exploit": torch.full((n,), int(ex[0]) if ex else -1, dtype=torch.int32, device=obs.device),
                "dev_idx": dev_idx, "dev_cnt": torch.full((n,), len(dv), dtype=torch.int32, device=obs.device),
                "app": torch.full((n,), HL.app_index_value(a[3]), dtype=torch.int32, device=obs.device)}


def mask_to_list(mask: torch.Tensor, L: int):
    """[n, M] bool device mask -> (dev_idx [n, L] int16 ascending ids, dev_cnt [n] int32), on the device.
    (Torch fallback of cygym_write_actions' in-kernel compaction; used for batches without that entry point.)"""
    order = torch.sort(mask.to(torch.int8), dim=1, descending=True, stable=True).indices   # chosen ids first, ascending
    cnt = mask.sum(dim=1).clamp(max=L).to(torch.int32)
    idx = order[:, :L].to(torch.int16)
    idx = torch.where(torch.arange(L, device=mask.device)[None, :] < cnt[:, None], idx, torch.zeros_like(idx))
    return idx.contiguous(), cnt


def can_train(policy) -> bool:
    """May this defender policy ever emit action 10 (Detector.train)?  A policy that declares `action_types` (the
    set of action types it can emit) answers statically; anything else is assumed to."""
    types = getattr(policy, "action_types", None)
    if types is None:
        types = getattr(policy, "types", None)
    return True if types is None else (10 in set(int(x) for x in types))


def _write_rows(batch, act, r, a, L):
    """Scatter one strategy's chosen actions (dict of [n] / [n, L] / [n, M] device tensors) into rows `r` of the
    batch's action tensors: ONE launch of the library's fused scatter (which also compacts a device mask into the
    ascending id list), or the torch fallback for batch-likes without it (the oracle harness of the tests)."""
    if hasattr(batch, "write_actions"):
        batch.write_actions(r, a, act)
        return
    if "dev_mask" in a:
        a["dev_idx"], a["dev_cnt"] = mask_to_list(a["dev_mask"], L)
    act["atype"][:, 0].index_copy_(0, r, a["atype"].to(torch.int32))
    act["exploit"][:, 0, 0].index_copy_(0, r, a["exploit"].to(torch.int32))
    act["n_exploit"][:, 0].index_copy_(0, r, (a["exploit"] >= 0).to(torch.int32))
    act["app"][:, 0].index_copy_(0, r, a["app"].to(torch.int32))
    act["dev_cnt"][:, 0].index_copy_(0, r, a["dev_cnt"].to(torch.int32))
    act["dev_idx"].index_copy_(0, r, a["dev_idx"].to(torch.int16))


def _group_actors(batch, items, M, env_order_rpg=None):
    """Strategies of one role and sub-batch that are actor networks of ONE architecture with equally many rows (a multiple
    of 16) each are evaluated as a population: one launch for all of them (policies.ActorPolicyGroup).  env_order_rpg: the
    sub-batch is the whole grid in env order and env e plays strategy (e // env_order_rpg) % len(items) -- the population then
    needs no row ids at all (rows_per_group of cygym_actor_mlp)."""
    from .policies import ActorPolicyGroup
    if len(items) < 2 or not hasattr(batch, "actor_head_decode"):
        return items
    keys = [ActorPolicyGroup.key(p, M) for p, _, _, _, _ in items]
    n0 = int(items[0][1].numel())
    if keys[0] is None or any(k != keys[0] for k in keys) or n0 % 16 or any(int(it[1].numel()) != n0 for it in items):
        return items
    r64 = torch.cat([it[1] for it in items])
    ids = r64.cpu().numpy()
    sl = slice(int(ids[0]), int(ids[-1]) + 1) if (np.diff(ids) == 1).all() else None      # (the defender's strategies: one ascending range)
    grp = ActorPolicyGroup([it[0] for it in items])
    rpg = env_order_rpg if (env_order_rpg and env_order_rpg % 16 == 0 and grp.fused_mlp(batch)) else None
    return [(grp, r64, r64.to(torch.int32), sl, rpg)]


def simulate_grid(batch, def_policies, att_policies, n_mc: int, T: int, randomize: bool = True,
                  group=None, n_total: int | None = None, cell_offset: int = 0, timers: dict | None = None,
                  graph: bool = False, streams: int = 1, merge_launches: bool | None = None, local_only: bool = False):
    """The |D| x |A| x n_mc grid of `simulate_game` (do_agent.py:1875-2089 / worker :129-287) with CLOSED-LOOP
    strategies, as one batch: cell (i, j, mc) is env slot i*|A|*n_mc + j*n_mc + mc.

    A policy is a callable `policy(obs, t, M, L) -> dict` evaluated on the device for all the cells that play it:
      obs      [n, W] float32 role observation of those cells (defender: 6M, attacker: 4M + MaxExploits;
               CyberDefenseEnv.py:194-257), a device tensor
      t        the role's turn number (tick // 2)
      returns  atype [n] i32, exploit [n] i32 (one exploit index, -1 = none), dev_idx [n, L] i16 + dev_cnt [n] i32
               (or `dev_mask` [n, M] bool instead of the two), app [n] i32 -- device tensors
    or an object with `write(batch, act, rows, obs)` that fills the rows itself (cygym_amd.policies.ActorPolicy: actor
    forward + ONE fused decode-and-scatter launch).  A policy may declare `action_types` (iterable of the action
    types it can emit) and `tick_free = True` (its action does not depend on t).  Baseline names and fixed sequences
    are accepted too (wrapped in SequencePolicy; they follow the global tick like the reference).

    Per tick: one policy call per distinct strategy of the acting role -> rows scattered into the batch's action
    tensors (one fused launch per strategy) -> cygym_step, which also writes the next actor's role view and adds the
    reward to the env's episode return while it is alive (an env that reports done stops contributing: the
    reference breaks out of its loop, :271-274).

    streams = S > 1: the batch is walked as S contiguous sub-batches, each on its own HIP stream (cygym_step_range):
    a sub-batch's policy evaluation overlaps the other sub-batches' ticks, and its next tick starts when ITS slowest
    env is done -- the batched counterpart of the reference's process-per-rollout fan-out (do_agent.py:1928-1942).
    graph=True: when every policy is tick_free and none can train, ticks 6, 7 are captured in a HIP graph (all
    streams) and replayed for the rest of the horizon (no host work per tick); otherwise the flag is ignored.

    merge_launches: where the next role's whole plan is one actor launch over every env in env order and the batch has the
    shape for it (BatchedCyberDefenseEnv.can_step_actor: 256 devices, fixed topology, at most 16 envs per CU), a tick and the
    NEXT role's actor run as ONE launch (cygym_step_actor) -- a turn of the loop per launch.  Default: in the eager loop only
    (+5 % there; a replayed HIP graph already hides the second launch's ramp: 31.4 vs 32.0 us per tick at 1 x 1 x 4096,
    34.7 vs 34.4 at 2 x 2 x 1024).

    Detector.train (defender action 10): when some defender policy can emit it the batch must have been created
    with detector=True; after every defender tick the 4-byte status word says whether any env asked, and the
    requests are serviced before the next tick (the reference trains inside the tick, :961).  At the end any env
    that ran a scan without a current forest (CG_E_UNPINNED) raises -- payoffs are never returned from all-"D"
    scans silently.

    local_only: the batch holds the WHOLE grid even though torch.distributed is initialised (every rank runs a grid of its
    own: bench.py's weak-scaling leg) -- no gather across ranks.

    Returns (U_def, U_att) [|D|, |A|]: mean over mc of the per-role reward sums."""
    nD, nA = len(def_policies), len(att_policies)
    cells = nD * nA * n_mc
    n_total = cells if n_total is None else n_total
    N, M, L, dev = batch.N, batch.M, batch.L, batch.obs.device
    if cell_offset + N > cells:
        raise ValueError("batch holds more envs than grid cells")
    ROLES = (HL.DEFENDER, HL.ATTACKER)
    pol = {HL.DEFENDER: [p if (callable(p) or hasattr(p, "write")) else SequencePolicy(p, HL.DEFENDER) for p in def_policies],
           HL.ATTACKER: [p if (callable(p) or hasattr(p, "write")) else SequencePolicy(p, HL.ATTACKER) for p in att_policies]}
    trains = any(can_train(p) for p in pol[HL.DEFENDER])
    has_det = bool(getattr(batch, "detector", False))
    if trains and not has_det:
        raise ValueError("a defender strategy can emit action 10 (Detector.train): create the batch with detector=True "
                         "(or declare the policy's `action_types` without 10)")
    fused = hasattr(batch, "role_obs")            # the product batch: fused role views, scatter, return accumulators
    if merge_launches is None:
        merge_launches = not graph
    split = timers is not None and bool(timers.get("split"))
    S_sub = max(1, min(int(streams), N)) if (fused and not trains and not split) else 1
    cell_np = np.arange(cell_offset, cell_offset + N)
    strat_np = {HL.DEFENDER: cell_np // (nA * n_mc), HL.ATTACKER: (cell_np // n_mc) % nA}
    bounds = [(j * N // S_sub, (j + 1) * N // S_sub) for j in range(S_sub)]
    # plan[role][sub-batch] = [(policy, env ids int64, env ids int32, slice or None when the ids are not one range)]
    plan = {r: [] for r in ROLES}
    for r in ROLES:
        for lo, hi in bounds:
            items = []
            for k, p in enumerate(pol[r]):
                ids = lo + np.nonzero(strat_np[r][lo:hi] == k)[0]
                if ids.size == 0:
                    continue
                sl = slice(int(ids[0]), int(ids[-1]) + 1) if ids[-1] - ids[0] + 1 == ids.size else None
                t64 = torch.from_numpy(ids).to(dev)
                items.append((p, t64, t64.to(torch.int32), sl, None))
            # the whole grid in env order: env e plays defender strategy e // (nA n_mc) and attacker strategy (e // n_mc) % nA
            rpg = (nA * n_mc if r == HL.DEFENDER else n_mc) if (S_sub == 1 and cell_offset == 0 and N == cells) else None
            plan[r].append(_group_actors(batch, items, M, rpg) if fused else items)
    bl = baseline_schedule(def_policies, att_policies, cell_np, n_mc, min(T, 4), _cfg_baseline_code(batch))
    # (ticks >= 2 repeat with period 2: rows 2 / 3 of the schedule; shorter runs only have the first rows)
    mode_words = [torch.from_numpy(((bl[t] + 1) << S.MODE_BASELINE_SHIFT) | (t % 2)).to(device=dev, dtype=torch.int32)
                  for t in range(bl.shape[0])]
    batch.reset()
    if randomize:
        batch.randomize()                                  # do_agent.py:189-190
    # one action dict per role: the roles share every tensor but the mode words (no per-tick copy in steady state)
    acts = {r: dict(batch.act) for r in ROLES}
    for r in acts:
        acts[r]["mode"] = torch.zeros_like(batch.act["mode"])
    batch.act["n_groups"].zero_()
    batch.act["n_exploit"].zero_()
    # a role whose strategies all build their observation on chip from the state (policies.ActorPolicy.reads_state) needs no
    # role-view tensor: the tick then does not write one
    needs_view = {r: not fused or any(not (hasattr(p, "reads_state") and p.reads_state(batch)) for items in plan[r] for p, _, _, _, _ in items)
                  for r in ROLES}

    def _one_launch(r):
        """The role's whole plan as ONE actor launch over every env in env order (so that a tick can carry it: cygym_step_actor)?"""
        if not fused or S_sub != 1 or trains or split or len(plan[r][0]) != 1:
            return False
        p, _, _, sl, rpg = plan[r][0][0]
        in_order = rpg is not None or (sl is not None and sl.start == 0 and sl.stop == N)
        return bool(in_order and hasattr(p, "reads_state") and p.reads_state(batch) and hasattr(batch, "can_step_actor")
                    and batch.can_step_actor(p.n_out(M)) and merge_launches)

    carried = {r: _one_launch(r) for r in ROLES}      # the role's actor runs inside the PREVIOUS tick's launch
    pending = {r: False for r in ROLES}               # ... and has already written the role's next actions
    if fused:
        batch.reset_returns()
        if needs_view[HL.DEFENDER]:
            batch.prime_view(HL.DEFENDER)         # the first actor's observation (every later one comes from the tick itself)
    else:
        totals = torch.zeros((N, 2), dtype=torch.float64, device=dev)
        alive = torch.ones(N, dtype=torch.bool, device=dev)
    # timers: {"split": True} asks for the per-tick split (observe / policy+scatter / step, HIP events around each part:
    # eager single-stream loop only); in any case the dict receives "loop_s" / "ticks" = HIP-event time of the tick loop
    ev = [] if split else None

    def mark():
        if ev is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append(e)

    def tick(t, j=0):
        """Tick t of sub-batch j (on the current stream)."""
        nonlocal totals, alive
        role = ROLES[t % 2]
        nxt = ROLES[(t + 1) % 2]
        act = acts[role]
        lo, hi = bounds[j]
        mark()
        obs = (batch.role_obs[role] if needs_view[role] else None) if fused else batch.observe(1 if role == HL.DEFENDER else 2)
        mark()
        if t < 4 and t < len(mode_words):
            act["mode"][lo:hi].copy_(mode_words[t][lo:hi])
        for p, r64, r32, sl, rpg in ([] if pending[role] else plan[role][j]):
            if fused and hasattr(p, "write_by_env") and p.fused_mlp(batch):
                whole = sl is not None and sl.start == 0 and sl.stop == N      # (every env, in order: no row-id indirection)
                if rpg is not None:
                    p.write_by_env(batch, act, None, obs, role, rows_per_group=rpg)      # a population in env order
                else:
                    p.write_by_env(batch, act, None if whole else r32, obs, role)      # whole actor + decode + scatter: one launch, no gather
                continue
            o = obs[sl] if sl is not None else obs.index_select(0, r64)
            if fused and hasattr(p, "write"):
                p.write(batch, act, r32, o)
            else:
                a = p(o, t if getattr(p, "uses_global_tick", False) else t // 2, M, L)
                if fused:
                    batch.write_actions(r32, a, act)
                else:
                    _write_rows(batch, batch.act, r64, a, L)
        mark()
        pending[role] = False
        if fused and carried[nxt] and t + 1 < T:
            # this tick AND the next role's actor as one launch: the actor writes the next tick's actions into acts[nxt]
            p, _, _, _, rpg = plan[nxt][0][0]
            kw = {"rows_per_group": rpg} if rpg is not None else {}
            try:
                p.write_by_env(batch, acts[nxt], None, None, nxt, step={"act": act, "view": None, "full_obs": False, "returns": True}, **kw)
                pending[nxt] = True
            except _CygymError as exc:      # (the handle's launch plan does not have the shared shape after all: two launches)
                if getattr(exc, "code", None) != _EUNSUPPORTED:      # (anything but "this launch shape is not available" is an error)
                    raise
                carried[HL.DEFENDER] = carried[HL.ATTACKER] = False
                batch.step_range(lo, hi - lo, act, view=nxt if needs_view[nxt] else None, full_obs=False, returns=True)
        elif fused:
            batch.step_range(lo, hi - lo, act, view=nxt if needs_view[nxt] else None, full_obs=False, returns=True)
        else:
            batch.act["mode"].copy_(act["mode"])
            _, raw, _, done = batch.step()
            totals[:, t % 2] += torch.where(alive, raw, torch.zeros_like(raw))
            alive = alive & (done == 0)
        mark()
        if trains and role == HL.DEFENDER and (batch.take_status() & S.E_DET_PENDING):
            batch.service_detectors()        # Detector.train is synchronous in the reference (volt_typhoon_env.py:961)

    def ticks(t0, t1, cur):
        """Ticks [t0, t1) of every sub-batch: sub-batch j on side stream j (fork from / join into `cur`)."""
        if S_sub == 1:
            for t in range(t0, t1):
                tick(t)
            return
        for st in side:
            st.wait_stream(cur)
        for t in range(t0, t1):
            for j, st in enumerate(side):
                with torch.cuda.stream(st):
                    tick(t, j)
        for st in side:
            cur.wait_stream(st)

    use_graph = (graph and fused and not trains and T >= 10 and not split
                 and all(getattr(p, "tick_free", False) for r in pol for p in pol[r]))
    side = [torch.cuda.Stream(device=dev) for _ in range(S_sub)] if S_sub > 1 else []
    on_gpu = dev.type == "cuda"
    if timers is not None and on_gpu:
        loop_ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        loop_ev[0].record()
    t = 0
    if use_graph:
        main = torch.cuda.current_stream(dev)
        ticks(0, 6, main)                         # mode words settle at tick 4; also the warm-up of the captured pair
        t = 6
        cap = torch.cuda.Stream(device=dev)
        cap.wait_stream(main)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap):
                ticks(t, t + 2, cap)
            # (capturing does not execute: ticks 6, 7 run with the first replay)
            while t + 2 <= T:
                g.replay()
                t += 2
        main.wait_stream(cap)
    if t < T:
        ticks(t, T, torch.cuda.current_stream(dev) if on_gpu else None)
    if timers is not None and on_gpu:
        loop_ev[1].record()
        loop_ev[1].synchronize()
        timers["loop_s"] = timers.get("loop_s", 0.0) + loop_ev[0].elapsed_time(loop_ev[1]) * 1e-3
        timers["ticks"] = timers.get("ticks", 0) + T
        timers["graph"] = bool(use_graph)
        timers["streams"] = S_sub
    if has_det or hasattr(batch, "unpinned_envs"):
        n_bad = batch.unpinned_envs()
        if n_bad:
            raise RuntimeError(f"{n_bad} env(s) ran a scan in trained-detector mode without a current forest "
                               "(CG_E_UNPINNED): their payoffs are not the reference's")
    if fused:
        if batch.take_status() & abi.DECODE_TRUNCATED:
            raise RuntimeError("a policy chose more devices than the batch's max_devs holds: create the batch with a larger max_devs")
        totals = batch.ret
    if ev is not None:
        torch.cuda.synchronize(dev)
        names = ("observe", "policy+scatter", "step")
        for j, name in enumerate(names):
            timers[name] = timers.get(name, 0.0) + sum(ev[4 * i + j].elapsed_time(ev[4 * i + j + 1]) for i in range(len(ev) // 4)) * 1e-3
    both = totals if local_only else sharding.gather_by_env(totals, n_total, group)   # no-op on one rank
    if both.shape[0] != cells:
        raise ValueError("gathered cells do not cover the grid")
    g = both.reshape(nD, nA, n_mc, 2).mean(dim=2)
    return g[..., 0].cpu().numpy(), g[..., 1].cpu().numpy()
