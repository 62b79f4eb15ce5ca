"""Batched rollout consumer: the |D| x |A| x N_MC grid of
independent episodes that the reference evaluates with a process pool
(`build_payoff_matrices` do_agent.py:1666-1753 -> `simulate_game` :1875-2089, one
`env.step` per tick per process) becomes ONE batch -- a cell (i, j, mc) is an env slot --
and, for open-loop strategies, ONE launch of cygym_rollout.

Open-loop strategies are the ones whose action at tick t does not depend on the
observation: the reference's baselines (`action=None` with base_line in {"No Defense",
"No Attack", "Preset"}) and fixed sequences (`strat.actions[t % len(strat.actions)]`,
do_agent.py:237-238): `payoff_grid`, one cygym_rollout launch.

Closed-loop strategies (anything that maps the role observation to an action every tick --
the reference's actor networks, do_agent.py:212-262) run through `simulate_grid`: per tick
one cygym_observe launch, one batched policy evaluation per distinct strategy (torch, on
the device, on the rows of the cells that play it) and one cygym_step launch.  Nothing in
that loop touches the host: observations, actions, rewards and the done mask stay device
tensors, so the loop can be enqueued ahead of the GPU (or captured in a HIP graph).
"""
from __future__ import annotations

import numpy as np
import torch

from . import host_logic as HL
from . import sharding
from . import spec as S

BASELINE_DEF = {"No Defense": (8, [0], [], 0), "Preset": (7, [0], [], 0)}
BASELINE_ATT = {"No Attack": (3, [0], [], 0), "Preset": (2, [0], [], 0)}


def _action_at(strategy, t, role):
    """strategy: a baseline name, or a list of reference-style 4-tuples (cycled)."""
    if isinstance(strategy, str):
        table = BASELINE_DEF if role == HL.DEFENDER else BASELINE_ATT
        if strategy not in table:
            raise ValueError(f"unknown {role} baseline {strategy!r}")
        # what step(None) substitutes (:847-874); the device lists of the reference's defaults only feed
        # no-op action types, so the empty list is equivalent
        return table[strategy]
    return strategy[t % len(strategy)]


def payoff_grid(batch, def_strategies, att_strategies, n_mc: int, T: int, randomize: bool = True,
                group=None, n_total: int | None = None, cell_offset: int = 0):
    """Fill the batch with the cells [cell_offset, cell_offset + batch.N) of the row-major grid
    (i over defender strategies, j over attacker strategies, mc), run T alternating ticks
    (defender on even ticks, do_agent.py:207) in one fused launch and return
    (U_def [|D|,|A|], U_att [|D|,|A|]) = mean over mc of the per-role reward sums
    (`def_total += r` on defender turns, `att_total += r` on attacker turns, :266-270).

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
    for n in range(batch.N):
        c = cell_offset + n
        i, j = c // (nA * n_mc), (c // n_mc) % nA
        for t in range(T):
            role = HL.DEFENDER if t % 2 == 0 else HL.ATTACKER
            a = _action_at(def_strategies[i] if role == HL.DEFENDER else att_strategies[j], t // 2, role)
            row = {k: v[t] for k, v in host.items()}
            HL.encode_into(row, n, role, [a], False, batch.M)
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
    `payoff_grid` accepts, so that mixed grids (baseline rows against neural columns) run through one loop."""

    def __init__(self, strategy, role):
        self.strategy, self.role = strategy, role

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
    """[n, M] bool device mask -> (dev_idx [n, L] int16 ascending ids, dev_cnt [n] int32), on the device."""
    order = torch.sort(mask.to(torch.int8), dim=1, descending=True, stable=True).indices   # chosen ids first, ascending
    cnt = mask.sum(dim=1).clamp(max=L).to(torch.int32)
    idx = order[:, :L].to(torch.int16)
    idx = torch.where(torch.arange(L, device=mask.device)[None, :] < cnt[:, None], idx, torch.zeros_like(idx))
    return idx.contiguous(), cnt


def simulate_grid(batch, def_policies, att_policies, n_mc: int, T: int, randomize: bool = True,
                  group=None, n_total: int | None = None, cell_offset: int = 0):
    """The |D| x |A| x n_mc grid of `simulate_game` (do_agent.py:1875-2089 / worker :129-287) with CLOSED-LOOP
    strategies, as one batch: cell (i, j, mc) is env slot i*|A|*n_mc + j*n_mc + mc.

    A policy is a callable `policy(obs, t, M, L) -> dict` evaluated on the device for all the cells that play it:
      obs      [n, W] float32 role observation of those cells (defender: 6M, attacker: 4M + MaxExploits;
               CyberDefenseEnv.py:194-257), a device tensor
      t        the role's turn number (tick // 2)
      returns  atype [n] i32, exploit [n] i32 (one exploit index, -1 = none), dev_idx [n, L] i16 + dev_cnt [n] i32
               (or `dev_mask` [n, M] bool instead of the two), app [n] i32 -- device tensors
    Baseline names and fixed sequences are accepted too (wrapped in SequencePolicy).

    Per tick: cygym_observe -> one policy call per distinct strategy of the acting role -> rows scattered into the
    batch's action tensors -> cygym_step.  An env that reports done stops contributing (the reference breaks out
    of its loop, :271-274).  Returns (U_def, U_att) [|D|, |A|]: mean over mc of the per-role reward sums."""
    nD, nA = len(def_policies), len(att_policies)
    cells = nD * nA * n_mc
    n_total = cells if n_total is None else n_total
    N, M, L, dev = batch.N, batch.M, batch.L, batch.obs.device
    if cell_offset + N > cells:
        raise ValueError("batch holds more envs than grid cells")
    pol = {HL.DEFENDER: [p if callable(p) else SequencePolicy(p, HL.DEFENDER) for p in def_policies],
           HL.ATTACKER: [p if callable(p) else SequencePolicy(p, HL.ATTACKER) for p in att_policies]}
    cell = torch.arange(cell_offset, cell_offset + N, device=dev)
    strat_of = {HL.DEFENDER: cell // (nA * n_mc), HL.ATTACKER: (cell // n_mc) % nA}
    rows = {r: [torch.nonzero(strat_of[r] == k).flatten() for k in range(len(pol[r]))] for r in pol}
    batch.reset()
    if randomize:
        batch.randomize()                                  # do_agent.py:189-190
    act = batch.act
    totals = torch.zeros((N, 2), dtype=torch.float64, device=dev)
    alive = torch.ones(N, dtype=torch.bool, device=dev)
    act["n_groups"].zero_()
    act["n_exploit"].zero_()
    for t in range(T):
        role = HL.DEFENDER if t % 2 == 0 else HL.ATTACKER
        obs = batch.observe(1 if role == HL.DEFENDER else 2)
        act["mode"].fill_(HL.mode_code(role))
        for k, p in enumerate(pol[role]):
            r = rows[role][k]
            if r.numel() == 0:
                continue
            a = p(obs.index_select(0, r), t // 2, M, L)
            if "dev_mask" in a:
                a["dev_idx"], a["dev_cnt"] = mask_to_list(a["dev_mask"], L)
            act["atype"][:, 0].index_copy_(0, r, a["atype"].to(torch.int32))
            act["exploit"][:, 0, 0].index_copy_(0, r, a["exploit"].to(torch.int32))
            act["n_exploit"][:, 0].index_copy_(0, r, (a["exploit"] >= 0).to(torch.int32))
            act["app"][:, 0].index_copy_(0, r, a["app"].to(torch.int32))
            act["dev_cnt"][:, 0].index_copy_(0, r, a["dev_cnt"].to(torch.int32))
            act["dev_idx"].index_copy_(0, r, a["dev_idx"].to(torch.int16))
        _, raw, _, done = batch.step()
        totals[:, t % 2] += torch.where(alive, raw, torch.zeros_like(raw))
        alive = alive & (done == 0)
    both = sharding.gather_by_env(totals, n_total, group)   # no-op on one rank
    if both.shape[0] != cells:
        raise ValueError("gathered cells do not cover the grid")
    g = both.reshape(nD, nA, n_mc, 2).mean(dim=2)
    return g[..., 0].cpu().numpy(), g[..., 1].cpu().numpy()
