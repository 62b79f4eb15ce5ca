"""Batched rollout consumer: the |D| x |A| x N_MC grid of
independent episodes that the reference evaluates with a process pool
(`build_payoff_matrices` do_agent.py:1666-1753 -> `simulate_game` :1875-2089, one
`env.step` per tick per process) becomes ONE batch -- a cell (i, j, mc) is an env slot --
and, for open-loop strategies, ONE launch of cygym_rollout.

Open-loop strategies are the ones whose action at tick t does not depend on the
observation: the reference's baselines (`action=None` with base_line in {"No Defense",
"No Attack", "Preset"}) and fixed sequences (`strat.actions[t % len(strat.actions)]`,
do_agent.py:237-238).  Closed-loop (neural) strategies keep using per-tick `step`.
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
