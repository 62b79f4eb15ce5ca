"""Batched transition collection for the reference's DDPG best-response training -- the data-collection half of
`DoubleOracle.ddpg_best_response`'s loop (do_agent.py:1334-1460; the same shape in utils.py:1060-1125) for every env of a batch
at once, all tensors on the device:

    turn = 'defender' if t % 2 == 0 else 'attacker'                                   (:1335)
    our turn:   raw = actor(state); vec = clip(raw + N(0, noise_std), -1, +1)          (:1369-1374)
                noise_std = max(sigma_min, noise_std * decay_rate)                      (:1375)
                action = decode_action(vec, n_types, D, E, A)                           (:1377-1383)
                _, raw_reward, reward, done = env.step(action); next_state = my_state   (:1407-1408)
                replay_buffer.push(state, vec, reward, next_state, done)                (:1424)
    their turn: the opponent strategy's action, env.step; state = my_state              (:1449-1456)

Here the actor runs once for all envs, the noise is one batched draw, decode_action + the scatter into the action tensors is
ONE launch (cygym_decode_actions: the action vectors have to exist in HBM anyway -- the replay buffer stores them) and the
tick writes the learner's next view itself.  `train_ddpg` (the update on the replay buffer) is the caller's.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import host_logic as HL
from . import spec as S


@dataclass
class Transitions:
    """What the loop pushes into the replay buffer (do_agent.py:1424), stacked: [T, N, ...] device tensors."""
    state: torch.Tensor        # [T, N, W] the learner's view at the decision
    action_vec: torch.Tensor   # [T, N, n_out] the clipped noisy action vector that was decoded
    reward: torch.Tensor       # [T, N] float64 shaped reward (the third return of env.step)
    raw_reward: torch.Tensor   # [T, N] float64
    next_state: torch.Tensor   # [T, N, W] the learner's view right after its own step.  On a row whose step reported `done` (batches
                               # with auto_reset) this is the view of the RELOADED state -- the tick writes the view of what it leaves behind
                               # -- where the reference pushes the terminal state's view; `done` masks the bootstrap term either way
    done: torch.Tensor         # [T, N] bool
    noise_std: float           # where the exploration schedule ended


@torch.no_grad()
def collect(batch, role: str, actor, opponent, n_decisions: int, n_types: int, n_exploits: int | None = None, n_apps: int = 0, *,
            type_map=None, noise_std: float = 0.0, sigma_min: float = 0.0, decay_rate: float = 1.0, clip=(-1.0, 1.0), generator=None,
            t0: int = 0) -> Transitions:
    """Collect `n_decisions` transitions of `role` in every env of `batch` (the for-loop of do_agent.py:1334-1460 without the
    update).  actor(state [N, W]) -> [N, n_types + M + n_exploits + n_apps] action vectors; opponent: a baseline name / fixed
    sequence, a policy(obs, t, M, L) -> action tensors, or an object with write(batch, act, rows, obs).  The loop's tick
    counter starts at `t0` (turns follow t % 2 like the reference's, not the envs' step_num).  The reference leaves its loop
    at the first done (:1439); a batch goes on: with auto_reset the env restarts from its snapshot, and `done` marks the row."""
    from .rollout_grid import SequencePolicy, _baseline_code
    if role not in (HL.DEFENDER, HL.ATTACKER):
        raise ValueError("role must be 'attacker' or 'defender'")
    other = HL.ATTACKER if role == HL.DEFENDER else HL.DEFENDER
    N, M, L, dev = batch.N, batch.M, batch.L, batch.device
    n_exploits = batch.cfg.max_exploits if n_exploits is None else int(n_exploits)
    opp = opponent if (callable(opponent) or hasattr(opponent, "write")) else SequencePolicy(opponent, other)
    bl_code = _baseline_code(opponent, other)      # a baseline opponent: env.base_line stays set from its first turn on
    cur_bl = None
    tm = None if type_map is None else torch.as_tensor(type_map, dtype=torch.int32, device=dev)
    rows_all = torch.arange(N, dtype=torch.int32, device=dev)
    act = batch.act
    mode_word = {HL.DEFENDER: torch.full((N,), S.MODE_DEFENDER, dtype=torch.int32, device=dev),
                 HL.ATTACKER: torch.full((N,), S.MODE_ATTACKER, dtype=torch.int32, device=dev)}
    rec = {k: [] for k in ("state", "action_vec", "reward", "raw_reward", "next_state", "done")}
    batch.prime_view(role)
    state = batch.role_obs[role].clone()
    t, sigma = int(t0), float(noise_std)
    while len(rec["done"]) < n_decisions:
        turn = HL.DEFENDER if t % 2 == 0 else HL.ATTACKER
        if turn != role and bl_code >= 0:
            cur_bl = bl_code
        act["mode"].copy_(mode_word[turn])
        if cur_bl is not None:
            act["mode"] |= (cur_bl + 1) << S.MODE_BASELINE_SHIFT
        act["n_groups"].zero_()
        if turn == role:
            vec = actor(state).float()
            if sigma > 0.0:
                vec = vec + torch.randn(vec.shape, generator=generator, device=dev, dtype=torch.float32) * sigma
            if clip is not None:
                vec = vec.clamp(clip[0], clip[1])
            sigma = max(float(sigma_min), sigma * float(decay_rate))
            vec = vec.contiguous()
            batch.decode_actions(None, vec, n_types, n_exploits, n_apps, tm, act)
            _, raw, shaped, done = batch.step(act, view=role, full_obs=False)
            nxt = batch.role_obs[role].clone()
            rec["state"].append(state); rec["action_vec"].append(vec); rec["reward"].append(shaped.clone()); rec["raw_reward"].append(raw.clone())
            rec["next_state"].append(nxt); rec["done"].append(done != 0)
            state = nxt
        else:
            if hasattr(opp, "write"):
                opp.write(batch, act, rows_all, batch.observe(1 if other == HL.DEFENDER else 2))
            else:
                a = opp(batch.observe(1 if other == HL.DEFENDER else 2), t if getattr(opp, "uses_global_tick", False) else t // 2, M, L)
                batch.write_actions(rows_all, a, act)
            batch.step(act, view=role, full_obs=False)
            state = batch.role_obs[role].clone()
        t += 1
    st = {k: torch.stack(v) for k, v in rec.items()}
    return Transitions(noise_std=sigma, **st)
