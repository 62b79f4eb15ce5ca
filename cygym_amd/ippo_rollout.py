"""Batched rollout collection for the reference's per-device actor-critic agents -- the data-collection loop of
`IPPOCommBestResponse.train` / MAPPO (IPPO.py:503-640, MAPPO.py:503-640) for every env of a batch at once, all tensors on
the device:

    turn = "defender" if env.step_num % 2 == 0 else "attacker"                       (:507)
    our turn:   v = build_visibility_mask(env, role); out = net(state, adj)           (:511-519)
                one Categorical per visible device over the role's action types, one for the exploit, one for the app,
                logp = sum of their log-probabilities                                 (:524-555)
                groups = per-type device lists, single-device types keep one device   (:560-571)
                env.step(groups); Step(state, logp, value, reward, done, ...)         (:574-600)
    their turn: the opponent's action, env.step                                       (:602-606)
    done:       fresh env + randomize_compromise_and_ownership + counters zeroed      (:613-624)

Here: the visibility mask is a tensor op on the flag plane (BatchedCyberDefenseEnv.visibility_mask), the net is evaluated
for all envs in one forward, the Categoricals are sampled as one batched draw, the grouping is ONE launch
(cygym_group_actions) and the tick one more (which also writes the next actor's role view).  The envs of a batch tick in
lock step from a common step_num (episodes end at the step cap only, CyberDefenseEnv.py:547-552), so the turn is the same
for every env; the cap's auto-reset reloads the snapshot the batch was created with, and the ownership reshuffle follows.

`gae` is compute_gae (IPPO.py:301-310) for [T, N] tensors.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import host_logic as HL
from . import spec as S

DEFENDER_NOOP, ATTACKER_NOOP = 8, 3          # IPPO.py:25-26
SINGLE_DEVICE_TYPES = (11, 12)               # IPPO.py:27
REWARD_SCALE = 1e-1                          # IPPO.py:31


@dataclass
class Rollout:
    """What `local_batch` holds (IPPO.py:588-599), stacked: [T, N, ...] device tensors, T = decisions of the role."""
    state: torch.Tensor          # [T, N, W] role observation at the decision
    logp: torch.Tensor           # [T, N]
    value: torch.Tensor          # [T, N]
    reward: torch.Tensor         # [T, N] shaped reward, clipped to +-1e6 (:581)
    raw_reward: torch.Tensor     # [T, N] float64
    done: torch.Tensor           # [T, N] bool
    per_dev_types: torch.Tensor  # [T, N, M] int64 (0 where invisible)
    exp: torch.Tensor            # [T, N] int64
    app: torch.Tensor            # [T, N] int64
    vis_mask: torch.Tensor       # [T, N, M] float32
    last_state: torch.Tensor     # [N, W] the role's view of the state the loop ended in (bootstrap value, :626-632)
    last_vis: torch.Tensor       # [N, M]


def gae(rewards: torch.Tensor, values: torch.Tensor, dones: torch.Tensor, gamma: float = 0.99, lam: float = 0.95):
    """compute_gae (IPPO.py:301-310) along dim 0 for every env at once: rewards, dones [T, N]; values [T + 1, N].
    Returns (advantages, returns) [T, N] float32."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards, dtype=torch.float32)
    last = torch.zeros_like(rewards[0], dtype=torch.float32)
    nonterm = 1.0 - dones.to(torch.float32)
    for t in range(T - 1, -1, -1):
        delta = rewards[t] + gamma * values[t + 1] * nonterm[t] - values[t]
        last = delta + gamma * lam * nonterm[t] * last
        adv[t] = last
    return adv, adv + values[:-1]


def masked_adjacency(vis: torch.Tensor) -> torch.Tensor:
    """What the reference's GAT layers receive, for a batch: build_adjacency(env, D) is all ones for the reference's Subnet (it
    has neither `edges()` nor a fallback other than np.ones, IPPO.py:52-72), and masked_adjacency (IPPO.py:98-110) turns it
    into v v^T with the diagonal set to v.  vis [N, M] in {0, 1} -> [N, M, M] float32."""
    v = vis.to(torch.float32)
    out = v[:, :, None] * v[:, None, :]
    eye = torch.eye(v.shape[1], device=v.device, dtype=torch.float32)[None] * v[:, :, None]
    return out * (1 - eye) + eye


def _opt(t):
    return None if t is None else t.float().contiguous()


def _sample(logits: torch.Tensor, greedy: bool, generator):
    """One Categorical per row of `logits` [..., K]: (sample, log-probability of the sample)."""
    logp_all = torch.log_softmax(logits.float(), dim=-1)
    if greedy:
        idx = torch.argmax(logp_all, dim=-1)
    else:
        flat = logp_all.reshape(-1, logp_all.shape[-1]).exp()
        idx = torch.multinomial(flat, 1, generator=generator).reshape(logp_all.shape[:-1])
    return idx, torch.gather(logp_all, -1, idx.unsqueeze(-1)).squeeze(-1)


@torch.no_grad()
def collect(batch, role: str, net, opponent, n_decisions: int, *, greedy: bool = False, generator=None, n_types: int | None = None,
            randomize_on_reset: bool = True, max_ticks: int | None = None, fused_sampling: bool = True) -> Rollout:
    """Collect `n_decisions` decisions of `role` in every env of `batch` (the while-loop of IPPO.py:503-611).

    net(state [N, W], vis [N, M]) -> dict with "per_dev_type_logits" [N, M, K], "value" [N] (or [N, 1]), optional
        "exp_logits" [N, E], "app_logits" [N, A] -- the outputs the reference's networks produce (:517-519, :541-555); how
        the net uses the mask (a GAT over masked_adjacency(vis), an MLP ...) is its own business.
    opponent: what plays the other role -- a baseline name / fixed sequence (rollout_grid.SequencePolicy semantics), a
        policy(obs, t, M, L) -> action tensors, or an object with write(batch, act, rows, obs) (policies.ActorPolicy).
    The batch must have been created with max_groups >= the role's action types and max_devs >= M, and auto_reset on.
    fused_sampling (default): the Categoricals are sampled inside the grouping launch (cygym_sample_group_actions: addressed
    Philox draws, log-probabilities summed in the kernel); False: torch.multinomial with `generator`, then cygym_group_actions.
    """
    from .rollout_grid import SequencePolicy, _baseline_code
    if role not in (HL.DEFENDER, HL.ATTACKER):
        raise ValueError("role must be 'attacker' or 'defender'")
    other = HL.ATTACKER if role == HL.DEFENDER else HL.DEFENDER
    N, M, L, dev = batch.N, batch.M, batch.L, batch.device
    noop = DEFENDER_NOOP if role == HL.DEFENDER else ATTACKER_NOOP
    opp = opponent if (callable(opponent) or hasattr(opponent, "write")) else SequencePolicy(opponent, other)
    # a baseline opponent sets env.base_line on its turn and nobody resets it (IPPO.py:395-397): from then on EVERY tick runs
    # under that baseline (volt_typhoon_env.py:847-874, :913-914) -- carried in the mode word like the reference's loops do
    bl_code = _baseline_code(opponent, other)
    cur_bl = None
    step_num = batch.state["ienv"][:, S.I_STEP_NUM]
    s0 = int(step_num[0].item())
    if not bool((step_num == s0).all().item()):
        raise ValueError("the envs of the batch must share their step_num (they tick in lock step)")
    cap = int(batch.cfg.episode_limit)                                # done iff step_num > cap (CyberDefenseEnv.py:547-552)
    if not batch.cfg.auto_reset:
        raise ValueError("create the batch with auto_reset=1: a done env starts over (IPPO.py:613-624)")
    rows_all = torch.arange(N, dtype=torch.int32, device=dev)
    act = batch.act
    rec = {k: [] for k in ("state", "logp", "value", "reward", "raw_reward", "done", "per_dev_types", "exp", "app", "vis_mask")}
    mode_word = {HL.DEFENDER: torch.full((N,), S.MODE_DEFENDER, dtype=torch.int32, device=dev),
                 HL.ATTACKER: torch.full((N,), S.MODE_ATTACKER, dtype=torch.int32, device=dev)}
    s, ticks = s0, 0
    turn = HL.DEFENDER if s % 2 == 0 else HL.ATTACKER
    batch.prime_view(turn)
    limit = max_ticks if max_ticks is not None else 4 * n_decisions + 8
    while len(rec["logp"]) < n_decisions and ticks < limit:
        turn = HL.DEFENDER if s % 2 == 0 else HL.ATTACKER
        obs = batch.role_obs[turn]
        if turn != role and bl_code >= 0:
            cur_bl = bl_code
        act["mode"].copy_(mode_word[turn])
        if cur_bl is not None:
            act["mode"] |= (cur_bl + 1) << S.MODE_BASELINE_SHIFT
        if turn == role:
            vis = batch.visibility_mask(role)
            out = net(obs, vis)
            pdt = out["per_dev_type_logits"]
            K = int(pdt.shape[-1]) if n_types is None else int(n_types)
            if fused_sampling and K == int(pdt.shape[-1]) <= 32:
                # sampling, log-probabilities and grouping in ONE launch (addressed Philox draws instead of torch's generator)
                t8, e32, a32, logp = batch.sample_group_actions(None, pdt.float().contiguous(), _opt(out.get("exp_logits")), _opt(out.get("app_logits")),
                                                                role, noop=noop, single_types=SINGLE_DEVICE_TYPES, greedy=greedy, act=act)
                types, exp_i, app_i = t8.to(torch.int64), e32.to(torch.int64), a32.to(torch.int64)
            else:
                types, lp = _sample(pdt, greedy, generator)                       # [N, M]
                visb = vis > 0.5
                types = torch.where(visb, types.clamp(0, K - 1), torch.zeros_like(types))   # invisible: in-range dummy label (:531-533)
                logp = (lp * visb).sum(dim=1)
                if out.get("exp_logits") is not None and out["exp_logits"].shape[-1] > 0:
                    exp_i, lpe = _sample(out["exp_logits"], greedy, generator)
                    logp = logp + lpe
                else:
                    exp_i = torch.zeros(N, dtype=torch.int64, device=dev)
                if out.get("app_logits") is not None and out["app_logits"].shape[-1] > 0:
                    app_i, lpa = _sample(out["app_logits"], greedy, generator)
                    logp = logp + lpa
                else:
                    app_i = torch.zeros(N, dtype=torch.int64, device=dev)
                batch.group_actions(None, types, exp_i, app_i, role, n_types=K, noop=noop, single_types=SINGLE_DEVICE_TYPES, act=act)
            state_rec = obs.clone()
        else:
            act["n_groups"].zero_()
            if hasattr(opp, "write"):
                opp.write(batch, act, rows_all, obs)
            else:
                a = opp(obs, s if getattr(opp, "uses_global_tick", False) else s // 2, M, L)
                batch.write_actions(rows_all, a, act)
        nxt = HL.DEFENDER if (s + 1) % 2 == 0 else HL.ATTACKER
        if s + 1 > cap:            # this tick reports done: every env reloads its snapshot (step_num 0: a defender turn)
            nxt = HL.DEFENDER
        _, raw, shaped, done = batch.step(act, view=nxt, full_obs=False)
        if turn == role:
            rec["state"].append(state_rec); rec["logp"].append(logp); rec["value"].append(out["value"].reshape(N).float())
            rec["reward"].append(torch.where(torch.isfinite(shaped), shaped, raw.nan_to_num(0.0, 0.0, 0.0)).to(torch.float32).clamp(-1e6, 1e6)); rec["raw_reward"].append(raw.clone())   # (IPPO.py:575-581: a non-finite shaped reward falls back to nan_to_num(raw))
            rec["done"].append(done != 0); rec["per_dev_types"].append(types); rec["exp"].append(exp_i); rec["app"].append(app_i)
            rec["vis_mask"].append(vis)
        s += 1
        ticks += 1
        if s > cap:
            s = 0
            if randomize_on_reset:
                batch.randomize()                                           # :615-616
                batch.prime_view(HL.DEFENDER)                               # (the reshuffle changed the state the view was written from)
    if not rec["logp"]:
        raise RuntimeError("no decision of the role within the tick limit")
    last_state = batch.observe(1 if role == HL.DEFENDER else 2)
    st = {k: torch.stack(v) for k, v in rec.items()}
    return Rollout(last_state=last_state, last_vis=batch.visibility_mask(role), **st)
