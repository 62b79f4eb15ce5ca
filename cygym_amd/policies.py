"""Closed-loop strategies for the batched rollout consumer (cygym_amd/rollout_grid.simulate_grid).

`ActorPolicy` is the batched form of branch (D) of the reference's rollout loop (do_agent.py:240-262): a parametric
actor maps the role observation to an action vector [type logits | device values | exploit values | app values] and
`DoubleOracle.decode_action` (do_agent.py:935-998) turns it into the action tuple.  Here the actor is any torch
module evaluated on all the cells that play the strategy at once, and the decoding + scatter into the batch's
action tensors is ONE launch of the library (cygym_decode_actions).  `reference_actor` builds the reference's own
architecture (do_agent.py:357-370); `mlp_actor` a smaller one.
"""
from __future__ import annotations

import torch
from torch import nn


class ActorPolicy:
    tick_free = True      # the action does not depend on the tick number: the loop may be captured in a HIP graph

    def __init__(self, net: nn.Module, n_types: int, n_exploits: int, n_apps: int = 0, type_map=None, epsilon: float = 0.0):
        self.net, self.n_types, self.n_exploits, self.n_apps = net, int(n_types), int(n_exploits), int(n_apps)
        self.epsilon = float(epsilon)      # epsilon-greedy action type (do_agent.py:972-973), fused path only
        self.fuse_head = True              # run the last Linear layer inside the decode launch when it fits
        self.fuse_mlp = True               # ... and the whole network when it is a plain Linear-ReLU stack (cygym_actor_mlp_decode)
        self.from_state = True             # ... which then builds the role view on chip from the batch's state (no view tensor)
        self.type_map = None if type_map is None else torch.as_tensor(type_map, dtype=torch.int32)
        # what the policy can emit (simulate_grid asks: action 10 needs a detector batch)
        self.action_types = list(range(self.n_types)) if type_map is None else sorted({int(x) for x in self.type_map.tolist()})

    def _map(self, device):
        if self.type_map is not None and self.type_map.device != device:
            self.type_map = self.type_map.to(device)
        return self.type_map

    def _split_head(self, M, max_out=512):
        """(body modules, last nn.Linear, tanh?) when the actor is a Sequential ending in Linear [+ Tanh] that the fused
        head kernel can take (cygym_actor_head_decode: H <= 256, <= 512 outputs), else None.  Cached per output limit."""
        cache = self.__dict__.setdefault("_heads", {})
        max_out = (int(M), max_out, id(self.net))   # (the split depends on the device count and on WHICH net: a policy may serve several batches)
        if max_out not in cache:
            cache[max_out] = None
            if isinstance(self.net, nn.Sequential) and len(self.net) >= 2:
                mods = list(self.net)
                tanh = isinstance(mods[-1], nn.Tanh)
                last = mods[-2] if tanh else mods[-1]
                if isinstance(last, nn.Linear) and last.in_features <= 256 and last.out_features <= max_out[1] \
                        and last.out_features == self.n_types + M + self.n_exploits + self.n_apps and last.weight.dtype == torch.float32:
                    cache[max_out] = (type(self.net)(*mods[: -2 if tanh else -1]), last, tanh)
        return cache[max_out]

    def _split_mlp(self, M):
        """(hidden nn.Linear layers, last nn.Linear, tanh?) when the WHOLE actor is a Linear-ReLU stack the fused actor kernel
        can take (cygym_actor_mlp_decode: 1 to 3 hidden layers, widths multiples of 16 up to 256, <= 8192 outputs -- vectors wider
        than 512 are decoded in chunks), else None."""
        cache = self.__dict__.setdefault("_mlps", {})
        key = (int(M), id(self.net))   # (per device count and net: a policy may serve batches of different sizes, or get a new net)
        if key not in cache:
            cache[key] = None
            head = self._split_head(M, max_out=8192)
            if head is not None:
                body, last, tanh = head
                mods = list(body)
                lins = mods[0::2]
                if (len(mods) % 2 == 0 and 1 <= len(lins) <= 3 and all(isinstance(m, nn.ReLU) for m in mods[1::2])
                        and all(isinstance(m, nn.Linear) and m.weight.dtype == torch.float32 and m.out_features % 16 == 0
                                and 16 <= m.out_features <= 256 for m in lins)):
                    cache[key] = (lins, last, tanh)
        return cache[key]

    def _packed(self, batch, M):
        """Fragment-ordered copies of the actor's weights (batch.pack_linear), redone when a parameter changes."""
        lins, last, tanh = self._split_mlp(M)
        ver = (int(M),) + tuple((m.weight._version, m.weight.data_ptr(), None if m.bias is None else m.bias._version) for m in lins + [last])
        if getattr(self, "_pk_ver", None) != ver:
            self._pk = ([(batch.pack_linear(m.weight), None if m.bias is None else m.bias.detach().contiguous(), m.out_features) for m in lins],
                        (batch.pack_linear(last.weight, 64), None if last.bias is None else last.bias.detach().contiguous()))
            self._pk_ver = ver
        return self._pk

    def fused_mlp(self, batch) -> bool:
        return bool(self.fuse_head and self.fuse_mlp and hasattr(batch, "actor_mlp_decode") and self._split_mlp(batch.M) is not None)

    def reads_state(self, batch) -> bool:
        """May the fused actor build its observation on chip from the batch's state (cygym_actor_mlp.obs_role) instead of
        reading a role-view tensor?  (Then the tick need not write that view.)"""
        return self.fused_mlp(batch) and self.from_state and batch.M % 2 == 0

    @torch.no_grad()
    def write_by_env(self, batch, act, rows, obs_all, role=None, step=None):
        """The whole actor + decode + scatter in ONE launch (cygym_actor_mlp_decode), for the envs `rows`: the observation is
        built on chip from the batch's state when `role` is given and reads_state(batch), else read in place from the batch's
        role view `obs_all` [N, K]; only when fused_mlp(batch).  step = {...}: a tick runs first, in the same launch
        (BatchedCyberDefenseEnv.actor_mlp_decode)."""
        hidden, head = self._packed(batch, batch.M)
        from_state = role is not None and self.reads_state(batch)
        batch.actor_mlp_decode(rows, None if from_state else obs_all, hidden, head, self.n_types, self.n_exploits, self.n_apps,
                               self._map(batch.device), act, epsilon=self.epsilon, tanh=self._split_mlp(batch.M)[2], obs_by_env=True,
                               obs_role=role if from_state else None, step=step)

    def n_out(self, M):
        return self.n_types + M + self.n_exploits + self.n_apps

    @torch.no_grad()
    def write(self, batch, act, rows, obs):
        """Fused path: when the actor is a plain Linear-ReLU stack, ONE launch for the network, the decode and the scatter
        into rows `rows` of the action tensors (cygym_actor_mlp_decode); otherwise the actor's body in torch, then one
        decode-and-scatter launch that also runs the last Linear layer when it has at most 512 outputs."""
        if self.fused_mlp(batch) and obs.dtype == torch.float32 and obs.dim() == 2 and obs.stride(1) == 1:
            hidden, head_p = self._packed(batch, batch.M)
            batch.actor_mlp_decode(rows, obs, hidden, head_p, self.n_types, self.n_exploits, self.n_apps, self._map(obs.device), act,
                                   epsilon=self.epsilon, tanh=self._split_mlp(batch.M)[2])
            return
        head = self._split_head(batch.M) if (self.fuse_head and hasattr(batch, "actor_head_decode")) else None
        if head is not None:
            body, last, tanh = head
            ver = (last.weight._version, last.weight.data_ptr())
            if getattr(self, "_wt_ver", None) != ver:          # k-major copy of the layer's weights, redone when they change
                self._wt, self._wt_ver = batch.head_weights(last.weight), ver
            batch.actor_head_decode(rows, body(obs), self._wt, last.bias, self.n_types, self.n_exploits, self.n_apps,
                                    self._map(obs.device), act, epsilon=self.epsilon, tanh=tanh)
            return
        batch.decode_actions(rows, self.net(obs), self.n_types, self.n_exploits, self.n_apps, self._map(obs.device), act,
                             epsilon=self.epsilon)

    @torch.no_grad()
    def __call__(self, obs, t, M, L):
        """The same decoding with torch ops (batch-likes without cygym_decode_actions: the tests' oracle harness)."""
        if self.epsilon > 0.0:
            raise NotImplementedError("epsilon-greedy types are drawn in cygym_decode_actions (needs the envs' rng ticks)")
        v = self.net(obs)
        k = self.n_types
        at = torch.argmax(v[:, :k], dim=1).to(torch.int32) if k > 0 else torch.zeros(v.shape[0], dtype=torch.int32, device=v.device)
        tm = self._map(obs.device)
        if tm is not None:
            at = tm[at.long()]
        ex = torch.argmax(v[:, k + M: k + M + self.n_exploits], dim=1) if self.n_exploits > 0 else torch.zeros_like(at)
        app = torch.argmax(v[:, k + M + self.n_exploits: k + M + self.n_exploits + self.n_apps], dim=1) if self.n_apps > 0 else torch.zeros_like(at)
        return {"atype": at, "exploit": ex.to(torch.int32), "dev_mask": v[:, k: k + M] > 0, "app": app.to(torch.int32)}


class ActorPolicyGroup:
    """A population of ActorPolicy strategies of ONE architecture (same layer shapes, activations, decode layout, type map
    and epsilon) evaluated together: every hidden layer is one batched GEMM over the stacked weights (torch.baddbmm) and the
    last layer + decode + scatter of all of them is ONE launch (cygym_actor_head_decode with n_groups).  The cost of a tick
    then does not grow with the number of strategies of a grid -- the |D| x |A| grids of a Double-Oracle population share
    their architecture.  Built by simulate_grid when it applies; rows arrive ordered by strategy, equally many (a multiple of
    16) per strategy."""

    tick_free = True

    def __init__(self, policies):
        self.policies = list(policies)
        p0 = self.policies[0]
        self.n_types, self.n_exploits, self.n_apps, self.epsilon = p0.n_types, p0.n_exploits, p0.n_apps, p0.epsilon
        self.action_types = sorted({t for p in self.policies for t in p.action_types})
        self._cache = None

    @staticmethod
    def key(p, M):
        """Hashable architecture signature of an ActorPolicy that the group forward can run, or None."""
        if not isinstance(p, ActorPolicy) or not p.fuse_head:
            return None
        split = p._split_head(M)
        if split is None and p.fuse_mlp and p._split_mlp(M) is not None:      # more than 512 outputs: the whole-actor launch only
            split = p._split_head(M, max_out=8192)
        if split is None:
            return None
        body, last, tanh = split
        sig = []
        for m in body:
            if isinstance(m, nn.Linear):
                if m.bias is None or m.weight.dtype != torch.float32:
                    return None
                sig.append(("L", m.in_features, m.out_features))
            elif isinstance(m, nn.ReLU):
                sig.append(("R",))
            else:
                return None
        tm = None if p.type_map is None else tuple(int(x) for x in p.type_map.tolist())
        return (tuple(sig), last.in_features, last.out_features, bool(tanh), p.n_types, p.n_exploits, p.n_apps, tm, p.epsilon)

    def _stacked(self, batch):
        heads = [p._split_head(batch.M) for p in self.policies]
        mods = [[m for m in h[0] if isinstance(m, nn.Linear)] + [h[1]] for h in heads]
        ver = tuple((m.weight._version, m.weight.data_ptr(), m.bias._version) for ms in mods for m in ms)
        if self._cache is None or self._cache[0] != ver:
            n_lin = len(mods[0]) - 1
            Ws = [torch.stack([ms[l].weight.detach().t() for ms in mods]).contiguous() for l in range(n_lin)]      # [S, in, out]
            bs = [torch.stack([ms[l].bias.detach() for ms in mods])[:, None, :].contiguous() for l in range(n_lin)]  # [S, 1, out]
            Wh = torch.stack([batch.head_weights(ms[-1].weight) for ms in mods]).contiguous()                       # [S, H, pitch]
            bh = torch.stack([ms[-1].bias.detach() for ms in mods]).contiguous()                                      # [S, n_out]
            self._cache = (ver, Ws, bs, Wh, bh)
        return self._cache[1:]

    def fused_mlp(self, batch) -> bool:
        return all(p.fused_mlp(batch) for p in self.policies)

    def _packed_all(self, batch):
        packs = [p._packed(batch, batch.M) for p in self.policies]
        ver = tuple(p._pk_ver for p in self.policies)
        if getattr(self, "_pk_ver", None) != ver:
            n_h = len(packs[0][0])
            cat = lambda ts: None if ts[0] is None else torch.cat([t.reshape(-1) for t in ts]).contiguous()  # noqa: E731
            hidden = [(cat([pk[0][l][0] for pk in packs]), cat([pk[0][l][1] for pk in packs]), packs[0][0][l][2]) for l in range(n_h)]
            head = (cat([pk[1][0] for pk in packs]), cat([pk[1][1] for pk in packs]))
            self._pk, self._pk_ver = (hidden, head), ver
        return self._pk

    def reads_state(self, batch) -> bool:
        return all(p.reads_state(batch) for p in self.policies)

    @torch.no_grad()
    def write_by_env(self, batch, act, rows, obs_all, role=None, step=None, rows_per_group=None):
        """All the actors of the population, whole networks + decode + scatter, in ONE launch (observations as in
        ActorPolicy.write_by_env).  rows_per_group: rows in ENV order (rows = None), env e playing actor (e // rows_per_group) %
        len(policies) -- the grid layouts of rollout_grid; default: rows ordered actor after actor."""
        hidden, head = self._packed_all(batch)
        p0 = self.policies[0]
        from_state = role is not None and self.reads_state(batch)
        batch.actor_mlp_decode(rows, None if from_state else obs_all, hidden, head, self.n_types, self.n_exploits, self.n_apps,
                               p0._map(batch.device), act, epsilon=self.epsilon, tanh=p0._split_mlp(batch.M)[2],
                               n_groups=len(self.policies), obs_by_env=True, obs_role=role if from_state else None, step=step,
                               rows_per_group=rows_per_group)

    def n_out(self, M):
        return self.policies[0].n_out(M)

    @torch.no_grad()
    def write(self, batch, act, rows, obs):
        S = len(self.policies)
        if self.fused_mlp(batch) and obs.dtype == torch.float32 and obs.dim() == 2 and obs.stride(1) == 1:
            hidden, head_p = self._packed_all(batch)
            p0 = self.policies[0]
            batch.actor_mlp_decode(rows, obs, hidden, head_p, self.n_types, self.n_exploits, self.n_apps, p0._map(obs.device), act,
                                   epsilon=self.epsilon, tanh=p0._split_mlp(batch.M)[2], n_groups=S)
            return
        if self.policies[0]._split_head(batch.M) is None:
            raise ValueError("a population with more than 512 outputs runs through the whole-actor launch only (float32 observations)")
        body, last, tanh = self.policies[0]._split_head(batch.M)
        Ws, bs, Wh, bh = self._stacked(batch)
        x = obs.reshape(S, obs.shape[0] // S, obs.shape[1])
        l = 0
        for m in body:
            if isinstance(m, nn.Linear):
                x = torch.baddbmm(bs[l], x, Ws[l])
                l += 1
            else:
                x = torch.relu_(x)
        hidden = x.reshape(obs.shape[0], -1)
        batch.actor_head_decode(rows, hidden, Wh, bh, self.n_types, self.n_exploits, self.n_apps, self.policies[0]._map(obs.device), act,
                                epsilon=self.epsilon, tanh=tanh, n_groups=S)


class FusedMLP(nn.Sequential):
    """nn.Sequential of Linear / ReLU / Tanh whose Linear + ReLU pairs run as ONE GEMM with a ReLU epilogue
    (torch._addmm_activation) on 2-D inputs -- one launch less per hidden layer of a closed-loop tick."""

    def forward(self, x):
        mods = list(self)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Linear) and x.dim() == 2:
                if i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU) and hasattr(torch, "_addmm_activation"):
                    x = torch._addmm_activation(m.bias, x, m.weight.t())
                    i += 2
                    continue
                x = torch.addmm(m.bias, x, m.weight.t())
            else:
                x = m(x)
            i += 1
        return x


def mlp_actor(state_dim: int, action_dim: int, hidden=(64,), seed: int = 0, device="cpu", tanh: bool = False) -> nn.Module:
    """Linear-ReLU stack ending in a linear layer of `action_dim` outputs (tanh on top like the reference's actor when
    asked); default-initialised from `seed`."""
    g = torch.Generator().manual_seed(int(seed))
    layers, d = [], int(state_dim)
    for h in hidden:
        layers += [nn.Linear(d, int(h)), nn.ReLU()]
        d = int(h)
    layers.append(nn.Linear(d, int(action_dim)))
    if tanh:
        layers.append(nn.Tanh())
    net = FusedMLP(*layers)
    with torch.no_grad():
        for p in net.parameters():      # same distribution as nn.Linear's default (uniform +- 1/sqrt(fan_in)), seeded
            bound = 1.0 / (p.shape[-1] ** 0.5) if p.dim() > 1 else 1.0 / (state_dim ** 0.5)
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * bound)
    return net.to(device).eval()


def reference_actor(state_dim: int, action_dim: int, seed: int = 0, device="cpu") -> nn.Module:
    """The reference's DDPG actor (do_agent.py:357-370): state -> 256 -> 256 -> action_dim, ReLU, tanh."""
    return mlp_actor(state_dim, action_dim, hidden=(256, 256), seed=seed, device=device, tanh=True)


@torch.no_grad()
def calibrate_device_head(policy: ActorPolicy, obs: torch.Tensor, M: int, fraction: float):
    """Shift the bias of the actor's device outputs so that on `obs` a fraction `fraction` of the device values is
    positive, i.e. the policy lists about fraction * M devices per action.  A freshly initialised actor selects every
    second device (its outputs are symmetric around 0); trained policies act on a handful -- benchmarks of the closed
    loop calibrate their random actors to the list lengths they want to measure."""
    last = [m for m in policy.net.modules() if isinstance(m, nn.Linear)][-1]
    k = policy.n_types
    v = policy.net(obs)[:, k: k + M]
    q = torch.quantile(v.flatten().float()[: 1 << 22], 1.0 - float(fraction))
    last.bias[k: k + M] -= q
