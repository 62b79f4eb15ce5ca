"""Shared by the grid-consumer tests (CPU and GPU): the oracle behind the batch surface simulate_grid / payoff_grid
use, and an exact integer-weight closed-loop policy."""
import numpy as np
import torch

from cygym_amd import spec as S


class OracleGrid:
    """The oracle behind the members simulate_grid / payoff_grid use of a batch (CPU tensors over its arrays).  It has
    neither fused role views nor the fused action scatter: the loop then takes its observe() / torch fallbacks, so the
    product's fused paths are checked against the plain ones.  Detector.train is answered with scikit-learn on the
    oracle's own history ring (cygym_amd.detector.fit_forest) -- the reference's estimator, the request's seed."""

    def __init__(self, topo, cfg, N, init, G, L, detector=False):
        from oracle import driver as od
        self.ob = od.OracleBatch(topo, cfg, N, detector=detector)
        self.init, self.cfg, self.detector = init, cfg, detector
        self.N, self.M, self.L = N, topo.M, L
        self.act_np = od.alloc_actions(N, G, L)
        self.act = {k: torch.from_numpy(v) for k, v in self.act_np.items()}   # shared memory
        self.obs = torch.zeros(1)
        self.fitted = 0
        self.reset()

    def reset(self):
        self.ob.load_state(self.init)

    def randomize(self):
        self.ob.randomize()

    def observe(self, role):
        return torch.from_numpy(self.ob.observe(role))

    def step(self):
        obs, raw, shaped, done = self.ob.step(self.act_np)
        return torch.from_numpy(obs), torch.from_numpy(raw.copy()), torch.from_numpy(shaped.copy()), torch.from_numpy(done.copy())

    def alloc_rollout(self, T):
        act = {k: torch.zeros((T,) + tuple(v.shape), dtype=v.dtype) for k, v in self.act.items()}
        act["exploit"].fill_(-1); act["app"].fill_(-1)
        return act, {"raw": torch.zeros((T, self.N), dtype=torch.float64)}

    def rollout(self, act, out):
        """cygym_rollout's contract on the oracle: T ticks of a pre-staged script, trainings serviced after their tick."""
        for t in range(act["mode"].shape[0]):
            for k in self.act_np:
                self.act_np[k][...] = act[k][t].numpy()
            _, raw, _, _ = self.step()
            out["raw"][t] = raw
            if self.detector and (self.take_status() & S.E_DET_PENDING):
                self.service_detectors()
        return out

    def take_status(self):
        return int(np.bitwise_or.reduce(self.ob.state["ienv"][:, S.I_FLAGS]) & (S.E_TOPO_OVF | S.E_BUSY_SAT | S.E_DET_PENDING | S.E_UNPINNED))

    def unpinned_envs(self):
        return int(((self.ob.state["ienv"][:, S.I_FLAGS] & S.E_UNPINNED) != 0).sum())

    def service_detectors(self):
        from cygym_amd import detector as D
        st = self.ob.state
        for e in np.nonzero(st["ienv"][:, S.I_FLAGS] & S.E_DET_PENDING)[0]:
            hdr = st["forest"][e]
            rows = D.training_window(st["hist"][e], int(hdr[4]), bool(self.cfg.turbo), self.cfg.turbo_train_max_logs, self.cfg.turbo_train_stride)
            self.ob.install_forest(e, D.fit_forest(rows, D.fit_seed(self.cfg.seed, self.cfg.env_id_base + int(e), int(hdr[3])), n_fits=int(hdr[6])))
            self.fitted += 1


class IntPolicy:
    """Closed-loop test policy: integer weights in {-1, 0, 1}, ReLU, argmax with an index tie-break -- all values are
    integers far below 2^24, so float32 arithmetic is exact on every device."""

    def __init__(self, role, M, types, seed):
        rs = np.random.RandomState(seed)
        self.role, self.M, self.types = role, M, list(types)
        self.action_types = self.types
        self.F = 6 if role == "defender" else 4
        self.w_dev = torch.tensor(rs.randint(-1, 2, size=(self.F,)), dtype=torch.float32)           # per-device score
        self.w_hid = torch.tensor(rs.randint(-1, 2, size=(self.F * M, 8)), dtype=torch.float32)
        self.w_out = torch.tensor(rs.randint(-1, 2, size=(8, len(self.types))), dtype=torch.float32)
        self.mod = int(rs.randint(3, 8))

    def __call__(self, obs, t, M, L):
        dev = obs.device
        x = obs[:, : self.F * M]
        h = torch.relu(x @ self.w_hid.to(dev))
        logits = h @ self.w_out.to(dev)
        key = logits * 16 + torch.arange(len(self.types), device=dev, dtype=torch.float32)      # unique maximum
        atype = torch.tensor(self.types, dtype=torch.int32, device=dev)[torch.argmax(key, dim=1)]
        score = (x.reshape(-1, M, self.F) * self.w_dev.to(dev)).sum(dim=2) + (t % 5)
        mask = (torch.remainder(score, self.mod) == 0)
        if self.role == "defender":
            mask = mask & (x.reshape(-1, M, self.F)[:, :, 5] != 1)       # skip rows that say "not yet added"
        expl = torch.remainder(h.sum(dim=1), 3).to(torch.int32) - 1                               # -1, 0 or 1
        return {"atype": atype, "exploit": expl, "dev_mask": mask, "app": torch.remainder(h[:, 0], 4).to(torch.int32)}




class IntActor(torch.nn.Module):
    """An actor network (role observation -> action vector [type logits | device values | exploit values | app values],
    do_agent.py:1016-1020) with integer weights: every value is a small integer, exact in float32 on CPU and GPU, and
    `16 * z + position` makes every argmax unique -- so the fused decode, the torch decode and np.argmax agree."""

    def __init__(self, state_dim, action_dim, seed, hidden=8):
        super().__init__()
        rs = np.random.RandomState(seed)
        self.w1 = torch.nn.Parameter(torch.tensor(rs.randint(-1, 2, size=(state_dim, hidden)), dtype=torch.float32), requires_grad=False)
        self.w2 = torch.nn.Parameter(torch.tensor(rs.randint(-1, 2, size=(hidden, action_dim)), dtype=torch.float32), requires_grad=False)
        self.pos = torch.nn.Parameter(torch.arange(action_dim, dtype=torch.float32) - action_dim // 3, requires_grad=False)

    def forward(self, x):
        return (torch.relu(x @ self.w1) @ self.w2) * 16 + self.pos


def int_mlp_actor(state_dim, action_dim, hidden, seed, device="cpu"):
    """cygym_amd.policies.mlp_actor with integer weights (exact in float32 on every device and in every summation order)
    and a last-layer bias that makes every arg-max unique: the architecture the fused head / population paths take."""
    from cygym_amd.policies import mlp_actor
    net = mlp_actor(state_dim, action_dim, (hidden,), seed=seed)
    rs = np.random.RandomState(seed)
    lin = [m for m in net if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        lin[0].weight.copy_(torch.tensor(rs.randint(-1, 2, size=lin[0].weight.shape), dtype=torch.float32))
        lin[0].bias.copy_(torch.tensor(rs.randint(-2, 3, size=lin[0].bias.shape), dtype=torch.float32))
        lin[1].weight.copy_(torch.tensor(rs.randint(-1, 2, size=lin[1].weight.shape) * 16, dtype=torch.float32))
        lin[1].bias.copy_(torch.arange(action_dim, dtype=torch.float32) - action_dim // 3)
    return net.to(device)
