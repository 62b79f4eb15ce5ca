"""The batched rollout consumer (cygym_amd/rollout_grid.py) against the CPU oracle.

`payoff_grid` (open loop, one fused launch) and `simulate_grid` (closed loop: observation -> policy -> action every
tick, all on the device) must give the payoff matrices the same strategies earn on the ORACLE, where the very same
policy code runs on the oracle's observations on the CPU -- the reference-shaped loop of do_agent.py:206-272 with the
oracle standing in for the reference env.  The policies are integer-weight networks: every intermediate value is a
small integer, exact in float32, so CPU and GPU evaluate them to identical actions."""
import numpy as np
import pytest

import golden_io as gio
from cygym_amd import abi
from cygym_amd import spec as S

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


class OracleGrid:
    """The oracle behind the few members simulate_grid / payoff_grid use of a batch (CPU tensors over its arrays)."""

    def __init__(self, topo, cfg, N, init, G, L):
        from oracle import driver as od
        self.ob = od.OracleBatch(topo, cfg, N)
        self.init = init
        self.N, self.M, self.L = N, topo.M, L
        self.act_np = od.alloc_actions(N, G, L)
        self.act = {k: torch.from_numpy(v) for k, v in self.act_np.items()}   # shared memory
        self.obs = torch.zeros(1)
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


class IntPolicy:
    """Closed-loop test policy: integer weights in {-1, 0, 1}, ReLU, argmax with an index tie-break -- all values are
    integers far below 2^24, so float32 arithmetic is exact on every device."""

    def __init__(self, role, M, types, seed):
        rs = np.random.RandomState(seed)
        self.role, self.M, self.types = role, M, list(types)
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


def _setup(M=64, n_mc=3):
    from cygym_amd.topology import make_topology
    topo, init, ck = make_topology(M, 4, seed=8, n_active=56)
    cfg = abi.EnvConfig(seed=8, **ck)
    return topo, init, cfg


def test_payoff_grid_equals_the_oracle_loop():
    """Open loop (baselines and fixed sequences), one cygym_rollout launch, against the oracle stepped tick by tick."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.rollout_grid import payoff_grid, simulate_grid
    topo, init, cfg = _setup()
    T, n_mc = 24, 3
    D = ["No Defense", [(1, [0], [3, 9, 12], 0), (7, [0], [5], 0), (6, [0], [1, 2, 3, 4], 0)], [(13, [0], [7], 0)]]
    A = ["No Attack", [(1, [0], [], 0)], [(2, [0], [], 0), (1, [1], [], 0)]]
    N = len(D) * len(A) * n_mc
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=8)
    U_def, U_att = payoff_grid(batch, D, A, n_mc, T, randomize=True)
    og = OracleGrid(topo, cfg, N, init, 1, 8)
    E_def, E_att = simulate_grid(og, D, A, n_mc, T, randomize=True)       # same strategies, oracle, tick by tick
    np.testing.assert_allclose(U_def, E_def, rtol=0, atol=1e-9)
    np.testing.assert_allclose(U_att, E_att, rtol=0, atol=1e-9)
    got = batch.state_numpy()
    got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(got, og.ob.state, "payoff_grid")
    assert (U_def[:, 0] >= U_def[:, 1]).all(), "an idle attacker can only help the defender"
    batch.close()


def test_closed_loop_grid_equals_the_oracle_loop():
    """Closed loop: observation -> integer-weight policy -> action every tick, on the device, no host round trip;
    the same policies on the oracle's observations give the expected payoffs and final state."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.rollout_grid import simulate_grid
    topo, init, cfg = _setup()
    M, T, n_mc, L = topo.M, 40, 4, 16
    D = [IntPolicy("defender", M, [1, 4, 5, 6, 7, 8, 9, 13, 2], 1), IntPolicy("defender", M, [1, 6, 9, 12, 11, 3], 2), "No Defense"]
    A = [IntPolicy("attacker", M, [1, 2, 3], 3), IntPolicy("attacker", M, [1, 1, 2], 4), [(1, [0], [], 0)]]
    N = len(D) * len(A) * n_mc
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=L)
    U_def, U_att = simulate_grid(batch, D, A, n_mc, T, randomize=True)
    og = OracleGrid(topo, cfg, N, init, 1, L)
    E_def, E_att = simulate_grid(og, D, A, n_mc, T, randomize=True)
    np.testing.assert_allclose(U_def, E_def, rtol=0, atol=1e-9)
    np.testing.assert_allclose(U_att, E_att, rtol=0, atol=1e-9)
    got = batch.state_numpy()
    got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(got, og.ob.state, "simulate_grid")
    for k in ("atype", "dev_cnt", "dev_idx", "exploit"):      # the last tick's actions were the same on both sides
        np.testing.assert_array_equal(batch.act[k].cpu().numpy(), og.act_np[k], err_msg=k)
    assert len(np.unique(np.round(U_def, 6))) > 3, "the strategies must actually differ in payoff"
    batch.close()


def test_view_step_cost_does_not_grow_with_the_batch():
    """CyberDefenseEnvView.step launches only its own env (cygym_step_range) and writes only its own action row:
    the other envs of the batch neither tick nor have their action rows touched."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.env_view import CyberDefenseEnvView
    topo, init, cfg = _setup()
    batch = BatchedCyberDefenseEnv(topo, cfg, 512, init, device="cuda:0", max_groups=2, max_devs=16)
    batch.act["atype"].fill_(77)
    before = batch.state_numpy()
    env = CyberDefenseEnvView(batch, 300)
    env.mode = "attacker"
    env.step((1, [0], [], 0))
    env.mode = "defender"
    env.step([(1, [0], [3, 4], 0), (2, [0], [], 0)])
    after = batch.state_numpy()
    others = np.arange(512) != 300
    for k in ("live", "ienv", "fenv", "ring"):
        np.testing.assert_array_equal(before[k][others], after[k][others], err_msg=k)
    assert after["ienv"][300, S.I_STEP_NUM] == before["ienv"][300, S.I_STEP_NUM] + 2
    at = batch.act["atype"].cpu().numpy()
    assert (at[others] == 77).all() and at[300, 0] == 1 and at[300, 1] == 2
    batch.close()
