"""Batched payoff grid (one fused launch) == the reference-shaped loop: one env per cell,
env.mode = ...; env.step(action) per tick (do_agent.py:206-272)."""
import numpy as np
import pytest

from cygym_amd import abi

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_payoff_grid_equals_stepwise_loops():
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.env_view import CyberDefenseEnvView
    from cygym_amd.rollout_grid import payoff_grid, _action_at
    from cygym_amd.topology import make_topology
    M, T, n_mc = 64, 24, 3
    topo, init, ck = make_topology(M, 4, seed=8, n_active=56)
    cfg = abi.EnvConfig(seed=8, **ck)
    D = ["No Defense", [(1, [0], [3, 9, 12], 0), (7, [0], [5], 0), (6, [0], [1, 2, 3, 4], 0)], [(13, [0], [7], 0)]]
    A = ["No Attack", [(1, [0], [], 0)], [(2, [0], [], 0), (1, [1], [], 0)]]
    N = len(D) * len(A) * n_mc
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=8)
    U_def, U_att = payoff_grid(batch, D, A, n_mc, T, randomize=True)
    # the same cells, stepped one env-tick at a time through the per-env view
    ref = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=8)
    ref.randomize()
    got = np.zeros((N, 2))
    for n in range(N):
        i, j = n // (len(A) * n_mc), (n // n_mc) % len(A)
        env = CyberDefenseEnvView(ref, n)
        for t in range(T):
            env.mode = "defender" if t % 2 == 0 else "attacker"
            a = _action_at(D[i] if t % 2 == 0 else A[j], t // 2, env.mode)
            _, r, _, done, info, _ = env.step(a)
            got[n, t % 2] += r
    exp = got.reshape(len(D), len(A), n_mc, 2).mean(axis=2)
    np.testing.assert_allclose(U_def, exp[..., 0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(U_att, exp[..., 1], rtol=0, atol=1e-9)
    assert (U_def[:, 0] >= U_def[:, 1]).all(), "an idle attacker can only help the defender"
    batch.close(); ref.close()
