"""The DDPG best-response side of the drop-in boundary (do_agent.py:1334-1460): the batched transition collector
(cygym_amd/ddpg_rollout.py) against the reference loop run on the CPU oracle."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from cygym_amd import abi, host_logic as HL, spec as S  # noqa: E402


@pytest.mark.gpu
@pytest.mark.parametrize("role", ["defender", "attacker"])
def test_collect_equals_the_reference_loop_on_the_oracle(role):
    """ddpg_rollout.collect without exploration noise (integer-weight actor: exact on CPU and GPU) against the loop of
    do_agent.py:1334-1460 on the oracle: turn = t % 2, actor -> decode_action -> env.step on the learner's turns, a fixed
    sequence on the opponent's; states, action vectors, both rewards, next states, dones and the final state; then the
    exploration schedule and the clipping with noise on."""
    from grid_util import IntActor
    from oracle import driver as od
    import golden_io as gio
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.ddpg_rollout import collect
    from cygym_amd.policies import ActorPolicy
    from cygym_amd.topology import make_topology
    M, N, n_dec = 64, 20, 11
    topo, init, ck = make_topology(M, 4, seed=8, n_active=56)
    cfg = abi.EnvConfig(seed=8, **ck)
    X = cfg.max_exploits
    types = [1, 4, 5, 6, 7, 8, 9, 13, 2, 12, 11] if role == "defender" else [1, 2, 3]
    n_apps = 4 if role == "defender" else 0
    W = 6 * M if role == "defender" else 4 * M + X
    other = "attacker" if role == "defender" else "defender"
    opp_seq = [(1, [0], [], 0), (2, [1], [], 0), (3, [0], [], 0)] if other == "attacker" else [(1, [0], [3, 9, 12], 0), (8, [0], [], 0), (6, [0], [1, 2], 0)]
    actor = IntActor(W, len(types) + M + X + n_apps, 21)
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M)
    tr = collect(batch, role, actor.to("cuda:0"), opp_seq, n_dec, len(types), X, n_apps, type_map=types, clip=None)
    assert tr.state.shape == (n_dec, N, W) and tr.action_vec.shape[2] == len(types) + M + X + n_apps

    ob = od.OracleBatch(topo, cfg, N)
    ob.load_state(init)
    act = od.alloc_actions(N, 1, M)
    pol = ActorPolicy(IntActor(W, len(types) + M + X + n_apps, 21), len(types), X, n_apps, type_map=types)
    code = 1 if role == "defender" else 2
    t, k = 0, 0
    state = ob.observe(code)
    while k < n_dec:
        turn = "defender" if t % 2 == 0 else "attacker"
        act["exploit"][:] = -1
        if turn == role:
            a = pol(torch.from_numpy(state), t // 2, M, M)
            np.testing.assert_array_equal(tr.state[k].cpu().numpy(), state, err_msg=f"state at decision {k}")
            np.testing.assert_array_equal(tr.action_vec[k].cpu().numpy(), pol.net(torch.from_numpy(state)).numpy())
            for e in range(N):
                devs = np.nonzero(a["dev_mask"][e].numpy())[0].tolist()
                HL.encode_into(act, e, role, [(int(a["atype"][e]), [int(a["exploit"][e])], devs, int(a["app"][e]))], False, M)
            _, raw, shaped, done = ob.step(act)
            nxt = ob.observe(code)
            np.testing.assert_allclose(tr.raw_reward[k].cpu().numpy(), raw, rtol=0, atol=1e-9)
            np.testing.assert_allclose(tr.reward[k].cpu().numpy(), shaped, rtol=0, atol=1e-9)
            np.testing.assert_array_equal(tr.next_state[k].cpu().numpy(), nxt)
            np.testing.assert_array_equal(tr.done[k].cpu().numpy(), done != 0)
            state = nxt
            k += 1
        else:
            for e in range(N):
                HL.encode_into(act, e, turn, [opp_seq[t % len(opp_seq)]], False, M)
            ob.step(act)
            state = ob.observe(code)
        t += 1
    got = batch.state_numpy()
    got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(got, ob.state, f"ddpg collect {role}")
    # exploration: noise added, clipped to [-1, 1], the schedule decays to its floor
    g = torch.Generator(device="cuda:0").manual_seed(1)
    small = lambda x: actor.to("cuda:0")(x) * 2.0 ** -14      # noqa: E731  (inside the clip range most of the time)
    tr2 = collect(batch, role, small, opp_seq, 6, len(types), X, n_apps, type_map=types, noise_std=0.5, sigma_min=0.2, decay_rate=0.5, generator=g)
    assert tr2.noise_std == 0.2 and float(tr2.action_vec.abs().max()) <= 1.0
    assert float((tr2.action_vec[0] - small(tr2.state[0])).abs().mean()) > 0.05
    batch.close()
