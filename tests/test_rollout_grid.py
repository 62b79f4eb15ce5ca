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


from grid_util import IntPolicy, OracleGrid  # noqa: E402


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


def test_closed_loop_grid_trains_and_scans_like_the_oracle_loop():
    """Policies that emit defender action 10 (Detector.train) and 5 (scan): the loop services the trainings after the
    tick that asked (the reference trains inside the tick, volt_typhoon_env.py:961), later scans walk the forests,
    and payoffs / final state equal the oracle loop that fits with scikit-learn on its own history ring."""
    pytest.importorskip("sklearn")
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.rollout_grid import simulate_grid
    topo, init, cfg = _setup()
    M, T, n_mc, L = topo.M, 60, 3, 16
    D = [IntPolicy("defender", M, [10, 5, 5, 1, 10, 5], 11), IntPolicy("defender", M, [5, 10, 6, 9, 5, 13, 8], 12),
         [(10, [0], [], 0), (8, [0], [], 0), (5, [0], [3, 9, 17], 0)]]
    A = [IntPolicy("attacker", M, [1, 1, 2], 13), [(1, [0], [], 0)]]
    N = len(D) * len(A) * n_mc
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=L, detector=True)
    U_def, U_att = simulate_grid(batch, D, A, n_mc, T, randomize=True)
    og = OracleGrid(topo, cfg, N, init, 1, L, detector=True)
    E_def, E_att = simulate_grid(og, D, A, n_mc, T, randomize=True)
    assert og.fitted > N, "the strategies must actually train (and retrain) their detectors"
    np.testing.assert_allclose(U_def, E_def, rtol=0, atol=1e-9)
    np.testing.assert_allclose(U_att, E_att, rtol=0, atol=1e-9)
    got = batch.state_numpy()
    got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(got, og.ob.state, "simulate_grid + detector")
    np.testing.assert_array_equal(got["forest"], og.ob.state["forest"])
    fl = got["ienv"][:, S.I_FLAGS]
    assert (fl & S.E_DET_TRAIN).sum() > N // 2 and not (fl & (S.E_UNPINNED | S.E_DET_PENDING)).any()
    assert int(got["ienv"][:, S.I_SCAN_CNT].sum()) > 0
    batch.close()
    # the same strategies on a batch that cannot fit forests are refused up front, not answered with all-"D" scans
    plain = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=L)
    with pytest.raises(ValueError, match="detector=True"):
        simulate_grid(plain, D, A, n_mc, 4)
    plain.close()


def test_open_loop_script_with_trainings_is_cut_or_refused():
    """payoff_grid / rollout() with defender action 10 followed by scans: on a detector batch the fused launch is cut
    after every training tick and serviced (== the oracle loop); without detector buffers it raises."""
    pytest.importorskip("sklearn")
    from cygym_amd import _lib
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.rollout_grid import payoff_grid, simulate_grid
    topo, init, cfg = _setup()
    T, n_mc = 30, 4
    D = [[(10, [0], [], 0), (5, [0], [3, 9, 17], 0), (5, [0], [1], 0), (1, [0], [2, 4], 0), (5, [0], [7, 8], 0)], "No Defense"]
    A = [[(1, [0], [], 0)], [(1, [1], [], 0), (2, [0], [], 0)]]
    N = len(D) * len(A) * n_mc
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=8, detector=True)
    U_def, U_att = payoff_grid(batch, D, A, n_mc, T, randomize=True)
    og = OracleGrid(topo, cfg, N, init, 1, 8, detector=True)
    E_def, E_att = simulate_grid(og, D, A, n_mc, T, randomize=True)
    assert og.fitted >= 2 * n_mc
    np.testing.assert_allclose(U_def, E_def, rtol=0, atol=1e-9)
    np.testing.assert_allclose(U_att, E_att, rtol=0, atol=1e-9)
    got = batch.state_numpy()
    got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(got, og.ob.state, "payoff_grid + detector")
    np.testing.assert_array_equal(got["forest"], og.ob.state["forest"])
    batch.close()
    plain = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=8)
    with pytest.raises(_lib.CygymError, match="without a current forest"):
        payoff_grid(plain, D, A, n_mc, T, randomize=True)
    plain.close()


def test_baseline_names_persist_for_the_other_role():
    """env.base_line is assigned by a baseline strategy before its step and stays for the other role's turns
    (do_agent.py:218-221): against a "No Attack" attacker a scripted defender is a no-op from its second turn on
    (volt_typhoon_env.py:913-914).  Checked on the schedule itself and against the oracle loop."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.rollout_grid import baseline_schedule, payoff_grid, simulate_grid
    D = [[(7, [0], [5, 6], 0), (1, [0], [3], 0)], "No Defense"]
    A = ["No Attack", [(1, [0], [], 0)]]
    sch = baseline_schedule(D, A, np.arange(4), 1, 5, 0)
    np.testing.assert_array_equal(sch[:, 0], [0, 3, 3, 3, 3])      # scripted defender x "No Attack"
    np.testing.assert_array_equal(sch[:, 1], [0, 0, 0, 0, 0])      # scripted x scripted: stays "Nash"
    np.testing.assert_array_equal(sch[:, 2], [1, 3, 1, 3, 1])      # "No Defense" x "No Attack": each sets its own
    np.testing.assert_array_equal(sch[:, 3], [1, 1, 1, 1, 1])
    topo, init, cfg = _setup()
    T, n_mc = 12, 2
    N = len(D) * len(A) * n_mc
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=8)
    U_def, U_att = payoff_grid(batch, D, A, n_mc, T, randomize=False)
    flags = batch.state_numpy()["flags"]
    # cell (0, 0): the removal of devices 5, 6 happens at t = 0 only (base_line still "Nash"); the clean of device 3 at
    # t = 2 is already a no-op; cell (0, 1) keeps executing its script
    og = OracleGrid(topo, cfg, N, init, 1, 8)
    E_def, E_att = simulate_grid(og, D, A, n_mc, T, randomize=False)
    np.testing.assert_allclose(U_def, E_def, rtol=0, atol=1e-9)
    np.testing.assert_allclose(U_att, E_att, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(flags, og.ob.state["flags"])
    C_def, C_att = simulate_grid(batch, D, A, n_mc, T, randomize=False)       # the closed-loop path agrees
    np.testing.assert_allclose(C_def, E_def, rtol=0, atol=1e-9)
    np.testing.assert_allclose(C_att, E_att, rtol=0, atol=1e-9)
    batch.close()


def test_fused_role_views_status_word_and_action_scatter():
    """cygym_step's optional role views equal cygym_observe after the tick; obs == NULL leaves the full observation
    alone; the status word reports pending trainings; cygym_write_actions equals the per-field torch scatter."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.rollout_grid import mask_to_list
    for M, blocks, N in ((64, 4, 200), (256, 1, 130), (37, 2, 64)):
        from cygym_amd.topology import make_topology
        topo, init, ck = make_topology(M, blocks, seed=4, n_active=max(8, M - 8))
        cfg = abi.EnvConfig(seed=4, **ck)
        L = max(1, M // 8)
        env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=L, detector=(M == 64))
        for t in range(30):
            env.gen_actions(t)
            role = "attacker" if t % 2 == 0 else "defender"     # the NEXT actor
            keep = env.obs.clone()
            env.step(view=role, full_obs=(t % 3 != 0))
            np.testing.assert_array_equal(env.role_obs[role].cpu().numpy(), env.observe(1 if role == "defender" else 2).cpu().numpy(),
                                          err_msg=f"M={M} t={t} {role} view")
            if t % 3 == 0:
                assert torch.equal(env.obs, keep), "obs == NULL must not be written"
        if M == 64:     # action 10 on a non-empty log raises DET_PENDING in the status word until serviced
            env.take_status()
            env.act["mode"].fill_(S.MODE_DEFENDER); env.act["atype"].fill_(10); env.act["dev_cnt"].zero_()
            env.step()
            assert env.take_status() & S.E_DET_PENDING
            assert env.service_detectors() > 0
            env.act["atype"].fill_(8)
            env.step()
            assert not (env.take_status() & S.E_DET_PENDING)
        # fused scatter vs torch
        g = torch.Generator(device="cpu").manual_seed(M)
        rows = torch.randperm(N, generator=g)[: N // 2].sort().values.to("cuda:0")
        n = rows.numel()
        a = {"atype": torch.randint(0, 14, (n,), generator=g).to("cuda:0", torch.int32),
             "exploit": (torch.randint(0, 4, (n,), generator=g) - 1).to("cuda:0", torch.int32),
             "app": torch.randint(0, 4, (n,), generator=g).to("cuda:0", torch.int32),
             "dev_mask": (torch.rand((n, M), generator=g) < 0.2).to("cuda:0")}
        before = {k: v.clone() for k, v in env.act.items()}
        env.write_actions(rows, a)
        idx, cnt = mask_to_list(a["dev_mask"], L)
        exp = before
        exp["atype"][rows, 0] = a["atype"]; exp["exploit"][rows, 0, 0] = a["exploit"]; exp["n_exploit"][rows, 0] = (a["exploit"] >= 0).to(torch.int32)
        exp["app"][rows, 0] = a["app"]; exp["dev_cnt"][rows, 0] = cnt; exp["dev_idx"][rows] = idx
        for k in exp:
            assert torch.equal(env.act[k], exp[k]), f"M={M} write_actions {k}"
        b = {"atype": a["atype"], "exploit": a["exploit"], "app": a["app"], "dev_idx": idx.flip(1).contiguous(), "dev_cnt": cnt}
        env.write_actions(rows, b)
        want = idx.flip(1)
        want = torch.where(torch.arange(L, device=want.device)[None, :] < cnt[:, None], want, torch.zeros_like(want))
        assert torch.equal(env.act["dev_idx"][rows], want)
        env.close()


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
