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
    batch.close()
    # the closed-loop path agrees (a fresh batch: reset() keeps an env's draw counter running, like a new episode)
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=8)
    C_def, C_att = simulate_grid(batch, D, A, n_mc, T, randomize=False)
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


def test_actor_policies_fused_decode_equals_the_oracle_loop_eager_and_graph():
    """cygym_amd.policies.ActorPolicy: actor forward + ONE fused decode-and-scatter launch (cygym_decode_actions =
    do_agent.decode_action for a batch), role views and episode returns from the tick kernel -- eager and replayed from
    a HIP graph -- against the same actors decoded with torch ops on the oracle loop."""
    from grid_util import IntActor
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.policies import ActorPolicy
    from cygym_amd.rollout_grid import simulate_grid
    topo, init, cfg = _setup()
    M, X, T, n_mc = topo.M, cfg.max_exploits, 41, 5
    def_types = [1, 4, 5, 6, 7, 8, 9, 13, 2, 12, 11, 3]
    def make(dev):
        D = [ActorPolicy(IntActor(6 * M, len(def_types) + M + X + 4, 21).to(dev), len(def_types), X, 4, type_map=def_types),
             ActorPolicy(IntActor(6 * M, len(def_types) + M + X + 4, 22).to(dev), len(def_types), X, 4, type_map=def_types), "No Defense"]
        A = [ActorPolicy(IntActor(4 * M + X, 4 + M + X, 23).to(dev), 4, X, 0), ActorPolicy(IntActor(4 * M + X, 3 + M + X, 24).to(dev), 3, X, 0)]
        return D, A
    N = 3 * 2 * n_mc
    og = OracleGrid(topo, cfg, N, init, 1, M)
    E_def, E_att = simulate_grid(og, *make("cpu"), n_mc, T, randomize=True)
    assert len(np.unique(np.round(E_def, 6))) > 3
    for graph, streams in ((False, 1), (True, 1), (False, 3), (True, 2)):
        batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M)
        U_def, U_att = simulate_grid(batch, *make("cuda:0"), n_mc, T, randomize=True, graph=graph, streams=streams)
        np.testing.assert_allclose(U_def, E_def, rtol=0, atol=1e-9, err_msg=f"graph={graph} streams={streams}")
        np.testing.assert_allclose(U_att, E_att, rtol=0, atol=1e-9, err_msg=f"graph={graph} streams={streams}")
        got = batch.state_numpy()
        got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
        assert not gio.compare_state(got, og.ob.state, f"actor grid graph={graph}")
        for k in ("atype", "dev_cnt", "dev_idx", "exploit", "app"):      # the last tick's decoded actions
            np.testing.assert_array_equal(batch.act[k].cpu().numpy(), og.act_np[k], err_msg=f"{k} graph={graph}")
        batch.close()
    # a list capacity the policies overflow is reported, not silently cut
    small = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=2)
    with pytest.raises(RuntimeError, match="max_devs"):
        simulate_grid(small, *make("cuda:0"), n_mc, 6, randomize=True)
    small.close()


def test_decode_actions_matches_numpy_including_epsilon_greedy():
    """cygym_decode_actions against do_agent.decode_action restated with numpy (:970-998): argmax type through the type
    map, ascending ids of positive device values, argmax exploit / app; with epsilon the coin and the uniform index of
    the addressed Philox draw (env, the env's rng tick, CG_SITE_EPS_TYPE)."""
    from cygym_amd import rng as R
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    M, N = 100, 77
    topo, init, ck = make_topology(M, 2, seed=6, n_active=90)
    cfg = abi.EnvConfig(seed=0x1234567890, env_id_base=1000, **ck)
    env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M)
    for t in range(3):
        env.gen_actions(t); env.step()                 # rng ticks move on
    X, n_types, n_apps = cfg.max_exploits, 9, 5
    tm = np.array([1, 4, 5, 6, 7, 8, 9, 13, 2], np.int32)
    g = torch.Generator().manual_seed(3)
    vec = torch.randn((N, n_types + M + X + n_apps + 3), generator=g)           # (stride wider than the layout)
    rows = torch.randperm(N, generator=g)[:50].sort().values
    v = vec[:50].numpy()
    ticks = env.state["ienv"][:, S.I_RNG_TICK].cpu().numpy()
    for eps in (0.0, 0.35, 1.0):
        env.act["atype"].fill_(-7)
        env.decode_actions(rows.to("cuda:0"), vec[:50].to("cuda:0"), n_types, X, n_apps, torch.from_numpy(tm).to("cuda:0"), epsilon=eps)
        at = np.argmax(v[:, :n_types], axis=1)
        if eps > 0:
            c = R.philox4x32_10_np(cfg.env_id_base + rows.numpy(), ticks[rows.numpy()], S.SITE_EPS_TYPE, 0, cfg.seed & 0xFFFFFFFF, cfg.seed >> 32)
            coin = c[0] < R.bernoulli_threshold(eps)
            at = np.where(coin, ((c[1].astype(np.uint64) * np.uint64(n_types)) >> np.uint64(32)).astype(np.int64), at)
            assert eps == 1.0 or (0 < coin.sum() < 50)
        got = {k: t.cpu().numpy() for k, t in env.act.items()}
        np.testing.assert_array_equal(got["atype"][rows.numpy(), 0], tm[at], err_msg=f"eps={eps}")
        assert (got["atype"][np.setdiff1d(np.arange(N), rows.numpy()), 0] == -7).all()
        for i, r in enumerate(rows.numpy()):
            ids = np.nonzero(v[i, n_types:n_types + M] > 0)[0]
            assert got["dev_cnt"][r, 0] == len(ids) and (got["dev_idx"][r, :len(ids)] == ids).all() and (got["dev_idx"][r, len(ids):] == 0).all()
            assert got["exploit"][r, 0, 0] == np.argmax(v[i, n_types + M:n_types + M + X]) and got["n_exploit"][r, 0] == 1
            assert got["app"][r, 0] == np.argmax(v[i, n_types + M + X:n_types + M + X + n_apps])
    env.close()


def test_actor_head_kernel_equals_linear_plus_decode():
    """cygym_actor_head_decode (last Linear layer + decode in one launch, action vectors in registers) against
    nn.Linear followed by cygym_decode_actions: exact on integer-valued weights (H = 64 and a two-slab H = 160), and on
    float weights with tanh wherever the decision is not a near-tie."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    for M, H, n_types, n_apps in ((256, 64, 11, 4), (64, 160, 3, 0), (37, 32, 14, 7), (64, 30, 5, 2)):   # (H % 4 != 0: the scalar variant)
        topo, init, ck = make_topology(M, 1 if M != 64 else 4, seed=2, n_active=max(8, M - 8))
        cfg = abi.EnvConfig(seed=2, **ck)
        N, X = 203, cfg.max_exploits
        env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M)
        other = {k: v.clone() for k, v in env.act.items()}
        n_out = n_types + M + X + n_apps
        g = torch.Generator().manual_seed(M)
        hidden = torch.randint(0, 4, (N, H + 5), generator=g).float().to("cuda:0")[:, :H]          # (row stride > H)
        W = (torch.randint(-1, 2, (n_out, H), generator=g) * 16).float().to("cuda:0")
        b = (torch.arange(n_out) - n_out // 3).float().to("cuda:0")
        rows = torch.randperm(N, generator=g)[:150].sort().values.to("cuda:0")
        tm = torch.arange(n_types, dtype=torch.int32, device="cuda:0") + 1
        env.actor_head_decode(rows, hidden[:150], env.head_weights(W), b, n_types, X, n_apps, tm, epsilon=0.3)
        env.decode_actions(rows, torch.addmm(b, hidden[:150], W.t()), n_types, X, n_apps, tm, act=other, epsilon=0.3)
        for k in other:
            assert torch.equal(env.act[k], other[k]), f"M={M} H={H}: {k}"
        # float weights + tanh: compare with torch on the rows whose decisions are clear
        Wf = torch.randn((n_out, H), generator=g).to("cuda:0") * 0.05
        bf = torch.randn((n_out,), generator=g).to("cuda:0") * 0.1
        env.actor_head_decode(None, hidden, env.head_weights(Wf), bf, n_types, X, n_apps, None, tanh=True)
        v = torch.tanh(torch.addmm(bf.double(), hidden.double(), Wf.double().t()).float())
        top2 = torch.topk(v[:, :n_types], 2, dim=1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-4
        assert clear.sum() > N // 2
        assert torch.equal(env.act["atype"][clear, 0], torch.argmax(v[:, :n_types], dim=1).to(torch.int32)[clear])
        dv = v[:, n_types:n_types + M]
        sure = (dv.abs() > 1e-4).all(dim=1)
        assert torch.equal(env.act["dev_cnt"][sure, 0], (dv > 0).sum(dim=1).to(torch.int32)[sure]) and sure.sum() > N // 2
        env.close()


def test_actor_mlp_kernel_equals_torch_forward_plus_decode():
    """cygym_actor_mlp_decode (the whole Linear-ReLU stack + last layer + decode in ONE launch, matrix cores, observation
    tile through LDS) against the torch forward followed by cygym_decode_actions: exact on integer-valued weights -- one to
    three hidden layers, observation widths that are one stage, several stages, several tiles and not a multiple of 16,
    16-, 8- and 4-byte aligned rows (the three copy variants), rows read in place by env id, and a population of actors."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    dev = "cuda:0"
    #       M   widths          K     stride  n_types n_apps  by_env  groups
    cases = ((256, (64,),         1536, 1536,   11,     4,      True,   1),
             (256, (64,),         1030, 1030,   3,      0,      True,   1),      # attacker view: 8-byte aligned rows
             (64,  (256, 256),    384,  388,    14,     2,      False,  1),
             (37,  (48, 32, 16),  229,  229,    5,      7,      False,  1),      # odd width and stride: dword copies; 3 tiles of 16
             (64,  (128,),        3400, 3400,   4,      0,      False,  1),      # three observation tiles, the last one partial
             (256, (64, 32),      1536, 1536,   9,      3,      True,   3),
             (600, (32,),         520,  520,    5,      7,      False,  1),      # 614 outputs: decoded in two chunks of 512
             (1100, (64, 16),     300,  300,    12,     4,      True,   1),      # 1118 outputs: three chunks, the app values in the last
             (600, (32,),         256,  256,    70,     3,      False,  2))      # more than 64 action types (type map from memory), a population
    for M, widths, K, stride, n_types, n_apps, by_env, S_ in cases:
        topo, init, ck = make_topology(M, 1 if M != 64 else 4, seed=2, n_active=max(8, M - 8))
        cfg = abi.EnvConfig(seed=2, **ck)
        N, X = 208, cfg.max_exploits
        env = BatchedCyberDefenseEnv(topo, cfg, N, init, device=dev, max_groups=1, max_devs=M)
        other = {k: v.clone() for k, v in env.act.items()}
        n_out = n_types + M + X + n_apps
        g = torch.Generator().manual_seed(M + K)
        obs = torch.randint(-1, 3, (N, stride), generator=g).float().to(dev)[:, :K]
        n = 48 * S_ if S_ > 1 else 150
        rows = torch.randperm(N, generator=g)[:n].sort().values.to(dev)
        tm = torch.arange(n_types, dtype=torch.int32, device=dev) + 1
        actors = []
        for a in range(S_):
            Ws, bs, d = [], [], K
            for w in widths:
                Ws.append(torch.randint(-1, 2, (w, d), generator=g).float().to(dev))
                bs.append(torch.randint(-3, 4, (w,), generator=g).float().to(dev))
                d = w
            Wh = (torch.randint(-1, 2, (n_out, d), generator=g) * 16).float().to(dev)
            bh = (torch.arange(n_out) - n_out // 3 + a).float().to(dev)
            actors.append((Ws, bs, Wh, bh))
        cat = lambda ts: torch.cat([t.reshape(-1) for t in ts]).contiguous()  # noqa: E731
        hidden = [(cat([env.pack_linear(a[0][l]) for a in actors]), cat([a[1][l] for a in actors]), widths[l]) for l in range(len(widths))]
        head = (cat([env.pack_linear(a[2], 64) for a in actors]), cat([a[3] for a in actors]))
        src = obs if by_env else obs.index_select(0, rows.long())
        env.actor_mlp_decode(rows, src, hidden, head, n_types, X, n_apps, tm, epsilon=0.3, n_groups=S_, obs_by_env=by_env)
        x_all = obs.index_select(0, rows.long())
        vecs = []
        for a, (Ws, bs, Wh, bh) in enumerate(actors):
            x = x_all[a * (n // S_):(a + 1) * (n // S_)]
            for W, b in zip(Ws, bs):
                x = torch.relu(torch.addmm(b, x, W.t()))
            vecs.append(torch.addmm(bh, x, Wh.t()))
        env.decode_actions(rows, torch.cat(vecs), n_types, X, n_apps, tm, act=other, epsilon=0.3)
        for k in other:
            assert torch.equal(env.act[k], other[k]), f"M={M} widths={widths} K={K}: {k}"
        assert int(env.act["dev_cnt"][rows.long(), 0].max()) > 0
        # float weights + tanh: compare with torch (float64) on the rows whose decisions are clear
        if S_ == 1:
            Ws, bs, d = [], [], K
            for w in widths:
                Ws.append(torch.randn((w, d), generator=g).to(dev) / d ** 0.5)
                bs.append(torch.randn((w,), generator=g).to(dev) * 0.1)
                d = w
            Wf = torch.randn((n_out, d), generator=g).to(dev) / d ** 0.5
            bf = torch.randn((n_out,), generator=g).to(dev) * 0.1
            hidden = [(env.pack_linear(W), b, w) for W, b, w in zip(Ws, bs, widths)]
            env.actor_mlp_decode(None, obs.contiguous() if not by_env else obs, hidden, (env.pack_linear(Wf, 64), bf), n_types, X, n_apps, None, tanh=True)
            x = obs.double()
            for W, b in zip(Ws, bs):
                x = torch.relu(torch.addmm(b.double(), x, W.double().t()))
            v = torch.tanh(torch.addmm(bf.double(), x, Wf.double().t()))
            top2 = torch.topk(v[:, :n_types], 2, dim=1).values
            clear = (top2[:, 0] - top2[:, 1]) > 1e-4
            assert clear.sum() > N // 2
            assert torch.equal(env.act["atype"][clear, 0], torch.argmax(v[:, :n_types], dim=1).to(torch.int32)[clear])
            dv = v[:, n_types:n_types + M]
            sure = (dv.abs() > 1e-4).all(dim=1)
            assert torch.equal(env.act["dev_cnt"][sure, 0], (dv > 0).sum(dim=1).to(torch.int32)[sure]) and sure.sum() > N // 4
        env.close()


def test_actor_mlp_builds_the_role_view_on_chip_from_the_state():
    """cygym_actor_mlp.obs_role: the fused actor builds the defender / attacker view of its 16 envs in LDS from the flag plane
    and the static columns instead of reading a view tensor -- same actions as the same actor run on cygym_observe's view
    (integer weights: exact), at 64 / 256 devices (one tile), 400 (two tiles, the second one partial), 600 and 2048 (action vectors wider than 512: decoded in chunks), also with rows
    given by env id and with the per-env anomaly plane of the slow-scan mode."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    dev = "cuda:0"
    for M, N, slow in ((256, 203, False), (64, 96, True), (400, 40, False), (600, 24, False), (2048, 20, False)):
        topo, init, ck = make_topology(M, 1 if M != 64 else 4, seed=5, n_active=M - 6)
        cfg = abi.EnvConfig(seed=5, **({**ck, "fast_scan": 0} if slow else ck))
        env = BatchedCyberDefenseEnv(topo, cfg, N, init, device=dev, max_groups=1, max_devs=max(4, M // 8), **({"detector": True} if slow else {}))
        env.randomize()
        for t in range(6):
            env.gen_actions(t); env.step()              # flags, known / not-yet-added bits and (slow scan) anomaly scores move
        X = cfg.max_exploits
        g = torch.Generator().manual_seed(M)
        for role, code, n_types, n_apps in (("defender", 1, 12, 3), ("attacker", 2, 3, 0)):
            K = env.role_width(role)
            n_out = n_types + M + X + n_apps
            view = env.observe(code)
            assert view.shape == (N, K)
            W1 = torch.randint(-1, 2, (32, K), generator=g).float().to(dev)
            b1 = torch.randint(-3, 4, (32,), generator=g).float().to(dev)
            Wh = (torch.randint(-1, 2, (n_out, 32), generator=g) * 16).float().to(dev)
            bh = (torch.arange(n_out) - n_out // 3).float().to(dev)
            hidden, head = [(env.pack_linear(W1), b1, 32)], (env.pack_linear(Wh, 64), bh)
            rows = torch.randperm(N, generator=g)[: N - 7].sort().values.to(dev)
            a_view = {k: v.clone() for k, v in env.act.items()}
            a_state = {k: v.clone() for k, v in env.act.items()}
            env.actor_mlp_decode(rows, view, hidden, head, n_types, X, n_apps, None, act=a_view, epsilon=0.25, obs_by_env=True)
            env.actor_mlp_decode(rows, None, hidden, head, n_types, X, n_apps, None, act=a_state, epsilon=0.25, obs_role=role)
            for k in a_view:
                assert torch.equal(a_view[k], a_state[k]), f"M={M} {role}: {k}"
            assert int(a_state["dev_cnt"][rows.long(), 0].max()) > 0 and len(torch.unique(a_state["atype"][rows.long(), 0])) > 1
        env.close()


def test_actor_population_is_one_batched_forward_and_one_head_launch():
    """Strategies that are actor networks of one architecture are evaluated as a population (ActorPolicyGroup: ONE
    cygym_actor_mlp_decode launch with n_groups for the whole networks; with fuse_mlp off, batched GEMMs + ONE
    cygym_actor_head_decode launch) -- same payoffs and final state as the oracle loop, which runs every actor on its own with
    the torch decode."""
    from grid_util import int_mlp_actor
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.policies import ActorPolicy
    from cygym_amd.rollout_grid import simulate_grid
    topo, init, cfg = _setup()
    M, X, T, n_mc = topo.M, cfg.max_exploits, 30, 16
    def_types = [1, 4, 5, 6, 7, 8, 9, 13, 2, 12, 11, 3]
    def make(dev):
        D = [ActorPolicy(int_mlp_actor(6 * M, len(def_types) + M + X + 4, 16, 31 + i, dev), len(def_types), X, 4, type_map=def_types) for i in range(2)]
        A = [ActorPolicy(int_mlp_actor(4 * M + X, 3 + M + X, 16, 41 + j, dev), 3, X, 0, type_map=[1, 2, 3]) for j in range(2)]
        return D, A
    N = 2 * 2 * n_mc
    og = OracleGrid(topo, cfg, N, init, 1, M)
    E_def, E_att = simulate_grid(og, *make("cpu"), n_mc, T, randomize=True)
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M)
    calls = []
    orig = batch.actor_mlp_decode
    def spy(*a, **k):
        calls.append(k.get("n_groups", 1))
        return orig(*a, **k)
    batch.actor_mlp_decode = spy
    U_def, U_att = simulate_grid(batch, *make("cuda:0"), n_mc, T, randomize=True, graph=True)
    assert calls and all(g == 2 for g in calls), calls          # one actor launch per tick, both actors of the acting role in it
    np.testing.assert_allclose(U_def, E_def, rtol=0, atol=1e-9)
    np.testing.assert_allclose(U_att, E_att, rtol=0, atol=1e-9)
    got = batch.state_numpy()
    got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(got, og.ob.state, "actor population")
    assert len(np.unique(np.round(E_def, 6))) > 2
    # the same with the body in torch (batched GEMMs) and only the last layer in the decode launch
    batch.close()
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M)
    D, A = make("cuda:0")
    for p in D + A:
        p.fuse_mlp = False
    heads = []
    orig_h = batch.actor_head_decode
    batch.actor_head_decode = lambda *a, **k: (heads.append(k.get("n_groups", 1)), orig_h(*a, **k))[1]
    U_def, U_att = simulate_grid(batch, D, A, n_mc, T, randomize=True, graph=True)
    assert heads and all(g == 2 for g in heads), heads
    np.testing.assert_allclose(U_def, E_def, rtol=0, atol=1e-9)
    np.testing.assert_allclose(U_att, E_att, rtol=0, atol=1e-9)
    batch.close()


def test_tick_and_next_actor_as_one_launch_equals_two_launches():
    """cygym_step_actor (merge_launches: a tick and the next role's whole-actor launch as ONE kernel, the actor reading the flag
    planes the tick left in LDS) against the two-launch loop on the same strategies -- float weights, epsilon-greedy action
    types (the addressed Philox draw reads the rng tick the tick has just written back), a 2 x 2 population grid in env order
    and a 1 x 1 grid, eager and graph replay: identical payoffs and final state, and the merged entry point really ran."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.policies import ActorPolicy, mlp_actor
    from cygym_amd.rollout_grid import simulate_grid
    from cygym_amd.topology import make_topology
    M = 256
    topo, init, ck = make_topology(M, 1, seed=11, max_extra=0)
    X = abi.EnvConfig(seed=11, **ck).max_exploits
    dt, at = [1, 4, 5, 6, 7, 8, 9, 11, 12, 13, 2], [1, 2, 3]
    # (the last case: episodes end inside the run -- the tick reloads the snapshot into LDS and the actor behind it must see that)
    for (nD, nA, n_mc), graph, cap in (((2, 2, 16), True, 1000), ((1, 1, 48), False, 1000), ((3, 2, 32), False, 1000), ((1, 2, 16), False, 9)):
        N = nD * nA * n_mc
        cfg = abi.EnvConfig(seed=11, lambda_events=0.0, auto_reset=int(cap < 1000), episode_limit=cap, **ck)

        def make():
            Dp = [ActorPolicy(mlp_actor(6 * M, len(dt) + M + X + 4, (32,), seed=100 + i, device="cuda:0"), len(dt), X, 4, type_map=dt, epsilon=0.3) for i in range(nD)]
            Ap = [ActorPolicy(mlp_actor(4 * M + X, len(at) + M + X, (32,), seed=200 + j, device="cuda:0"), len(at), X, 0, type_map=at, epsilon=0.3) for j in range(nA)]
            return Dp, Ap
        res = {}
        for merge in (False, True):
            batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M)
            calls = []
            orig = batch.actor_mlp_decode
            batch.actor_mlp_decode = lambda *a, **k: (calls.append(k.get("step") is not None), orig(*a, **k))[1]
            U = simulate_grid(batch, *make(), n_mc, 31, randomize=False, graph=graph, merge_launches=merge)
            res[merge] = (U, batch.state_numpy(), batch.ret.cpu().numpy().copy())
            assert any(calls) == merge, (merge, calls[:8])
            if merge:
                assert sum(calls) >= (8 if graph else 29)          # every tick but the first actor and the last tick (graph: the eager part)
            batch.close()
        (Ua, sa, ra), (Ub, sb, rb) = res[False], res[True]
        np.testing.assert_array_equal(Ua[0], Ub[0]); np.testing.assert_array_equal(Ua[1], Ub[1])
        np.testing.assert_array_equal(ra, rb)
        assert not gio.compare_state(sb, sa, f"merged grid {nD}x{nA}x{n_mc}")
        assert np.abs(ra).sum() > 0


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
