"""The per-env view (reference method surface) over the HIP batch, driven with the
reference-style Python actions stored in the golden fixtures -- the rollout-loop shape
`env.mode = ...; _, r, _, done, info, _ = env.step(action)` of do_agent.py:206-272."""
import numpy as np
import pytest

import golden_io as gio
from cygym_amd import spec as S

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

NAMES = [n for n in gio.fixture_names()
         if n.startswith(("s16_none", "s16_baselines", "s32_grouped", "s16_dups", "s16_mixed", "s16_zeroday", "s16_partial",
                          "s24_star", "s20_pa"))]


@pytest.mark.parametrize("name", NAMES)
def test_view_matches_reference(name):
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.env_view import CyberDefenseEnvView
    fx = gio.Fixture(name)
    T = min(fx.T, 120)
    batch = BatchedCyberDefenseEnv(fx.topo, fx.cfg, fx.N, fx.init, device="cuda:0", max_groups=fx.G, max_devs=fx.L)
    views = [CyberDefenseEnvView(batch, e) for e in range(fx.N)]
    alive = [True] * fx.N
    for t in range(T):
        for e, env in enumerate(views):     # envs tick one at a time, like separate reference objects
            if not alive[e]:
                continue
            env.mode = fx.mode_name(e, t)
            action = fx.python_action(e, t)
            partial = fx.is_partial(e, t)
            state, raw, shaped, done, info, logs = env.step(action, agent_cnt=fx.M + 1) if partial else env.step(action)
            if not fx.exp["topo_same"][e, t] and not fx.follows_topology():
                alive[e] = False
                continue
            exp_i = fx.exp["ienv"][e, t]
            np.testing.assert_array_equal(state, fx.exp["obs"][e, t].reshape(-1).astype(np.float64), err_msg=f"{name} e={e} t={t}")
            assert abs(raw - fx.exp["raw"][e, t]) < 1e-9 and abs(shaped - fx.exp["shaped"][e, t]) < 1e-9
            assert done == bool(fx.exp["done"][e, t])
            grouped = int(fx.z["act_n_groups"][e, t]) > 0
            assert info["step_count"] == int(exp_i[S.I_STEP_NUM]) - (0 if (grouped or partial) else 1)
            assert info["work_done"] == int(exp_i[S.I_WORK_DONE])
            assert info["Compromised_devices"] == int(exp_i[S.I_COMP_CNT])
            assert info["Scan_count"] == int(exp_i[S.I_SCAN_CNT])
            assert info["Edges Blocked"] == int(exp_i[S.I_EDGES_BLOCKED]) and info["Edges Added"] == int(exp_i[S.I_EDGES_ADDED])
            assert abs(info["defensive_cost"] - fx.exp["fenv"][e, t][S.D_DEF_COST]) < 1e-9
            if not grouped:
                assert info["executed_atype"] == int(exp_i[S.I_LAST_ATYPE])
            assert len(logs) == min(int(exp_i[S.I_LOG_TOTAL]), S.LOG_RING)
        # the other envs really stayed put while one ticked
        got = batch.state_numpy()
        for e in range(fx.N):
            if alive[e]:
                np.testing.assert_array_equal(got["flags"][e], fx.exp["flags"][e, t].astype(np.uint8))
    # role views and attribute surface
    env = views[0]
    assert env._get_defender_state().shape == (6 * fx.M,) and env._get_attacker_state().shape == (4 * fx.M + 6,)
    assert env.step_num == int(batch.state["ienv"][0, S.I_STEP_NUM].item())
    env.work_done = 0
    assert env.work_done == 0
    batch.close()


def test_view_errors_and_reset():
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.env_view import CyberDefenseEnvView
    fx = gio.Fixture("s16_mixed")
    batch = BatchedCyberDefenseEnv(fx.topo, fx.cfg, 2, {k: v[:2] for k, v in fx.init.items()}, device="cuda:0",
                                   max_groups=2, max_devs=16)
    env = CyberDefenseEnvView(batch, 1)
    with pytest.raises(ValueError):
        env.step((1, [0], [1], 0))                       # mode not set
    env.mode = "defender"
    with pytest.raises(ValueError):
        env.step((11, [0], [], 0))                       # volt_typhoon_env.py:966
    with pytest.raises(KeyError):
        env.step((1, [0], [99], 0))                      # unknown device id
    before = batch.state_numpy()["flags"].copy()
    s0 = env.reset()
    env.step((7, [0], [2, 3], 0))
    assert env.step_num == 1
    s1 = env.reset(from_init=True)
    np.testing.assert_array_equal(s0, s1)
    np.testing.assert_array_equal(batch.state_numpy()["flags"][0], before[0])   # env 0 untouched throughout
    env.base_line = "No Defense"
    assert env.base_line == "No Defense"
    st, r, _, _, info, _ = env.step(None)
    assert info["executed_atype"] == 8
    assert env.get_num_action_types("defender") == 14 and env.get_num_action_types("attacker") == 3
    a = env.sample_action()
    assert len(a) == 4
    batch.close()


def test_object_facade_reads_the_soa():
    """The reads the reference's agents make on env internals (IPPO.py:74-96, HMARL.py:126-137)."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.env_view import CyberDefenseEnvView
    fx = gio.Fixture("s64_mixed")
    batch = BatchedCyberDefenseEnv(fx.topo, fx.cfg, fx.N, fx.init, device="cuda:0", max_groups=1, max_devs=fx.L)
    env = CyberDefenseEnvView(batch, 1)
    for t in range(40):
        env.mode = fx.mode_name(1, t)
        env.step(fx.python_action(1, t))
    f = fx.exp["flags"][1, 39]
    devs = env._get_ordered_devices()
    assert len(devs) == fx.M
    for d in devs:   # IPPO.build_visibility_mask's predicate, both roles
        att = d.Known_to_attacker and d.attacker_owned and not d.Not_yet_added
        dfn = (not d.Not_yet_added) and d.attacker_owned
        assert att == bool((f[d.id] & S.F_KNOWN) and (f[d.id] & S.F_OWNED) and not (f[d.id] & S.F_NYA))
        assert dfn == bool((f[d.id] & S.F_OWNED) and not (f[d.id] & S.F_NYA))
        assert d.isCompromised == bool(f[d.id] & S.F_COMP)
        assert d.busy_time == int(fx.exp["busy"][1, 39, d.id])
        assert (d.workload is not None) == (fx.exp["wl"][1, 39, d.id] > 0)
    sim = env.simulator
    assert sim.getExploitsSize() == fx.topo.X and set(sim.subnet.net.keys()) == set(range(fx.M))
    assert len(sim.subnet.graph.get_edgelist()) == fx.topo.E
    n_blocked = int(fx.exp["blocked"][1, 39].sum())
    assert len(sim.subnet.graph.blocked_edges()) <= n_blocked and (n_blocked == 0) == (len(sim.subnet.graph.blocked_edges()) == 0)
    assert len(sim.logger.get_logs()) == min(int(fx.exp["ienv"][1, 39, S.I_LOG_TOTAL]), S.LOG_RING)
    live_ids = [i for i, d in sim.subnet.net.items() if not d.Not_yet_added]   # HMARL.py:470
    assert live_ids == [i for i in range(fx.M) if not (f[i] & S.F_NYA)]
    batch.close()


def test_facade_graph_follows_added_edges():
    """`env.simulator.subnet.graph` shows the edges evolve_network added to THIS env (star around the
    attacker-owned hub, CyberDefenseEnv.py:738-774), in igraph's neighbour order."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.env_view import CyberDefenseEnvView
    from cygym_amd import abi
    fx = gio.Fixture("s24_star")
    batch = BatchedCyberDefenseEnv(fx.topo, fx.cfg, fx.N, fx.init, device="cuda:0", max_groups=1, max_devs=fx.L)
    env = CyberDefenseEnvView(batch, 2)
    T = 100
    for t in range(T):
        env.mode = fx.mode_name(2, t)
        env.step(fx.python_action(2, t))
    n = int(fx.exp["ienv"][2, T - 1, S.I_FLAGS]) >> S.E_NX_SHIFT
    assert n > 0
    exp = abi.unpack_extra(fx.exp["extra"][2, T - 1], n, fx.K)
    g = env.simulator.subnet.graph
    assert g.ecount() == fx.topo.E + n
    assert g.get_edgelist()[fx.topo.E:] == [(u, v) for (u, v, _b) in exp]
    u0, v0, _ = exp[0]
    assert v0 in g.neighbors(u0, mode="out") and u0 in g.neighbors(v0, mode="in")
    assert g.neighbors(u0, mode="out") == sorted(g.neighbors(u0, mode="out"))
    assert g.get_eid(u0, v0) == fx.topo.E and g.get_eid(v0, v0, error=False) in (-1, g.get_eid(v0, v0, error=False))
    assert {(u, v) for (u, v, b) in exp if b} <= g.blocked_edges()
    batch.close()


def test_visibility_mask_on_device_matches_the_object_walk():
    """BatchedCyberDefenseEnv.visibility_mask (device tensor ops) == IPPO.build_visibility_mask's walk over the
    facade's Device objects (IPPO.py:74-96), for both roles, on a state that has every flag combination."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.env_view import CyberDefenseEnvView
    fx = gio.Fixture("s24_star")
    env = BatchedCyberDefenseEnv(fx.topo, fx.cfg, fx.N, fx.init, device="cuda:0", max_groups=fx.G, max_devs=fx.L)
    st = {k: fx.exp[k][:, 120] for k in gio.STATE_KEYS}
    st["extra"] = fx.exp["extra"][:, 120]
    env.load_state(st)
    for role in ("attacker", "defender"):
        got = env.visibility_mask(role).cpu().numpy()
        for e in range(fx.N):
            devs = CyberDefenseEnvView(env, e)._get_ordered_devices()
            exp = [float((d.Known_to_attacker if role == "attacker" else True) and d.attacker_owned and not d.Not_yet_added) for d in devs]
            np.testing.assert_array_equal(got[e], np.asarray(exp, np.float32), err_msg=f"{role} env {e}")
        assert got.sum() > 0
    env.close()


def test_logs_come_from_the_long_history_when_the_batch_keeps_one():
    """detector=True keeps the last 2048 comm-log entries per env (cygym_buffers.hist): the view's `logs` return value
    then holds min(total, 2048) entries whose tail equals the 32-entry ring the scans read."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.env_view import CyberDefenseEnvView
    from cygym_amd.topology import make_topology
    from cygym_amd import abi
    topo, init, ck = make_topology(64, 4, seed=5, n_active=56)
    cfg = abi.EnvConfig(seed=5, **ck)
    batch = BatchedCyberDefenseEnv(topo, cfg, 8, init, device="cuda:0", max_groups=1, max_devs=8, detector=True)
    env = CyberDefenseEnvView(batch, 3)
    env.mode = "attacker"
    logs = []
    for _ in range(12):
        logs = env.step((1, [0], [], 0))[5]
    total = int(batch.state["ienv"][3, S.I_LOG_TOTAL].item())
    assert total > S.LOG_RING, "the spread must have logged more than one ring's worth"
    assert len(logs) == min(total, S.HIST_RING)
    ring = batch.state["ring"][3].cpu().numpy().view(np.uint16).reshape(S.LOG_RING, 2)
    for j in range(total - S.LOG_RING, total):
        f, t = ring[j % S.LOG_RING]
        rec = logs[j - (total - len(logs))]
        assert (rec["from_device"], rec["to_device"], rec["kind"], rec["time_step"]) == (int(f), int(t), "A", 0)
    batch.close()


def test_views_of_one_batch_keep_their_own_base_line():
    """`env.base_line` is a plain attribute of each reference env object, assigned per turn by the rollout loops
    (do_agent.py:218-221).  Three views of ONE batch under three baselines must reproduce the three single-env fixtures the
    reference generated with those baselines (same network, env ids 0 / 1 / 2) -- ticking interleaved, none disturbing another."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.env_view import CyberDefenseEnvView
    import dataclasses
    names = ["s16_baselines_no_defense", "s16_baselines_no_attack", "s16_baselines_preset"]
    fxs = [gio.Fixture(n) for n in names]
    f0 = fxs[0]
    for j, fx in enumerate(fxs):      # one shared network, consecutive env ids: the three fixtures ARE one batch
        assert fx.N == 1 and fx.cfg.env_id_base == j and fx.cfg.seed == f0.cfg.seed
        for k in ("out_ptr", "out_col", "in_col", "dstatic", "vuln", "os_val"):
            np.testing.assert_array_equal(getattr(fx.topo, k), getattr(f0.topo, k))
    init = {k: np.concatenate([fx.init[k] for fx in fxs]) for k in gio.STATE_KEYS}
    cfg = dataclasses.replace(f0.cfg, baseline="Nash", env_id_base=0)   # the batch's own setting is none of the three
    L = max(fx.L for fx in fxs)
    batch = BatchedCyberDefenseEnv(f0.topo, cfg, 3, init, device="cuda:0", max_groups=max(fx.G for fx in fxs), max_devs=L)
    views = [CyberDefenseEnvView(batch, e) for e in range(3)]
    bls = ["No Defense", "No Attack", "Preset"]
    T = min(fx.T for fx in fxs)
    for t in range(T):
        for e, (env, fx) in enumerate(zip(views, fxs)):
            env.mode = fx.mode_name(0, t)
            env.base_line = bls[e]          # per turn, like the reference loop
            state, raw, shaped, done, info, logs = env.step(fx.python_action(0, t))
            np.testing.assert_array_equal(state, fx.exp["obs"][0, t].reshape(-1).astype(np.float64), err_msg=f"{names[e]} t={t}")
            assert abs(raw - fx.exp["raw"][0, t]) < 1e-9 and abs(shaped - fx.exp["shaped"][0, t]) < 1e-9
            assert info["executed_atype"] == int(fx.exp["ienv"][0, t][S.I_LAST_ATYPE])
        got = batch.state_numpy()
        for e, fx in enumerate(fxs):
            np.testing.assert_array_equal(got["flags"][e], fx.exp["flags"][0, t].astype(np.uint8))
            np.testing.assert_array_equal(got["ienv"][e][: S.I_FLAGS], fx.exp["ienv"][0, t][: S.I_FLAGS])
    assert [v.base_line for v in views] == bls and batch.cfg.baseline == "Nash"
    batch.close()


def test_counters_are_one_row_copy_and_writes_are_coalesced():
    """The reference zeroes twelve counters before a rollout, each behind a hasattr (do_agent.py:192-196): through the view that
    is ONE device-to-host row copy and ONE upload before the next launch, and the values land on the device."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.env_view import CyberDefenseEnvView
    fx = gio.Fixture("s16_mixed")
    batch = BatchedCyberDefenseEnv(fx.topo, fx.cfg, fx.N, fx.init, device="cuda:0", max_groups=fx.G, max_devs=fx.L)
    env, other = CyberDefenseEnvView(batch, 1), CyberDefenseEnvView(batch, 0)
    for t in range(12):
        env.mode = fx.mode_name(1, t)
        env.step(fx.python_action(1, t))
    assert env.step_num == 12 and env.work_done == int(fx.exp["ienv"][1, 11][S.I_WORK_DONE])
    fetches = []
    orig = CyberDefenseEnvView._counter_rows

    def counting(self):
        before = self._rows
        out = orig(self)
        if self._rows is not before:
            fetches.append(1)
        return out
    CyberDefenseEnvView._counter_rows = counting
    try:
        attrs = ["step_num", "defender_step", "attacker_step", "work_done", "checkpoint_count", "defensive_cost", "clearning_cost",
                 "revert_count", "scan_cnt", "compromised_devices_cnt", "edges_blocked", "edges_added"]
        for a in attrs:
            if hasattr(env, a):
                setattr(env, a, 0)
        assert len(fetches) == 0          # the row the last step brought back served all twelve reads
        assert env.step_num == 0 and env.defensive_cost == 0.0      # pending writes read back
        assert int(batch._state["ienv"][1, S.I_STEP_NUM].item()) == 12     # ... not on the device yet
        ie = batch.state["ienv"][1].cpu().numpy()                          # handing out `state` uploads them
        assert ie[S.I_STEP_NUM] == 0 and ie[S.I_WORK_DONE] == 0 and float(batch.state["fenv"][1, S.D_DEF_COST].item()) == 0.0
        assert ie[S.I_RNG_TICK] == int(fx.exp["ienv"][1, 11][S.I_RNG_TICK])     # untouched columns keep their values
        assert other.step_num == 0 and len(fetches) == 1                      # another view: its own row, one copy
        env.work_done = 7
        env.mode = fx.mode_name(1, 12)
        env.step(fx.python_action(1, 12))                                      # the launch flushes first
        assert env.step_num == 1 and env.work_done >= 7
    finally:
        CyberDefenseEnvView._counter_rows = orig
    s = env.seed(1234)
    assert s == [1234] and batch.cfg.seed == 1234
    env.set_exploit_seed(5)
    a = env.sample_exploits()
    env.set_exploit_seed(5)
    np.testing.assert_array_equal(a, env.sample_exploits())
    batch.close()
