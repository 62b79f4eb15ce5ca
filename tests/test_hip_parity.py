"""GPU parity: the HIP kernels (through the C ABI) against
  (1) the golden fixtures produced by the reference itself, tick by tick, bit-exact on
      every integer plane / counter, 1e-9 on rewards (north star: 1e-6), and
  (2) the CPU oracle on seeded synthetic batches (topology generator + action script),
  (3) size-independent properties at BASELINE.json's full sizes.
"""
import numpy as np
import pytest

import golden_io as gio
from cygym_amd import abi
from cygym_amd import spec as S
from cygym_amd.actions import gen_actions_numpy
from cygym_amd.topology import make_topology

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def _env(topo, cfg, n, init, **kw):
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    return BatchedCyberDefenseEnv(topo, cfg, n, init, device="cuda:0", **kw)


def _fixture_cases():
    """Every fixture through the full-feature kernels (detector buffers bound); those in which the reference never
    trains its detector also through the lean kernels."""
    out = []
    for name in gio.fixture_names():
        out.append((name, True))
        if not np.load(gio.os.path.join(gio.GOLDEN, name + ".npz"))["det_env"].size:
            out.append((name, False))
    return out


@pytest.mark.parametrize("name,detector", _fixture_cases())
def test_hip_matches_reference_fixture(name, detector):
    fx = gio.Fixture(name)
    env = _env(fx.topo, fx.cfg, fx.N, fx.init, max_groups=fx.G, max_devs=fx.L, detector=detector)
    from oracle import driver as od
    act = od.alloc_actions(fx.N, fx.G, fx.L)
    alive = np.ones(fx.N, bool)
    checked = 0
    for t in range(fx.T):
        fx.actions(t, act, flags=env.state["flags"].cpu().numpy())
        env.set_actions_numpy(act)
        view = "defender" if t % 2 else "attacker"      # the fused role view of the state the tick leaves behind
        obs, raw, shaped, done = env.step(view=view)
        if fx.det_events.get(t):   # the host's part of Detector.train, with the forest the reference fitted
            fx.service_detectors(t, env.state_numpy(), env.install_forest)
        got = env.state_numpy()
        assert not (got["ienv"][:, S.I_FLAGS] & S.E_UNPINNED).any(), f"{name} t={t}: a scan ran without a current forest"
        same = fx.exp["topo_same"][:, t].astype(bool)
        ovf = (got["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF) != 0
        if fx.follows_topology():   # the edges evolve_network adds are part of the compared state
            assert not ovf.any(), f"{name} t={t}: extra-edge list overflowed"
        else:
            assert np.array_equal(ovf[alive], ~same[alive]), f"{name} t={t}: TOPO_OVF {ovf} vs topo_same {same}"
            alive &= same
        if not alive.any():
            break
        sel = np.where(alive)[0]
        exp = fx.expected_state(t)
        g = {k: v[sel] for k, v in got.items()}
        g["ienv"] = g["ienv"].copy()
        g["ienv"][:, S.I_FLAGS] &= ~0x80   # kernel-private STAR_OK bit
        bad = gio.compare_state(g, {k: v[sel] for k, v in exp.items()}, f"{name} t={t}")
        assert not bad, "\n".join(bad[:8])
        slow = not fx.cfg.fast_scan     # per-log scan path: the anomaly column holds decision_function floats (1e-6)
        gio.assert_obs_equal(obs.cpu().numpy()[sel].reshape(len(sel), -1), fx.exp["obs"][sel, t].reshape(len(sel), -1), slow, f"{name} obs t={t}")
        np.testing.assert_allclose(raw.cpu().numpy()[sel], fx.exp["raw"][sel, t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(shaped.cpu().numpy()[sel], fx.exp["shaped"][sel, t], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(done.cpu().numpy()[sel], fx.exp["done"][sel, t])
        if view == "defender":
            gio.assert_obs_equal(env.role_obs[view].cpu().numpy()[sel], fx.exp["obs_def"][sel, t], slow, f"{name} fused defender view t={t}")
        else:
            np.testing.assert_array_equal(env.role_obs[view].cpu().numpy()[sel], fx.exp["obs_att"][sel, t], err_msg=f"{name} fused attacker view t={t}")
        if t % 7 == 0:
            gio.assert_obs_equal(env.observe(1).cpu().numpy()[sel], fx.exp["obs_def"][sel, t], slow, f"{name} observe(1) t={t}")
            np.testing.assert_array_equal(env.observe(2).cpu().numpy()[sel], fx.exp["obs_att"][sel, t])
        checked += 1
    assert checked > 0
    if detector and alive.all():
        got = env.state_numpy()
        fx.check_final_hist(got["hist"], got["ienv"][:, S.I_LOG_TOTAL])
    env.close()


def test_hip_randomize_matches_reference():
    fx = gio.Fixture("s16_randomize")
    env = _env(fx.topo, fx.cfg, fx.N, fx.pre, max_groups=1, max_devs=4)
    env.randomize()
    got = env.state_numpy()
    for k in ("flags", "busy", "wl", "comp_by"):
        np.testing.assert_array_equal(got[k], fx.init[k].astype(got[k].dtype), err_msg=k)
    np.testing.assert_array_equal(got["ienv"][:, S.I_RNG_TICK], fx.init["ienv"][:, S.I_RNG_TICK])
    env.close()


def _oracle_pair(M, blocks, N, seed, n_active=None, cfg_kw=None, env_id_base=0):
    from oracle import driver as od
    topo, init, ck = make_topology(M, blocks, seed=seed, n_active=n_active)
    ck.update(cfg_kw or {})
    cfg = abi.EnvConfig(seed=seed, env_id_base=env_id_base, **ck)
    L = max(1, M // 8)
    env = _env(topo, cfg, N, init, max_groups=1, max_devs=L)
    ob = od.OracleBatch(topo, cfg, N)
    ob.load_state(init)
    return topo, cfg, env, ob, L


@pytest.mark.parametrize("M,blocks,N,ticks,n_active", [(13, 1, 40, 200, 11), (37, 2, 33, 200, 30), (130, 3, 65, 120, 120),
                                                       (16, 2, 64, 300, 12), (64, 4, 512, 260, 56),
                                                       (256, 1, 256, 160, 230), (600, 4, 48, 60, 560),
                                                       (2048, 32, 24, 40, 2000)])
def test_hip_matches_oracle_synthetic(M, blocks, N, ticks, n_active):
    topo, cfg, env, ob, L = _oracle_pair(M, blocks, N, seed=3, n_active=n_active, env_id_base=123)
    for t in range(ticks):
        env.gen_actions(t)
        act = gen_actions_numpy(cfg.seed, cfg.env_id_base, N, M, topo.X, t, L)
        for k in act:   # the on-device script equals its numpy mirror (lists: the first dev_cnt entries)
            g = env.act[k].cpu().numpy().reshape(act[k].shape)
            if k == "dev_idx":
                g = np.where(np.arange(L)[None, :] < act["dev_cnt"][:, :1], g, 0)
            np.testing.assert_array_equal(g, act[k], err_msg=f"script {k} t={t}")
        obs, raw, shaped, done = env.step()
        o_obs, o_raw, o_shaped, o_done = ob.step(act)
        np.testing.assert_allclose(raw.cpu().numpy(), o_raw, rtol=0, atol=1e-9, err_msg=f"raw t={t}")
        np.testing.assert_allclose(shaped.cpu().numpy(), o_shaped, rtol=0, atol=1e-9, err_msg=f"shaped t={t}")
        if t % 5 == 0 or t == ticks - 1:
            got = env.state_numpy()
            got["ienv"] = got["ienv"].copy()
            got["ienv"][:, S.I_FLAGS] &= ~0x80
            bad = gio.compare_state(got, ob.state, f"M={M} t={t}")
            assert not bad, "\n".join(bad[:8])
            np.testing.assert_array_equal(obs.cpu().numpy(), o_obs, err_msg=f"obs t={t}")
    # where evolve would have to ADD edges (hub deactivated on a sparse net) both sides must say so
    got = env.state_numpy()
    np.testing.assert_array_equal(got["ienv"][:, S.I_FLAGS] & (S.E_TOPO_OVF | S.E_BUSY_SAT),
                                  ob.state["ienv"][:, S.I_FLAGS] & (S.E_TOPO_OVF | S.E_BUSY_SAT))
    if M < 500:   # dense attacker edges: the topology never needs to change
        assert not (ob.state["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF).any()
    env.close()


@pytest.mark.parametrize("M,blocks,N,ticks,n_active,max_extra", [(24, 1, 96, 300, 12, 160), (64, 4, 128, 240, 40, 160),
                                                                 (256, 1, 96, 160, 200, 192), (24, 1, 64, 200, 12, 6)])
def test_hip_matches_oracle_with_added_edges(M, blocks, N, ticks, n_active, max_extra):
    """evolve_network ADDS edges (star around the attacker-owned hub, CyberDefenseEnv.py:738-774): ownership is
    reshuffled before the episode and keeps changing (p_attacker > 0, removals), so the per-env extra-edge lists
    fill up while spread / probe / block / unblock run over the merged rows.  max_extra=6 also overflows the
    list: both sides must then raise CG_E_TOPO_OVF at the same tick and drop the same edges."""
    from oracle import driver as od
    topo, init, ck = make_topology(M, blocks, seed=9, n_active=n_active, max_extra=max_extra)
    ck.update(dict(lambda_events=1.6, p_add=0.45, p_attacker=0.08, num_of_device=max(2, n_active // 2), min_network_size=2))
    cfg = abi.EnvConfig(seed=17, env_id_base=5000, **ck)
    L = max(1, M // 8)
    env = _env(topo, cfg, N, init, max_groups=1, max_devs=L)
    ob = od.OracleBatch(topo, cfg, N)
    ob.load_state(init)
    env.randomize()
    ob.randomize()
    seen_edges = 0
    for t in range(ticks):
        env.gen_actions(t)
        act = gen_actions_numpy(cfg.seed, cfg.env_id_base, N, M, topo.X, t, L)
        if t % 3 == 0:   # lean on block / unblock aimed at attacker-owned devices (the star's endpoints)
            fl = ob.state["flags"]
            for e in range(0, N, 2):
                if act["mode"][e] != S.MODE_DEFENDER:
                    continue
                owned = np.flatnonzero(fl[e] & S.F_OWNED)
                if owned.size:
                    k = min(L, owned.size)
                    act["atype"][e, 0] = 6 if (t // 3 + e) % 3 else 9
                    act["dev_cnt"][e, 0] = k
                    act["dev_idx"][e, :k] = owned[:k]
            env.set_actions_numpy(act)
        obs, raw, shaped, done = env.step()
        o_obs, o_raw, o_shaped, o_done = ob.step(act)
        np.testing.assert_allclose(raw.cpu().numpy(), o_raw, rtol=0, atol=1e-9, err_msg=f"raw t={t}")
        if t % 4 == 0 or t == ticks - 1:
            got = env.state_numpy()
            got["ienv"] = got["ienv"].copy()
            got["ienv"][:, S.I_FLAGS] &= ~0x80
            np.testing.assert_array_equal(got["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF, ob.state["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF,
                                          err_msg=f"TOPO_OVF t={t}")
            bad = gio.compare_state(got, ob.state, f"M={M} t={t}")
            assert not bad, "\n".join(bad[:8])
            np.testing.assert_array_equal(obs.cpu().numpy(), o_obs, err_msg=f"obs t={t}")
        seen_edges = max(seen_edges, int((ob.state["ienv"][:, S.I_FLAGS].astype(np.int64) >> S.E_NX_SHIFT).max()))
    assert seen_edges > 0, "the scenario never added an edge"
    ovf = (ob.state["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF) != 0
    assert ovf.any() == (max_extra == 6)
    env.close()


def test_hip_auto_reset_and_episode_cap():
    """Episode cap (CyberDefenseEnv.py:549) with auto-reset: after the cap the env restarts from the snapshot."""
    topo, cfg, env, ob, L = _oracle_pair(16, 2, 32, seed=5, n_active=14, cfg_kw=dict(episode_limit=20, auto_reset=1))
    for t in range(50):
        env.gen_actions(t)
        act = gen_actions_numpy(cfg.seed, cfg.env_id_base, 32, 16, topo.X, t, L)
        _, raw, _, done = env.step()
        _, o_raw, _, o_done = ob.step(act)
        np.testing.assert_array_equal(done.cpu().numpy(), o_done)
        np.testing.assert_allclose(raw.cpu().numpy(), o_raw, rtol=0, atol=1e-9)
    got = env.state_numpy()
    got["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(got, ob.state, "auto_reset")
    assert (got["ienv"][:, S.I_STEP_NUM] == 50 - 21 * 2).all()
    env.close()


def test_hip_reset_subset():
    topo, cfg, env, ob, L = _oracle_pair(64, 4, 16, seed=9, n_active=60)
    for t in range(30):
        env.gen_actions(t)
        env.step()
        ob.step(gen_actions_numpy(cfg.seed, cfg.env_id_base, 16, 64, topo.X, t, L))
    ids = [1, 5, 15]
    env.reset(ids)
    ob.reset(ids)
    got = env.state_numpy()
    got["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(got, ob.state, "reset subset")
    env.close()


@pytest.mark.parametrize("M,blocks,N", [(64, 4, 4096), (256, 1, 4096), (256, 1, 16384), (2048, 32, 4096)])
def test_full_size_properties(M, blocks, N):
    """BASELINE.json sizes: shard-independence (global env id keys the RNG), oracle agreement on a
    sampled sub-batch, and state invariants."""
    from oracle import driver as od
    seed = 11
    topo, init, ck = make_topology(M, blocks, seed=seed)
    cfg = abi.EnvConfig(seed=seed, env_id_base=0, **ck)
    L = max(1, M // 8)
    env = _env(topo, cfg, N, init, max_groups=1, max_devs=L)
    # a second env holding only the LAST 64 envs of the batch, and the oracle on the same slice
    base = N - 64
    cfg_tail = abi.EnvConfig(seed=seed, env_id_base=base, **ck)
    tail = _env(topo, cfg_tail, 64, init, max_groups=1, max_devs=L)
    ob = od.OracleBatch(topo, cfg_tail, 64)
    ob.load_state(init)
    ticks = 40 if M < 2048 else 16
    ret = torch.zeros(N, dtype=torch.float64, device="cuda:0")
    for t in range(ticks):
        env.gen_actions(t)
        tail.gen_actions(t)
        _, raw, _, _ = env.step()
        _, raw_t, _, _ = tail.step()
        ret += raw
        assert torch.equal(raw[base:], raw_t), f"shard dependence at t={t}"
        act = gen_actions_numpy(seed, base, 64, M, topo.X, t, L)
        _, o_raw, _, _ = ob.step(act)
        np.testing.assert_allclose(raw_t.cpu().numpy(), o_raw, rtol=0, atol=1e-9)
    full = env.state_numpy()
    part = tail.state_numpy()
    for k in ("flags", "busy", "wl", "comp_by", "blocked", "ienv"):
        np.testing.assert_array_equal(full[k][base:], part[k], err_msg=k)
    part["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(part, ob.state, "tail vs oracle")
    # a device that is not part of the network holds no workload (evolve removal / action 7 clear it)
    assert not (((full["flags"] & S.F_NYA) != 0) & (full["wl"] > 0)).any()
    assert (full["ienv"][:, S.I_STEP_NUM] == ticks).all()
    assert (full["ienv"][:, S.I_DEF_STEP] + full["ienv"][:, S.I_ATT_STEP] == ticks).all()
    assert np.isfinite(ret.cpu().numpy()).all()
    env.close(); tail.close()


@pytest.mark.parametrize("M,blocks,N,T", [(64, 4, 96, 37), (256, 1, 128, 30), (600, 4, 24, 22)])
def test_rollout_equals_stepwise(M, blocks, N, T):
    """cygym_rollout (T ticks fused in one launch) == T calls of cygym_step == the oracle, including an
    episode cap with auto-reset in the middle."""
    topo, cfg, env, ob, L = _oracle_pair(M, blocks, N, seed=21, n_active=M - 8, env_id_base=5,
                                         cfg_kw=dict(episode_limit=17, auto_reset=1))
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    topo2, init2, _ = make_topology(M, blocks, seed=21, n_active=M - 8)
    fused = BatchedCyberDefenseEnv(topo2, cfg, N, init2, device="cuda:0", max_groups=1, max_devs=L)
    act, out = fused.alloc_rollout(T)
    fused.gen_actions_rollout(0, act)
    fused.rollout(act, out)
    for t in range(T):
        env.gen_actions(t)
        obs, raw, shaped, done = env.step()
        o_obs, o_raw, o_shaped, o_done = ob.step(gen_actions_numpy(cfg.seed, cfg.env_id_base, N, M, topo.X, t, L))
        assert torch.equal(out["obs"][t], obs), f"obs t={t}"
        assert torch.equal(out["raw"][t], raw) and torch.equal(out["shaped"][t], shaped) and torch.equal(out["done"][t], done)
        np.testing.assert_allclose(raw.cpu().numpy(), o_raw, rtol=0, atol=1e-9)
        np.testing.assert_array_equal(done.cpu().numpy(), o_done)
    a, b = fused.state_numpy(), env.state_numpy()
    for k in ("live", "stash", "blocked", "blocked_in", "ring", "ienv", "fenv"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    b["ienv"] = b["ienv"].copy()
    b["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(b, ob.state, "stepwise vs oracle")
    assert out["done"].any(), "the episode cap must have been crossed"
    env.close(); fused.close()


@pytest.mark.parametrize("M,blocks,N,T,max_extra", [(24, 1, 64, 60, 160), (256, 1, 64, 40, 192)])
def test_rollout_with_added_edges(M, blocks, N, T, max_extra):
    """The fused rollout keeps the extra-edge list in LDS across ticks: edges added by evolve_network at tick t
    are walked at tick t+1 inside the same launch, an episode cap restores the snapshot's (empty) list, and the
    result equals stepping tick by tick and the oracle."""
    from oracle import driver as od
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    topo, init, ck = make_topology(M, blocks, seed=23, n_active=max(4, M // 2), max_extra=max_extra)
    ck.update(dict(lambda_events=1.6, p_add=0.45, p_attacker=0.08, num_of_device=max(2, M // 4), min_network_size=2,
                   episode_limit=23, auto_reset=1))
    cfg = abi.EnvConfig(seed=29, env_id_base=900, **ck)
    L = max(1, M // 8)
    env = _env(topo, cfg, N, init, max_groups=1, max_devs=L)
    fused = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=L)
    ob = od.OracleBatch(topo, cfg, N)
    ob.load_state(init)
    for b in (env, fused, ob):
        b.randomize()
    act, out = fused.alloc_rollout(T)
    fused.gen_actions_rollout(0, act)
    fused.rollout(act, out)
    seen = 0
    for t in range(T):
        env.gen_actions(t)
        obs, raw, shaped, done = env.step()
        o_obs, o_raw, o_shaped, o_done = ob.step(gen_actions_numpy(cfg.seed, cfg.env_id_base, N, M, topo.X, t, L))
        assert torch.equal(out["obs"][t], obs), f"obs t={t}"
        assert torch.equal(out["raw"][t], raw) and torch.equal(out["done"][t], done), f"t={t}"
        np.testing.assert_allclose(raw.cpu().numpy(), o_raw, rtol=0, atol=1e-9, err_msg=f"raw t={t}")
        seen = max(seen, int((ob.state["ienv"][:, S.I_FLAGS].astype(np.int64) >> S.E_NX_SHIFT).max()))
    a, b = fused.state_numpy(), env.state_numpy()
    for k in ("live", "stash", "blocked", "blocked_in", "ring", "ienv", "fenv"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    a["ienv"] = a["ienv"].copy()
    a["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(a, ob.state, "fused vs oracle")
    assert seen > 0 and out["done"].any()
    env.close(); fused.close()


def test_differential_fuzz_sample(monkeypatch):
    """A slice of tools/fuzz.py (random sizes, evolve parameters, extra-edge capacities incl. too small ones,
    reshuffled ownership, episode caps; per-tick vs oracle and fused vs per-tick).  The full campaign
    (1500 cases x 300 ticks, all agreeing) is run by hand on the GPU box; see DESIGN.md section 5."""
    import sys
    import os
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools", "fuzz.py")
    spec = importlib.util.spec_from_file_location("cg_tools_fuzz", path)
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    monkeypatch.setattr(sys, "argv", ["tools/fuzz.py", "--cases", "16", "--seed0", "5000", "--ticks", "150"])
    fuzz.main()   # exits non-zero (SystemExit) on the first mismatch


@pytest.mark.parametrize("M", [64, 256])
def test_long_rows_at_a_compile_time_size_run_on_the_run_time_kernels(M):
    """The 64- and 256-device kernels count and select a row's blocked bits in a fixed number of words (pool_pick<NW>,
    cg_defender.hpp): a topology with duplicate edges whose longest row exceeds the device count is routed to the run-time-size
    kernels by cygym_create (DevTopo::ct).  One device's out-entries are repeated until its row holds more than M slots; block /
    unblock lists aimed at it and at its neighbours, and every other action of the script, must match the oracle."""
    import dataclasses
    from oracle import driver as od
    topo, init, ck = make_topology(M, 4 if M == 64 else 1, seed=5, n_active=(M * 9) // 10, max_extra=0)
    op, oc = np.asarray(topo.out_ptr), np.asarray(topo.out_col)
    hub = int(np.argmax(np.diff(op)))
    row = oc[op[hub]:op[hub + 1]]
    reps = M // len(row) + 2                      # the row becomes reps * len(row) > M slots
    rows = [oc[op[u]:op[u + 1]] if u != hub else np.sort(np.tile(row, reps)) for u in range(M)]
    out_ptr = np.zeros(M + 1, np.int32); out_ptr[1:] = np.cumsum([len(r) for r in rows])
    out_col = np.concatenate(rows).astype(np.int32)
    assert out_ptr[hub + 1] - out_ptr[hub] > M
    in_ptr, in_col, in_eid = abi.build_in_csr(M, out_ptr, out_col)
    topo = dataclasses.replace(topo, out_ptr=out_ptr, out_col=out_col, in_ptr=in_ptr, in_col=in_col, in_eid=in_eid).normalised()
    EW = (len(out_col) + 31) // 32
    init = dict(init)
    init["blocked"] = np.zeros((1, EW), np.uint32); init["blocked_in"] = np.zeros((1, EW), np.uint32)
    cfg = abi.EnvConfig(seed=5, env_id_base=40, **ck)
    N, L, T = 48, max(1, M // 8), 80
    env = _env(topo, cfg, N, init, max_groups=1, max_devs=L)
    ob = od.OracleBatch(topo, cfg, N)
    ob.load_state(init)
    nb = np.unique(np.concatenate([[hub], row]))[:L]
    for t in range(T):
        env.gen_actions(t)
        act = gen_actions_numpy(cfg.seed, cfg.env_id_base, N, M, topo.X, t, L)
        if t % 2 == 0:   # defender ticks: every third env blocks / unblocks around the long row
            for e_ in range(0, N, 3):
                if act["mode"][e_] != S.MODE_DEFENDER:
                    continue
                act["atype"][e_, 0] = 6 if (t // 2 + e_) % 3 else 9
                act["dev_cnt"][e_, 0] = len(nb)
                act["dev_idx"][e_, :len(nb)] = nb
            env.set_actions_numpy(act)
        obs, raw, shaped, done = env.step()
        o_obs, o_raw, o_shaped, o_done = ob.step(act)
        np.testing.assert_allclose(raw.cpu().numpy(), o_raw, rtol=0, atol=1e-9, err_msg=f"raw t={t}")
        if t % 4 == 0 or t == T - 1:
            got = env.state_numpy()
            got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
            bad = gio.compare_state(got, ob.state, f"M={M} t={t}")
            assert not bad, "\n".join(bad[:8])
    assert (ob.state["blocked"] != 0).any()
    env.close()


def test_lists_read_from_global_memory(monkeypatch):
    """Run-time sizes where it buys a resident wave (4096 x 2048 with its extra-edge list: 5 -> 6 per CU) leave the tick's
    device list, the extra-edge list and the in-row bounds in global memory (choose_launch, cygym_hip.hip).  No network that
    fits a test picks that plan by itself: force it (CYGYM_CBY_GLOBAL + CYGYM_LISTS_GLOBAL) and run the added-edge scenarios --
    per tick with aimed block / unblock lists, and as a rollout -- against the oracle."""
    monkeypatch.setenv("CYGYM_CBY_GLOBAL", "1")
    monkeypatch.setenv("CYGYM_LISTS_GLOBAL", "1")
    topo, init, ck = make_topology(600, 4, seed=9, n_active=500, max_extra=192)
    env = _env(topo, abi.EnvConfig(seed=1, **ck), 8, init, max_groups=1, max_devs=75)
    plan = env.launch_plan()
    env.close()
    assert plan["comp_by_in_global"] == 1 and plan["lists_in_global"] == 1, plan
    test_hip_matches_oracle_with_added_edges(24, 1, 96, 200, 12, 160)
    test_hip_matches_oracle_with_added_edges(24, 1, 64, 120, 12, 6)
    test_hip_matches_oracle_with_added_edges(600, 4, 24, 60, 500, 192)
    test_rollout_with_added_edges(24, 1, 64, 60, 160)
    test_rollout_with_added_edges(600, 4, 24, 40, 192)
    test_hip_matches_oracle_synthetic(600, 4, 24, 60, 550)


@pytest.mark.parametrize("M,wpb", [(100, w) for w in (1, 2, 3, 4, 5, 6, 8, 12, 16)] + [(m, w) for m in (64, 256) for w in (1, 2, 4, 8, 16)])
def test_every_workgroup_shape(M, wpb, monkeypatch):
    """cygym_create picks the waves-per-workgroup shape from the LDS and register budgets, so a given network only ever
    exercises one of the shapes compiled for its size class (nine at run-time sizes, five at 64 and 256 devices): force
    each (CYGYM_WPB, the tuning hook of choose_launch) and check lean and full-feature kernels, per tick and as a
    rollout, against the oracle."""
    from oracle import driver as od
    monkeypatch.setenv("CYGYM_WPB", str(wpb))
    N, T = 50, 36   # N is no multiple of any shape above 2: the last workgroup carries idle waves
    for max_extra, lam in ((0, 0.0), (48, 1.4)):
        topo, init, ck = make_topology(M, 2 if M != 256 else 1, seed=11, n_active=(M * 9) // 10, max_extra=max_extra)
        if lam:
            ck.update(dict(lambda_events=lam, p_add=0.45, p_attacker=0.1, num_of_device=max(2, M // 3), min_network_size=2))
        cfg = abi.EnvConfig(seed=11, env_id_base=5, **ck)
        L = 12
        env = _env(topo, cfg, N, init, max_groups=1, max_devs=L)
        ob = od.OracleBatch(topo, cfg, N)
        ob.load_state(init)
        if lam:
            env.randomize(); ob.randomize()
        start = {k: v.clone() for k, v in env.state.items()}
        for t in range(T):
            env.gen_actions(t)
            act = gen_actions_numpy(cfg.seed, cfg.env_id_base, N, M, topo.X, t, L)
            obs, raw, shaped, done = env.step()
            o = ob.step(act)
            np.testing.assert_array_equal(obs.cpu().numpy(), o[0], err_msg=f"obs wpb={wpb} t={t}")
            np.testing.assert_allclose(raw.cpu().numpy(), o[1], rtol=0, atol=1e-9, err_msg=f"raw wpb={wpb} t={t}")
        got = env.state_numpy()
        got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
        bad = gio.compare_state(got, ob.state, f"wpb={wpb} K={max_extra}")
        assert not bad, "\n".join(bad[:8])
        # the same script as ONE rollout launch from the same start
        for k, v in start.items():
            env.state[k].copy_(v)
        a, out = env.alloc_rollout(T)
        env.gen_actions_rollout(0, a)
        env.rollout(a, out)
        got = env.state_numpy()
        got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
        bad = gio.compare_state(got, ob.state, f"rollout wpb={wpb} K={max_extra}")
        assert not bad, "\n".join(bad[:8])
        env.close()


def test_long_device_lists_replan_the_launch():
    """Device lists as long as the network (max_devs = M, longer than the M/8 the handle was created for): the library
    re-plans its LDS layout at the first step (cygym_step: max_devs > planned) -- also for the WIDE per-tick kernel, which
    keeps the in-CSR maps in LDS -- and block / unblock / clean over whole-network lists match the oracle."""
    from oracle import driver as od
    M, N = 256, 128
    topo, init, ck = make_topology(M, 1, seed=4, n_active=240)
    cfg = abi.EnvConfig(seed=4, **ck)
    small = _env(topo, cfg, N, init, max_groups=1, max_devs=M // 8)
    small.gen_actions(0)
    small.step()                                   # planned for lists of M/8 ...
    env = _env(topo, cfg, N, init, max_groups=1, max_devs=M)
    ob = od.OracleBatch(topo, cfg, N)
    ob.load_state(init)
    act = od.alloc_actions(N, 1, M)
    rs = np.random.RandomState(5)
    for t in range(24):
        act["mode"][:] = S.MODE_DEFENDER if t % 2 == 0 else S.MODE_ATTACKER
        act["atype"][:, 0] = rs.choice([6, 9, 1, 6, 7], size=N) if t % 2 == 0 else 1
        act["n_exploit"][:] = 1
        act["exploit"][:, 0, 0] = 0
        for e in range(N):
            k = int(rs.randint(M // 2, M + 1))
            act["dev_cnt"][e, 0] = k
            act["dev_idx"][e, :k] = rs.permutation(M)[:k]
        env.set_actions_numpy(act)
        obs, raw, shaped, done = env.step()
        o_obs, o_raw, _, _ = ob.step(act)
        got = env.state_numpy()
        got["ienv"] = got["ienv"].copy()
        got["ienv"][:, S.I_FLAGS] &= ~0x80
        bad = gio.compare_state(got, ob.state, f"t={t}")
        assert not bad, "\n".join(bad[:6])
        np.testing.assert_allclose(raw.cpu().numpy(), o_raw, rtol=0, atol=1e-9)
        np.testing.assert_array_equal(obs.cpu().numpy(), o_obs)
    small.close(); env.close()
