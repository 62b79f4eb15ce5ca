"""The IPPO / MAPPO side of the drop-in boundary (IPPO.py:503-640): the grouping of per-device decisions into
env.step(groups) for a batch (cygym_group_actions) and the batched rollout collector (cygym_amd/ippo_rollout.py), against a
numpy restatement of the reference's grouping and the CPU oracle stepped with the groups the reference would have built."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(__file__))
from cygym_amd import abi, host_logic as HL, rng as R, spec as S  # noqa: E402


def _np_visibility(flags, role):
    want = (S.F_KNOWN | S.F_OWNED) if role == "attacker" else S.F_OWNED
    return ((flags & (want | S.F_NYA)) == want)


def _picks(cfg, env_id, tick, counts, single):
    """Index a single-device type keeps among its `counts[t]` devices: the addressed Philox draw of cygym_group_actions."""
    out = {}
    for t in single:
        if counts.get(t, 0) > 0:
            u = R.draw(cfg.seed, cfg.env_id_base + env_id, tick, S.SITE_GROUP_PICK, t)
            out[t] = R.index(u, counts[t])
    return out


def test_gae_is_compute_gae():
    """ippo_rollout.gae against the reference's loop (IPPO.py:301-310), restated per env."""
    from cygym_amd.ippo_rollout import gae
    rs = np.random.RandomState(0)
    T, N = 17, 5
    r, v, d = rs.randn(T, N).astype(np.float32), rs.randn(T + 1, N).astype(np.float32), (rs.rand(T, N) < 0.2)
    adv, ret = gae(torch.from_numpy(r), torch.from_numpy(v), torch.from_numpy(d))
    for n in range(N):
        a = np.zeros(T, np.float32)
        last = 0.0
        for t in reversed(range(T)):
            nt = 1.0 - float(d[t, n])
            delta = r[t, n] + 0.99 * v[t + 1, n] * nt - v[t, n]
            last = delta + 0.99 * 0.95 * nt * last
            a[t] = last
        np.testing.assert_allclose(adv[:, n].numpy(), a, rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(ret[:, n].numpy(), a + v[:-1, n], rtol=1e-5, atol=1e-5)


def test_masked_adjacency_is_the_reference_formula():
    """ippo_rollout.masked_adjacency against IPPO.py:98-110 applied to the all-ones adjacency build_adjacency returns for the
    reference's Subnet: adj * (v v^T), diagonal put back for visible nodes only."""
    from cygym_amd.ippo_rollout import masked_adjacency
    rs = np.random.RandomState(1)
    v = (rs.rand(5, 9) < 0.6).astype(np.float32)
    got = masked_adjacency(torch.from_numpy(v)).numpy()
    for n in range(5):
        out = np.ones((9, 9), np.float32) * np.outer(v[n], v[n])
        eye = np.eye(9, dtype=np.float32) * v[n][:, None]
        np.testing.assert_array_equal(got[n], out * (1 - eye) + eye)


@pytest.mark.gpu
def test_group_actions_kernel_equals_the_numpy_grouping():
    """cygym_group_actions against IPPO.py:560-572 restated in numpy: per-type ascending device lists over the visible
    devices, single-device types (11, 12) keeping the device the addressed Philox draw picks, the no-op fallback, the role's
    visibility mask read off the flag plane or given -- defender and attacker, 13 ... 300 devices."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    for M, N in ((64, 40), (13, 9), (256, 33), (300, 12)):
        topo, init, ck = make_topology(M, 4 if M == 64 else 1, seed=3, n_active=max(8, M - 5))
        cfg = abi.EnvConfig(seed=77, env_id_base=500, **ck)
        env1 = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=max(4, M // 8))
        env1.randomize()
        for t in range(5):
            env1.gen_actions(t); env1.step()              # (the synthetic script is single-action: its own batch)
        st = env1.state_numpy()
        env1.close()
        env = BatchedCyberDefenseEnv(topo, cfg, N, st, device="cuda:0", max_groups=14, max_devs=M)
        ticks = st["ienv"][:, S.I_RNG_TICK]
        g = torch.Generator().manual_seed(M)
        for role, K, noop in (("defender", 14, 8), ("attacker", 5, 3)):
            for given_vis in (False, True):
                types = torch.randint(0, K, (N, M), generator=g)
                types[0] = noop                                              # a row without groups
                types[1, : M // 2] = 11 if role == "defender" else 1
                ex = torch.randint(0, cfg.max_exploits, (N,), generator=g)
                app = torch.randint(0, 4, (N,), generator=g)
                vis_np = (torch.rand((N, M), generator=g) < 0.5).numpy() if given_vis else _np_visibility(st["flags"], role)
                rows = torch.randperm(N, generator=g)[: N - 3].sort().values
                env.act["n_groups"].fill_(-9)
                env.group_actions(rows.to("cuda:0"), types[rows].to("cuda:0"), ex[rows].to("cuda:0"), app[rows].to("cuda:0"), role, n_types=K, noop=noop,
                                  visible=torch.from_numpy(vis_np[rows.numpy()]).to("cuda:0") if given_vis else None)
                got = {k: v.cpu().numpy() for k, v in env.act.items()}
                assert not (env.take_status() & abi.DECODE_TRUNCATED)
                for r in rows.numpy():
                    counts = {t: int((vis_np[r] & (types[r].numpy() == t)).sum()) for t in (11, 12)}
                    want = HL.group_actions_np(types[r].numpy(), vis_np[r], int(ex[r]), int(app[r]), K, noop, (11, 12),
                                               _picks(cfg, int(r), int(ticks[r]), counts, (11, 12)))
                    assert got["n_groups"][r] == len(want), (M, role, r)
                    used = 0
                    for gi, (at, exs, devs, ap) in enumerate(want):
                        assert got["atype"][r, gi] == at and got["n_exploit"][r, gi] == 1 and got["exploit"][r, gi, 0] == exs[0] and got["app"][r, gi] == ap
                        assert got["dev_cnt"][r, gi] == len(devs) and list(got["dev_idx"][r, used: used + len(devs)]) == devs, (M, role, r, gi)
                        used += len(devs)
                others = np.setdiff1d(np.arange(N), rows.numpy())
                assert (got["n_groups"][others] == -9).all()
        # capacity: more groups than max_groups / more devices than max_devs are cut and flagged
        small = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=2, max_devs=4)
        small.group_actions(None, (torch.arange(M) % 6)[None, :].repeat(N, 1).to("cuda:0"), None, None, "defender", visible=torch.ones((N, M), dtype=torch.uint8, device="cuda:0"))
        assert small.take_status() & abi.DECODE_TRUNCATED
        a = {k: v.cpu().numpy() for k, v in small.act.items()}
        assert (a["n_groups"] == 2).all() and (a["dev_cnt"].sum(axis=1) <= 4).all()
        small.close()
        # the attacker's default: exactly its 3 action types (get_num_action_types('attacker') == 3; the no-op, 3, lies outside):
        # type ids above 2 in a caller's tensor must never be grouped and stepped as attacker actions
        types = (torch.arange(M) % 7)[None, :].repeat(N, 1)
        env.group_actions(None, types.to("cuda:0"), None, None, "attacker", visible=torch.ones((N, M), dtype=torch.uint8, device="cuda:0"))
        a = {k: v.cpu().numpy() for k, v in env.act.items()}
        assert (a["n_groups"] == 3).all() and sorted(set(a["atype"][:, :3].reshape(-1).tolist())) == [0, 1, 2]
        env.close()


@pytest.mark.gpu
def test_sample_group_actions_kernel_equals_the_numpy_sampler():
    """cygym_sample_group_actions: per-device Categorical samples by the inverse CDF of softmax(logits) walked with the
    addressed Philox draw, their summed log-probability over the visible devices (+ exploit + app), then the grouping --
    against a float64 numpy restatement (rows whose every draw is clear of a CDF boundary compare exactly; the
    log-probabilities to 1e-4), plus the sample frequencies against softmax, and greedy = arg-max."""
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    for M, N, role, K, noop in ((64, 48, "defender", 14, 8), (100, 20, "attacker", 5, 3)):
        topo, init, ck = make_topology(M, 4 if M == 64 else 1, seed=4, n_active=M - 4)
        cfg = abi.EnvConfig(seed=123456789012, env_id_base=40, **ck)
        env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=14, max_devs=M)
        env.randomize()
        st = env.state_numpy()
        vis = _np_visibility(st["flags"], role)
        ticks = st["ienv"][:, S.I_RNG_TICK]
        g = torch.Generator().manual_seed(M)
        E, A = cfg.max_exploits, 4
        logits = torch.randn((N, M, K), generator=g) * 1.5
        el, al = torch.randn((N, E), generator=g), torch.randn((N, A), generator=g)
        types, ex, ap, logp = env.sample_group_actions(None, logits.to("cuda:0"), el.to("cuda:0"), al.to("cuda:0"), role, noop=noop)
        types, ex, ap, logp = types.cpu().numpy(), ex.cpu().numpy(), ap.cpu().numpy(), logp.cpu().numpy()
        got = {k: v.cpu().numpy() for k, v in env.act.items()}

        def head(l, u32):
            """(sample, log-probability, clear of every CDF boundary?) in float64"""
            l = l.astype(np.float64)
            p = np.exp(l - l.max())
            cdf = np.cumsum(p)
            target = (u32 / 4294967296.0) * cdf[-1]
            k = int(np.searchsorted(cdf, target, side="right"))
            k = min(k, len(l) - 1)
            return k, l[k] - l.max() - np.log(cdf[-1]), bool(np.abs(cdf - target).min() > 1e-4 * cdf[-1])

        n_checked = 0
        for e in range(N):
            want_t = np.zeros(M, np.int64)
            lp, clear = 0.0, True
            env_g = cfg.env_id_base + e
            for d in np.nonzero(vis[e])[0]:
                k, l1, c = head(logits[e, d].numpy(), R.draw(cfg.seed, env_g, int(ticks[e]), S.SITE_SAMPLE, int(d), 0))
                want_t[d] = k; lp += l1; clear &= c
            ke, l1, c = head(el[e].numpy(), R.draw(cfg.seed, env_g, int(ticks[e]), S.SITE_SAMPLE, 0, 1)); lp += l1; clear &= c
            ka, l1, c = head(al[e].numpy(), R.draw(cfg.seed, env_g, int(ticks[e]), S.SITE_SAMPLE, 0, 2)); lp += l1; clear &= c
            assert (types[e][~vis[e]] == 0).all()
            if not clear:
                continue
            n_checked += 1
            np.testing.assert_array_equal(types[e], want_t, err_msg=f"M={M} env {e}")
            assert ex[e] == ke and ap[e] == ka
            assert abs(logp[e] - lp) < 1e-4 * max(1.0, abs(lp)), (e, logp[e], lp)
            counts = {t: int((vis[e] & (want_t == t)).sum()) for t in (11, 12)}
            want = HL.group_actions_np(want_t, vis[e], ke, ka, K, noop, (11, 12), _picks(cfg, e, int(ticks[e]), counts, (11, 12)))
            assert got["n_groups"][e] == len(want)
            used = 0
            for gi, (at, exs, devs, a_) in enumerate(want):
                assert got["atype"][e, gi] == at and got["exploit"][e, gi, 0] == exs[0] and got["app"][e, gi] == a_ and got["dev_cnt"][e, gi] == len(devs)
                assert list(got["dev_idx"][e, used: used + len(devs)]) == devs
                used += len(devs)
        assert n_checked > N // 2
        # greedy = arg-max (first maximum), exactly
        tg, eg, ag, _ = env.sample_group_actions(None, logits.to("cuda:0"), el.to("cuda:0"), al.to("cuda:0"), role, noop=noop, greedy=True)
        np.testing.assert_array_equal(tg.cpu().numpy(), np.where(vis, logits.argmax(dim=-1).numpy(), 0))
        np.testing.assert_array_equal(eg.cpu().numpy(), el.argmax(dim=-1).numpy())
        np.testing.assert_array_equal(ag.cpu().numpy(), al.argmax(dim=-1).numpy())
        env.close()
    # frequencies: the same logits in every row and device, rng ticks moved on per env by stepping -> samples ~ softmax
    topo, init, ck = make_topology(64, 4, seed=4, n_active=60)
    cfg = abi.EnvConfig(seed=5, **ck)
    env = BatchedCyberDefenseEnv(topo, cfg, 2048, init, device="cuda:0", max_groups=14, max_devs=64)
    base = torch.tensor([0.0, 1.0, -1.0, 2.0, 0.5])
    lg = base[None, None, :].repeat(2048, 64, 1).contiguous().to("cuda:0")
    t, _, _, _ = env.sample_group_actions(None, lg, None, None, "attacker", noop=3)
    vis = env.visibility_mask("attacker").cpu().numpy() > 0.5
    tt = t.cpu().numpy()[vis]
    freq = np.bincount(tt, minlength=5) / tt.size
    p = torch.softmax(base, 0).numpy()
    assert tt.size > 5000 and np.abs(freq - p).max() < 4 * np.sqrt(p.max() / tt.size) + 2e-3, (freq, p)
    env.close()


class IntPerDeviceNet:
    """Per-device actor-critic with integer weights (exact in float32 everywhere): logits [N, M, K] from the device's own
    view row, unique arg-maxima; exploit / app logits and a value from sums of the view."""

    def __init__(self, role, M, K, E, seed):
        rs = np.random.RandomState(seed)
        self.F = 6 if role == "defender" else 4
        self.M, self.K, self.E = M, K, E
        self.w = torch.tensor(rs.randint(-2, 3, size=(self.F, K)), dtype=torch.float32)
        self.pos = torch.arange(K, dtype=torch.float32)
        self.dev_bias = torch.tensor(rs.randint(0, K, size=(M,)), dtype=torch.float32)

    def __call__(self, state, vis):
        dev = state.device
        x = state[:, : self.F * self.M].reshape(-1, self.M, self.F)
        z = x @ self.w.to(dev)                                                        # [N, M, K]
        onehot = torch.nn.functional.one_hot(self.dev_bias.long(), self.K).float().to(dev)   # a preferred type per device
        logits = (z + 3 * onehot[None]) * 16 + self.pos.to(dev)
        s = x.sum(dim=(1, 2))
        exp_logits = torch.stack([torch.remainder(s + e, 5) * 16 + e for e in range(self.E)], dim=1)
        app_logits = torch.stack([torch.remainder(s + 2 * a, 7) * 16 + a for a in range(4)], dim=1)
        return {"per_dev_type_logits": logits, "value": s, "exp_logits": exp_logits, "app_logits": app_logits}


@pytest.mark.gpu
@pytest.mark.parametrize("role,M,N,lam,max_extra", [("defender", 64, 24, 0.0, 0), ("attacker", 64, 24, 0.0, 0), ("defender", 100, 9, 0.7, 16),
                                                     ("attacker", 37, 17, 0.7, 0), ("defender", 256, 12, 0.7, 16), ("attacker", 256, 12, 0.0, 0)])
def test_collect_equals_the_reference_loop_on_the_oracle(role, M, N, lam, max_extra):
    """ippo_rollout.collect (greedy decisions: no sampling noise between CPU and GPU) against the loop of IPPO.py:503-624
    run env by env logic on the CPU oracle: turn from step_num, visibility mask, per-device arg-max types, the reference's
    grouping, env.step(groups), opponent turns, the episode cap with its reset + ownership reshuffle -- states, rewards,
    dones, recorded decisions and the final state bit for bit."""
    from oracle import driver as od
    import golden_io as gio
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.ippo_rollout import collect
    from cygym_amd.topology import make_topology
    n_dec = 14
    topo, init, ck = make_topology(M, 4 if M == 64 else 1, seed=8, n_active=M - 8, max_extra=max_extra)
    cfg = abi.EnvConfig(seed=8, auto_reset=1, episode_limit=11, lambda_events=lam, **ck)          # the cap falls inside the rollout
    X = cfg.max_exploits
    K = 14 if role == "defender" else X + 3
    noop = 8 if role == "defender" else 3
    other = "attacker" if role == "defender" else "defender"
    opp_seq = [(1, [0], [], 0), (2, [1], [], 0), (3, [0], [], 0)] if other == "attacker" else [(1, [0], [3, 9, 12], 0), (8, [0], [], 0), (6, [0], [1, 2], 0)]
    net = IntPerDeviceNet(role, M, K, X, 5 + M)
    batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=14, max_devs=M)
    ro = collect(batch, role, net, opp_seq, n_dec, greedy=True)
    assert ro.logp.shape == (n_dec, N) and ro.per_dev_types.shape == (n_dec, N, M)

    ob = od.OracleBatch(topo, cfg, N)
    ob.load_state(init)
    act = od.alloc_actions(N, 14, M)
    s, k = 0, 0
    while k < n_dec:
        turn = "defender" if s % 2 == 0 else "attacker"
        act["exploit"][:] = -1
        if turn == role:
            obs = torch.from_numpy(ob.observe(1 if role == "defender" else 2))
            vis = _np_visibility(ob.state["flags"], role)
            out = net(obs, torch.from_numpy(vis.astype(np.float32)))
            types = torch.argmax(out["per_dev_type_logits"], dim=-1).numpy()
            types = np.where(vis, types, 0)
            ex = torch.argmax(out["exp_logits"], dim=-1).numpy()
            ap = torch.argmax(out["app_logits"], dim=-1).numpy()
            ticks = ob.state["ienv"][:, S.I_RNG_TICK]
            for e in range(N):
                counts = {t: int((vis[e] & (types[e] == t)).sum()) for t in (11, 12)}
                groups = HL.group_actions_np(types[e], vis[e], int(ex[e]), int(ap[e]), K, noop, (11, 12), _picks(cfg, e, int(ticks[e]), counts, (11, 12)))
                HL.encode_into(act, e, role, groups, True, M)
            np.testing.assert_array_equal(ro.state[k].cpu().numpy(), obs.numpy(), err_msg=f"state at decision {k}")
            np.testing.assert_array_equal(ro.per_dev_types[k].cpu().numpy(), types, err_msg=f"types at decision {k}")
            np.testing.assert_array_equal(ro.vis_mask[k].cpu().numpy() > 0.5, vis)
            np.testing.assert_array_equal(ro.exp[k].cpu().numpy(), ex)
            np.testing.assert_array_equal(ro.value[k].cpu().numpy(), out["value"].numpy())
        else:
            a = opp_seq[s % len(opp_seq)]                                          # strat.actions[t % len] with the global tick (IPPO.py:399-402)
            for e in range(N):
                HL.encode_into(act, e, turn, [a], False, M)
        _, raw, shaped, done = ob.step(act)
        if turn == role:
            np.testing.assert_allclose(ro.raw_reward[k].cpu().numpy(), raw, rtol=0, atol=1e-9, err_msg=f"reward at decision {k}")
            np.testing.assert_array_equal(ro.done[k].cpu().numpy(), done != 0)
            k += 1
        s += 1
        if s > cfg.episode_limit:
            assert (done != 0).all()
            s = 0
            ob.randomize()
    # (the cap tick -- step_num 11 -> 12 -- is an attacker turn: only the attacker records a done, like in the reference)
    assert bool(ro.done.any()) == (role == "attacker") and not bool(ro.done.all())
    got = batch.state_numpy()
    got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
    assert not gio.compare_state(got, ob.state, f"collect {role}")
    np.testing.assert_array_equal(ro.last_state.cpu().numpy(), ob.observe(1 if role == "defender" else 2))
    batch.close()
