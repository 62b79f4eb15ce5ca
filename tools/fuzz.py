#!/usr/bin/env python3
"""tools/fuzz.py -- differential fuzzing of the HIP tick against the CPU oracle (GPU box only).

    python tools/fuzz.py [--cases 40] [--seed0 0] [--ticks 300]

Each case draws a topology size, activity level, evolve parameters (events, additions, attacker-owned
activations), extra-edge capacity (including too small ones), list capacity, an optional ownership reshuffle and
episode cap, baseline mode and group capacity, then steps both sides with the synthetic script -- every third
tick with a hand-aimed block / unblock / clean on attacker-owned devices, some ticks as step_grouped() calls,
some envs sitting a tick out or taking a partial tick, in 40 % of the cases with host-side calls between ticks
(reset / reshuffle of a random subset, baseline and reward-scale changes, role observations) -- and compares the
whole state bit for bit every few ticks, plus a fused rollout of the same script at the end.  Prints one line per case; exits non-zero on the first mismatch.
TEST INFRASTRUCTURE (uses oracle/): not part of the product path.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed0", type=int, default=0)
    ap.add_argument("--ticks", type=int, default=300)
    ap.add_argument("--envs", type=int, default=0, help="fix the batch size (default: 33..96, or 9..17 above 520 devices)")
    ap.add_argument("--sizes", default="13,16,24,37,64,100,130,200,256,300,520", help="device counts to draw from")
    a = ap.parse_args()
    sizes = [int(x) for x in a.sizes.split(",")]
    import torch
    import golden_io as gio
    from cygym_amd import abi, spec as S
    from cygym_amd.actions import gen_actions_numpy
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    from oracle import driver as od

    t_start = time.time()
    for case in range(a.seed0, a.seed0 + a.cases):
        rs = np.random.RandomState(1000 + case)
        M = int(rs.choice(sizes))
        blocks = (int(rs.choice([1, 1, 2, 4])) if M >= 16 else 1) * (8 if M >= 1024 else 1)
        n_active = int(rs.randint(max(3, M // 3), M + 1))
        K = int(rs.choice([0, 4, 16, 64, 128, 256]))
        N = int(rs.choice([33, 64, 96])) if M <= 520 else int(rs.choice([9, 17]))
        if a.envs:
            N = a.envs
        L = int(rs.choice([1, 2, max(1, M // 8), max(2, M // 4) & ~1, 7]))
        G = int(rs.choice([1, 1, 3]))                      # > 1: some ticks are step_grouped() calls
        baseline = str(rs.choice(["Nash", "Nash", "Nash", "No Defense", "Preset", "No Attack"]))
        ticks = min(a.ticks, 120 if M > 256 else a.ticks)
        topo, init, ck = make_topology(M, blocks, seed=case, n_active=n_active, max_extra=K)
        ck.update(dict(lambda_events=float(rs.choice([0.0, 0.7, 1.5, 3.0])), p_add=float(rs.choice([0.1, 0.4, 0.8])),
                       p_attacker=float(rs.choice([0.0, 0.05, 0.3])),
                       num_of_device=int(rs.randint(2, max(3, n_active))), min_network_size=2,
                       episode_limit=int(rs.choice([1000, 37])), auto_reset=int(rs.rand() < 0.5),
                       zero_day=int(rs.rand() < 0.2), zero_day_owned_mask=int(rs.randint(0, 4)),
                       fast_scan=int(rs.rand() >= 0.12),      # 0: the per-log scan path (history + anomaly planes, full-feature kernels)
                       turbo=int(rs.rand() < 0.2), turbo_ramp_steps=int(rs.choice([200, 40, 1])),
                       turbo_fraction_clients=float(rs.choice([0.05, 0.13, 0.5])), workload_period_base=int(rs.choice([50, 50, 4]))))
        cfg = abi.EnvConfig(seed=int(rs.randint(1 << 30)), env_id_base=int(rs.randint(1 << 20)), baseline=baseline, **ck)
        det = rs.rand() < 0.3 or not ck["fast_scan"]   # trained-detector mode: action 10 -> host fit -> scans walk the forest (full-feature kernels)
        env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=G, max_devs=L, detector=det)
        fused = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=G, max_devs=L, detector=det)
        ob = od.OracleBatch(topo, cfg, N, detector=det)
        ob.load_state(init)
        shuffle = rs.rand() < 0.6
        if shuffle:
            for b in (env, fused, ob):
                b.randomize()
        script = []
        n_trained = 0
        meddle = det or rs.rand() < 0.4      # host-side calls between ticks (then no fused replay of the script)
        for t in range(ticks):
            if meddle and t and t % 23 == 0:   # reset / reshuffle a random subset, flip the baseline, like do_agent.py:189-196
                ids = np.flatnonzero(rs.rand(N) < 0.3).astype(np.int32)
                if ids.size:
                    what = int(rs.randint(3))
                    if what == 0:
                        env.reset(ids); ob.reset(ids)
                    elif what == 1:
                        env.randomize(ids); ob.randomize(ids)
                    else:
                        import dataclasses
                        cfg = dataclasses.replace(cfg, baseline=str(rs.choice(["Nash", "No Defense", "Preset", "No Attack"])),
                                                  comp_scale=float(rs.choice([50.0, 10.0])))
                        env.set_config(cfg); ob.cfg = cfg
                for role in (1, 2):
                    if not np.allclose(env.observe(role).cpu().numpy(), ob.observe(role), rtol=0, atol=0.0 if ck["fast_scan"] else 1e-6):
                        print(f"case {case}: role-{role} observation differs at tick {t}")
                        sys.exit(1)
            act = gen_actions_numpy(cfg.seed, cfg.env_id_base, N, M, topo.X, t, L)
            if t % 3 == 0:
                fl = ob.state["flags"]
                for e in range(0, N, 2):
                    if act["mode"][e] != S.MODE_DEFENDER:
                        continue
                    owned = np.flatnonzero(fl[e] & S.F_OWNED)
                    if owned.size:
                        k = min(L, owned.size)
                        pick = owned[rs.permutation(owned.size)[:k]]
                        if rs.rand() < 0.3 and k > 1:
                            pick[1] = pick[0]           # a repeated device
                        act["atype"][e, 0] = int(rs.choice([6, 6, 9, 9, 1, 7, 13] + ([10, 10, 5, 5, 5] if det else [])))
                        act["dev_cnt"][e, 0] = k
                        act["dev_idx"][e, :k] = pick
            if G > 1:   # widen the single-action script to G groups; every 4th tick some defenders call step_grouped
                wide = od.alloc_actions(N, G, L)
                for k in ("mode", "n_groups", "dev_idx"):
                    wide[k][...] = act[k]
                for k in ("atype", "n_exploit", "app", "dev_cnt"):
                    wide[k][:, 0] = act[k][:, 0]
                wide["exploit"][:, 0] = act["exploit"][:, 0]
                if t % 4 == 1:
                    for e in range(1, N, 3):
                        if wide["mode"][e] != S.MODE_DEFENDER:
                            continue
                        g = int(rs.randint(1, G + 1))
                        wide["n_groups"][e] = g
                        used = 0
                        for j in range(g):
                            wide["atype"][e, j] = int(rs.choice([1, 2, 3, 10, 11, 8, 0, 1]))
                            c = int(rs.randint(0, max(1, (L - used)) + 1)) if used < L else 0
                            wide["dev_cnt"][e, j] = c
                            wide["dev_idx"][e, used:used + c] = rs.randint(0, M, size=c)
                            used += c
                act = wide
            if t % 7 == 3:      # some envs sit this tick out, some take a partial tick (step(action, agent_cnt))
                idle = rs.rand(N) < 0.15
                act["n_groups"][idle] = -1
                part = (rs.rand(N) < 0.1) & (act["n_groups"] == 0)
                act["mode"][part] |= S.MODE_PARTIAL
            script.append({k: v.copy() for k, v in act.items()})
            env.set_actions_numpy(act)
            obs, raw, shaped, done = env.step()
            o_obs, o_raw, o_shaped, o_done = ob.step(act)
            bad = []
            if det:   # the host's part of Detector.train: fit on the device-side history ring, hand the same forest to the oracle
                pend = np.flatnonzero(ob.state["ienv"][:, S.I_FLAGS] & S.E_DET_PENDING)
                n_fit = env.service_detectors()
                n_trained += n_fit
                if n_fit != pend.size:
                    bad.append(f"{n_fit} forests fitted, oracle has {pend.size} pending")
                if pend.size:     # one device-to-host copy for all of them
                    fo = env.state["forest"][torch.from_numpy(pend).to("cuda:0")].cpu().numpy().view(np.uint32)
                    for j, e in enumerate(pend):
                        ob.install_forest(int(e), fo[j])
            if not np.allclose(raw.cpu().numpy(), o_raw, rtol=0, atol=1e-9):
                bad.append("raw reward")
            if t % 5 == 0 or t == ticks - 1:
                got = env.state_numpy()
                got["ienv"] = got["ienv"].copy()
                got["ienv"][:, S.I_FLAGS] &= ~0x80
                bad += gio.compare_state(got, ob.state, f"t={t}")
                if not np.array_equal(got["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF, ob.state["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF):
                    bad.append("TOPO_OVF flags")
                if not (np.array_equal(obs.cpu().numpy(), o_obs) if ck["fast_scan"] else np.allclose(obs.cpu().numpy(), o_obs, rtol=0, atol=1e-6)):
                    bad.append("obs")
                if not ck["fast_scan"] and not np.allclose(got["anomaly"], ob.state["anomaly"], rtol=0, atol=1e-6):
                    bad.append("anomaly scores")
            if bad:
                print(f"case {case}: MISMATCH at tick {t}: M={M} blocks={blocks} n_active={n_active} K={K} N={N} L={L} "
                      f"shuffle={shuffle} cfg={ck}\n  " + "\n  ".join(bad[:6]))
                sys.exit(1)
        if meddle:
            nxmax = int((ob.state["ienv"][:, S.I_FLAGS].astype(np.int64) >> S.E_NX_SHIFT).max())
            scans = int(ob.state["ienv"][:, S.I_SCAN_CNT].sum())
            unp = int(((ob.state["ienv"][:, S.I_FLAGS] & S.E_UNPINNED) != 0).sum())
            print(f"case {case}: ok  M={M} K={K} N={N} L={L} G={G} with host-side resets / reshuffles / config changes "
                  f"ticks={ticks} max_extra_edges={nxmax}" + (f" detector: {n_trained} forests fitted, {scans} scans, {unp} unpinned envs" if det else "")
                  + f" [{time.time() - t_start:.0f}s]", flush=True)
            env.close(); fused.close()
            continue
        # the same script as ONE fused rollout must land in the same state
        r_act, r_out = fused.alloc_rollout(ticks)
        for k in r_act:
            r_act[k].copy_(torch.from_numpy(np.stack([s[k] for s in script]).astype(r_act[k].cpu().numpy().dtype)).reshape(r_act[k].shape))
        fused.rollout(r_act, r_out, check=False)    # (a grouped action 10 followed by a scan IS part of the fuzz: both sides flag it)
        fa, fb = fused.state_numpy(), env.state_numpy()
        for k in ("live", "stash", "blocked", "blocked_in", "ring", "ienv", "fenv"):
            if not np.array_equal(fa[k], fb[k]):
                print(f"case {case}: fused rollout differs from stepping in {k}: M={M} K={K} L={L} cfg={ck}")
                sys.exit(1)
        if K > 0:
            nx = (fb["ienv"][:, S.I_FLAGS].astype(np.int64) >> S.E_NX_SHIFT)
            for e in range(N):
                if not np.array_equal(fa["extra"][e, :nx[e]], fb["extra"][e, :nx[e]]):
                    print(f"case {case}: fused rollout extra-edge list differs (env {e})")
                    sys.exit(1)
        nxmax = int((ob.state["ienv"][:, S.I_FLAGS].astype(np.int64) >> S.E_NX_SHIFT).max())
        ovf = int(((ob.state["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF) != 0).sum())
        print(f"case {case}: ok  M={M} b={blocks} act={n_active} K={K} N={N} L={L} shuffle={int(shuffle)} lam={ck['lambda_events']} "
              f"p_att={ck['p_attacker']} G={G} bl={baseline!r} cap={ck['episode_limit']}/{ck['auto_reset']} ticks={ticks} max_extra_edges={nxmax} ovf_envs={ovf} "
              f"[{time.time() - t_start:.0f}s]", flush=True)
        env.close(); fused.close()
    print("fuzz: all cases agree")


if __name__ == "__main__":
    main()
