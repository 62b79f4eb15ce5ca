"""Development loop on the GPU box: parity of a development build (tools/devbuild.py) against the oracle at one size,
then per-tick / rollout timings.   CYGYM_SO=cygym_amd/libcygym_dev.so python tools/quick.py [M] [N] [ticks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import golden_io as gio
from cygym_amd import abi, spec as S
from cygym_amd.actions import gen_actions_numpy
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
from oracle import driver as od

M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 40
blocks = {64: 4, 256: 1, 2048: 32}.get(M, 1)

def parity(max_extra, detector, n=192, ticks=120, lam=0.0):
    topo, init, ck = make_topology(M, blocks, seed=3, n_active=int(M * 0.9), max_extra=max_extra)
    if lam: ck.update(dict(lambda_events=lam, p_add=0.45, p_attacker=0.08, num_of_device=max(2, M // 3), min_network_size=2))
    cfg = abi.EnvConfig(seed=3, env_id_base=77, **ck)
    L = max(1, M // 8)
    env = BatchedCyberDefenseEnv(topo, cfg, n, init, device="cuda:0", max_groups=1, max_devs=L, detector=detector)
    ob = od.OracleBatch(topo, cfg, n, detector=detector); ob.load_state(init)
    if lam: env.randomize(); ob.randomize()
    for t in range(ticks):
        env.gen_actions(t)
        act = gen_actions_numpy(cfg.seed, cfg.env_id_base, n, M, topo.X, t, L)
        obs, raw, shaped, done = env.step(); o = ob.step(act)
        if t % 8 == 0 or t == ticks - 1:
            got = env.state_numpy(); got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
            bad = gio.compare_state(got, ob.state, f"t={t}")
            assert not bad, "\n".join(bad[:6])
            assert np.array_equal(obs.cpu().numpy(), o[0]) and np.allclose(raw.cpu().numpy(), o[1], rtol=0, atol=1e-9)
    # the same script as one rollout from the start
    env.load_state(init)
    if lam: env.state["ienv"][:, S.I_RNG_TICK] = 0; env.randomize()
    act, out = env.alloc_rollout(ticks); env.gen_actions_rollout(0, act); env.rollout(act, out)
    got = env.state_numpy(); got["ienv"] = got["ienv"].copy(); got["ienv"][:, S.I_FLAGS] &= ~0x80
    bad = gio.compare_state(got, ob.state, "rollout")
    assert not bad, "\n".join(bad[:6])
    env.close()
    print(f"parity ok: M={M} max_extra={max_extra} detector={detector} lambda={lam}", flush=True)

def timing():
    dev = torch.device("cuda:0")
    mx = int(os.environ.get("QUICK_MAX_EXTRA", "0"))   # -1: room for two attacker stars (the full-feature kernels)
    topo, init, ck = make_topology(M, blocks, seed=0, max_extra=None if mx < 0 else mx)
    cfg = abi.EnvConfig(seed=0, auto_reset=1, lambda_events=0.0, **ck)
    env = BatchedCyberDefenseEnv(topo, cfg, N, init, device=dev, max_groups=1, max_devs=max(1, M // 8))
    W = 10
    scripts = []
    for t in range(W + K):
        a = {k: torch.empty_like(v) for k, v in env.act.items()}; env.gen_actions(t, a); scripts.append(a)
    for t in range(W): env.step(scripts[t])
    torch.cuda.synchronize()
    keep = {k: env.state[k].clone() for k in abi.BUFFER_FIELDS}
    act = {k: torch.stack([scripts[t][k] for t in range(W, W + K)]).contiguous() for k in scripts[0]}
    _, out = env.alloc_rollout(K)
    B = 16.0 * M + 24.0 * M + 2.0 * ((topo.E + 7) // 8) + (M / 8.0 + 16.0)
    for name, fn in (("per-tick", lambda: [env.step(scripts[t]) for t in range(W, W + K)]), ("rollout ", lambda: env.rollout(act, out))):
        ts = []
        for r in range(9):
            for k, v in keep.items(): env.state[k].copy_(v)
            torch.cuda.synchronize()
            env.timer_start(); fn(); ts.append(env.timer_stop())
        ms = float(np.median(ts))
        print(f"{name} N={N} M={M}: {ms / K * 1e3:.2f} us/tick  {N * K / (ms / 1e3):.3e} env-steps/s  frac {N * B * K / (ms / 1e3) / 8e12:.3f}  (min {min(ts) / K * 1e3:.2f} us)", flush=True)
    # defender ticks (even) and attacker ticks (odd) separately: one event pair per launch
    for k, v in keep.items(): env.state[k].copy_(v)
    torch.cuda.synchronize()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    evs[0].record()
    for i, t in enumerate(range(W, W + K)):
        env.step(scripts[t]); evs[i + 1].record()
    torch.cuda.synchronize()
    d = np.array([evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(K)])
    par = np.array([(W + i) & 1 for i in range(K)])
    print(f"   per launch: defender ticks {d[par == 0].mean():.2f} us (max {d[par == 0].max():.2f}), attacker ticks {d[par == 1].mean():.2f} us (max {d[par == 1].max():.2f})", flush=True)
    env.close()

if __name__ == "__main__":
    if not os.environ.get("QUICK_NO_PARITY"):
        parity(0, False); parity(M // 2, False, lam=1.6); parity(0, True)
    timing()
