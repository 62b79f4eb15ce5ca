#!/usr/bin/env python3
"""tools/exp_closed_loop.py -- the closed-loop grid consumer alone (for rocprofv3 --kernel-trace --stats):
    python tools/exp_closed_loop.py [--grid 2x2] [--ticks 206] [--hidden 64] [--eager] [--sub S]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cygym_amd import abi
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.policies import ActorPolicy, mlp_actor, calibrate_device_head
from cygym_amd.rollout_grid import simulate_grid
from cygym_amd.topology import make_topology

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="2x2"); ap.add_argument("--ticks", type=int, default=206)
ap.add_argument("--hidden", type=int, default=64); ap.add_argument("--eager", action="store_true")
ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--M", type=int, default=256)
ap.add_argument("--frac", type=float, default=1 / 16)
ap.add_argument("--no-randomize", action="store_true"); ap.add_argument("--maxdevs", type=int, default=0)
ap.add_argument("--eps", type=float, default=1.0); ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--no-mlp", action="store_true", help="actor body in torch, only the last layer in the decode launch")
ap.add_argument("--no-state", action="store_true", help="fused actor reads the role-view tensor instead of building the view from the state")
ap.add_argument("--no-merge", action="store_true", help="tick and next actor as two launches")
ap.add_argument("--loop-only", action="store_true", help="stop after the timed loops (profiling: no split, no script-stepping section)")
a = ap.parse_args()
nD, nA = (int(x) for x in a.grid.split("x"))
M = a.M
topo, init, ck = make_topology(M, {64: 4, 2048: 32}.get(M, 1), seed=0, max_extra=0)
cfg = abi.EnvConfig(seed=0, lambda_events=0.0, **ck)
X = cfg.max_exploits
dt, at = [1, 4, 5, 6, 7, 8, 9, 11, 12, 13, 2], [1, 2, 3]
n_mc = a.envs // (nD * nA)
N = nD * nA * n_mc
batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=a.maxdevs or M)
Dp = [ActorPolicy(mlp_actor(6 * M, len(dt) + M + X + 4, (a.hidden,), seed=100 + i, device="cuda:0"), len(dt), X, 4, type_map=dt, epsilon=a.eps) for i in range(nD)]
Ap = [ActorPolicy(mlp_actor(4 * M + X, len(at) + M + X, (a.hidden,), seed=200 + j, device="cuda:0"), len(at), X, 0, type_map=at, epsilon=a.eps) for j in range(nA)]
for p, role in [(p, 1) for p in Dp] + [(p, 2) for p in Ap]:
    calibrate_device_head(p, batch.observe(role), M, a.frac)
    p.fuse_mlp, p.from_state = not a.no_mlp, not a.no_state
simulate_grid(batch, Dp, Ap, n_mc, 16, graph=not a.eager, streams=a.streams, merge_launches=not a.no_merge)
for rep in range(3):
    tm = {}
    simulate_grid(batch, Dp, Ap, n_mc, a.ticks, graph=not a.eager, timers=tm, streams=a.streams, randomize=not a.no_randomize, merge_launches=not a.no_merge)
    print(f"grid {a.grid} hidden {a.hidden} mlp={not a.no_mlp} state={not a.no_state} merge={not a.no_merge} graph={tm['graph']} streams={tm['streams']}: {tm['loop_s'] / a.ticks * 1e6:.1f} us/tick, {N * a.ticks / tm['loop_s']:.3e} env-steps/s, "
          f"mean list {float(batch.act['dev_cnt'].float().mean()):.1f}")
if a.loop_only:
    sys.exit(0)
tm = {"split": True}
simulate_grid(batch, Dp, Ap, n_mc, 60, timers=tm, randomize=not a.no_randomize)
print({k: round(tm[k] / 60 * 1e6, 1) for k in ("observe", "policy+scatter", "step")})

# the synthetic script of bench.py stepped on the same batch (same max_devs), from the initial state: per-tick stepping reference
import time
for rnd in (False, True):
    batch.reset()
    if rnd:
        batch.randomize()
    L = batch.act["dev_idx"].shape[1]
    scripts = []
    for t in range(100):
        act = {k: torch.empty_like(v) for k, v in batch.act.items()}
        batch.gen_actions(t, act)
        scripts.append(act)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for t in range(100):
        batch.step(scripts[t])
    e1.record(); torch.cuda.synchronize()
    print(f"synthetic script on this batch (randomize={rnd}): {e0.elapsed_time(e1) * 10:.1f} us/tick; comp devices/env {float((batch.state['flags'] & 1).float().sum(1).mean()):.1f}")
