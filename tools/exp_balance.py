#!/usr/bin/env python3
"""Experiment: how much of a per-tick launch is imbalance between SIMDs / CUs?  The same multiset of actions per tick,
(a) in the script's random env order, (b) re-dealt so that heavy actions (spread / block / unblock) are evenly spaced
over the env ids (every workgroup and every SIMD gets its share)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from cygym_amd import abi
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
N, M, K = 4096, 256, 100
topo, init, ck = make_topology(M, 1, seed=0, max_extra=0)
cfg = abi.EnvConfig(seed=0, auto_reset=1, lambda_events=0.0, **ck)
env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M // 8)
def scripts(mode):
    out = []
    for t in range(K):
        act = {k: torch.empty_like(v) for k, v in env.act.items()}
        env.gen_actions(t, act)
        if mode != "random":
            at = act["atype"][:, 0]
            heavy = ((at == 1) if t % 2 else ((at == 6) | (at == 9))).to(torch.int64)
            if mode == "spaced":
                # deal: heavies to positions spaced evenly, lights fill the rest
                order = torch.argsort(heavy, descending=True, stable=True)          # heavy env ids first
                nh = int(heavy.sum())
                pos_h = (torch.arange(nh, device=at.device) * N // max(nh, 1))
                taken = torch.zeros(N, dtype=torch.bool, device=at.device); taken[pos_h] = True
                pos_l = torch.nonzero(~taken).flatten()
                dest = torch.cat([pos_h, pos_l])                                     # action of env order[i] goes to env dest[i]
            else:  # "clumped": all heavies first (worst case)
                order = torch.argsort(heavy, descending=True, stable=True); dest = torch.arange(N, device=at.device)
            for k in act:
                v = act[k].clone(); act[k][dest] = v[order]
        out.append(act)
    return out
for mode in ("random", "spaced", "clumped", "random"):
    sc = scripts(mode)
    env.load_state(init)
    for t in range(10): env.step(sc[t])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        env.load_state(init)
        torch.cuda.synchronize()
        e0.record()
        for t in range(K): env.step(sc[t])
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / K)
    print(f"{mode:8s}: {best:.2f} us/tick")
