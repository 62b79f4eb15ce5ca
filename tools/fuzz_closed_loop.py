#!/usr/bin/env python3
"""tools/fuzz_closed_loop.py -- differential fuzzing of the closed-loop grid consumer (GPU box only): simulate_grid with
actor-network strategies on the product batch (cygym_actor_mlp_decode: whole actor + decode + scatter in one launch, the
role view built on chip; populations via n_groups; HIP-graph replay) against the same strategies on the CPU oracle, where
every actor runs in torch and is decoded with the torch fallback.

    python tools/fuzz_closed_loop.py [--cases 40] [--seed0 0]

Each case draws the device count, the grid shape, the Monte-Carlo count (populations need a multiple of 16 rows per
strategy; other counts take one launch per strategy), the actors' depth and widths, epsilon-free integer weights (exact in
float32 in any summation order), the horizon, evolve events, an ownership reshuffle, and which of the fused paths are
switched off (whole-actor launch / on-chip view / graph replay).  Compares both payoff matrices (1e-9) and the whole final
state bit for bit.  Prints one line per case; exits non-zero on the first mismatch.
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


def int_actor(state_dim, action_dim, widths, seed, device):
    """Linear-ReLU stack with integer weights in {-1, 0, 1} (hidden layers) and multiples of 16 plus a position bias (last
    layer): every value a small integer, every arg-max unique."""
    import torch
    from cygym_amd.policies import mlp_actor
    net = mlp_actor(state_dim, action_dim, tuple(widths), seed=seed)
    rs = np.random.RandomState(seed)
    lin = [m for m in net if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for l, m in enumerate(lin[:-1]):
            # (sparse deeper layers keep the sums far below 2^24)
            w = rs.randint(-1, 2, size=m.weight.shape) * (rs.rand(*m.weight.shape) < (0.3 if l == 0 else 0.1))
            m.weight.copy_(torch.tensor(w, dtype=torch.float32))
            m.bias.copy_(torch.tensor(rs.randint(-2, 3, size=m.bias.shape), dtype=torch.float32))
        lin[-1].weight.copy_(torch.tensor(rs.randint(-1, 2, size=lin[-1].weight.shape) * 16 * (rs.rand(*lin[-1].weight.shape) < 0.3), dtype=torch.float32))
        lin[-1].bias.copy_(torch.arange(action_dim, dtype=torch.float32) - action_dim // 3)
    return net.to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed0", type=int, default=0)
    a = ap.parse_args()
    import torch
    import golden_io as gio
    from grid_util import OracleGrid
    from cygym_amd import abi, spec as S
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.policies import ActorPolicy
    from cygym_amd.rollout_grid import simulate_grid
    from cygym_amd.topology import make_topology
    t0 = time.time()
    for case in range(a.seed0, a.seed0 + a.cases):
        rs = np.random.RandomState(case)
        M = int(rs.choice([16, 24, 37, 64, 100, 130, 256, 256, 600]))      # (600: action vectors wider than 512)
        blocks = 4 if M == 64 else 1
        topo, init, ck = make_topology(M, blocks, seed=case, n_active=max(8, M - int(rs.randint(0, M // 4 + 1))), max_extra=int(rs.choice([0, 16])))
        cfg = abi.EnvConfig(seed=case, lambda_events=float(rs.choice([0.0, 0.7])), **ck)
        X = cfg.max_exploits
        nD, nA = int(rs.randint(1, 4)), int(rs.randint(1, 4))
        n_mc = int(rs.choice([16, 16, 32, 5, 19]))
        N = nD * nA * n_mc
        T = int(rs.randint(12, 41))
        n_hidden = int(rs.randint(1, 4))
        widths = [int(rs.choice([16, 32, 48, 64, 128])) for _ in range(n_hidden)]
        same_arch = bool(rs.rand() < 0.7)      # a population shares its architecture
        def_types = [1, 4, 5, 6, 7, 8, 9, 13, 2, 12, 11, 3]
        n_apps = int(rs.choice([0, 4]))
        randomize = bool(rs.rand() < 0.5)
        fuse_mlp, from_state, graph = bool(rs.rand() < 0.8), bool(rs.rand() < 0.8), bool(rs.rand() < 0.6)
        merge = bool(rs.rand() < 0.8)             # (tick + next actor as one launch where the shapes allow: 256 devices)

        def make(dev):
            def w(i):
                return widths if same_arch else [widths[0] if i % 2 == 0 else 16] + widths[1:]
            Dp = [ActorPolicy(int_actor(6 * M, len(def_types) + M + X + n_apps, w(i), 1000 * case + i, dev), len(def_types), X, n_apps, type_map=def_types)
                  for i in range(nD)]
            Ap = [ActorPolicy(int_actor(4 * M + X, 3 + M + X, w(j), 1000 * case + 100 + j, dev), 3, X, 0, type_map=[1, 2, 3]) for j in range(nA)]
            for p in Dp + Ap:
                p.fuse_mlp, p.from_state = fuse_mlp, from_state
            return Dp, Ap

        og = OracleGrid(topo, cfg, N, init, 1, M)
        E_def, E_att = simulate_grid(og, *make("cpu"), n_mc, T, randomize=randomize)
        batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M)
        U_def, U_att = simulate_grid(batch, *make("cuda:0"), n_mc, T, randomize=randomize, graph=graph, merge_launches=merge)
        got = batch.state_numpy()
        got["ienv"] = got["ienv"].copy()
        got["ienv"][:, S.I_FLAGS] &= ~0x80
        bad = gio.compare_state(got, og.ob.state, f"case {case}")
        ok = not bad and np.allclose(U_def, E_def, rtol=0, atol=1e-9) and np.allclose(U_att, E_att, rtol=0, atol=1e-9)
        print(f"case {case}: {'ok' if ok else 'MISMATCH'}  M={M} grid {nD}x{nA}x{n_mc} T={T} widths={widths} same_arch={same_arch} n_apps={n_apps} "
              f"randomize={randomize} lam={cfg.lambda_events} fuse_mlp={fuse_mlp} from_state={from_state} graph={graph} merge={merge} [{time.time() - t0:.0f}s]", flush=True)
        if not ok:
            print("\n".join(bad[:8]))
            print("U_def", U_def, "\nE_def", E_def)
            sys.exit(1)
        batch.close()
    print(f"{a.cases} cases agree")


if __name__ == "__main__":
    main()
