"""Time the reference's own step() on THIS container's host cores.

Run in the build container only (the reference never travels):
    PYTHONDONTWRITEBYTECODE=1 python oracle/harness/time_reference.py [--procs 8] [--seconds 8]
Unmodified reference source, plain MT19937 draws (no injection), the harness' stand-ins for the absent
gym / igraph / pymetis, alternating defender / attacker actions drawn like the golden scenarios.
Prints one JSON line per network size: steps/s of one process and of `procs` independent processes.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def worker(args):
    M, n_active, seconds, seed = args
    import ref_harness as H
    import make_golden as G
    env = H.build_env(M, n_active, init_seed=seed, strip_vuln_frac=0.5, extra_reachable=3, prewarm_star=True)
    fn = G.mixed_actions(M, G.ALL_DEF, G.ALL_ATT, max(1, M // 8))
    rs = np.random.RandomState(seed)
    for t in range(20):   # warm-up
        mode, action = fn(0, t, env, rs)
        env.mode = "defender" if mode == 0 else "attacker"
        env.step(action)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        mode, action = fn(0, n, env, rs)
        env.mode = "defender" if mode == 0 else "attacker"
        env.step(action)
        n += 1
    return n / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=min(8, os.cpu_count() or 1))
    ap.add_argument("--seconds", type=float, default=8.0)
    a = ap.parse_args()
    for M, n_active in ((16, 12), (64, 48), (256, 200)):
        one = worker((M, n_active, a.seconds, 1))
        with mp.get_context("fork").Pool(a.procs) as pool:
            many = pool.map(worker, [(M, n_active, a.seconds, 1 + i) for i in range(a.procs)])
        print(json.dumps({"devices": M, "reference_python_steps_per_s_1proc": round(one, 1),
                          f"reference_python_steps_per_s_{a.procs}procs": round(sum(many), 1),
                          "machine": f"build container host, {os.cpu_count()} cores"}), flush=True)


if __name__ == "__main__":
    main()
