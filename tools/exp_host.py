"""Is per-tick stepping host-bound?  Host time to ENQUEUE K cygym_step launches (no sync) against the GPU time they take.
    [CYGYM_SO=...] python tools/exp_host.py [M] [N] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from cygym_amd import abi
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 200
topo, init, ck = make_topology(M, {64: 4, 256: 1, 2048: 32}.get(M, 1), seed=0, max_extra=0)
cfg = abi.EnvConfig(seed=0, auto_reset=1, lambda_events=0.0, **ck)
env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=max(1, M // 8))
scripts = []
for t in range(K):
    a = {k: torch.empty_like(v) for k, v in env.act.items()}; env.gen_actions(t, a); scripts.append(a)
for t in range(K): env.step(scripts[t])
torch.cuda.synchronize()
for rep in range(3):
    env.load_state(init); torch.cuda.synchronize()
    t0 = time.perf_counter()
    env.timer_start()
    for t in range(K): env.step(scripts[t])
    t1 = time.perf_counter()
    ms = env.timer_stop()
    t2 = time.perf_counter()
    print(f"K={K}: host enqueue {1e6 * (t1 - t0) / K:.2f} us/step; GPU (events) {ms * 1e3 / K:.2f} us/step; wall incl. sync {1e6 * (t2 - t0) / K:.2f} us/step")
# the same K launches replayed from a HIP graph: no host in the loop
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for t in range(K): env.step(scripts[t])
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for t in range(K): env.step(scripts[t])
torch.cuda.synchronize()
for rep in range(3):
    env.load_state(init); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"HIP graph replay of the same {K} launches: {e0.elapsed_time(e1) * 1e3 / K:.2f} us/step")
