"""Host-side cost of one step() call (Python + ctypes + hipLaunchKernel), measured by issuing launches faster than the
GPU retires them is impossible -- so: time K calls of a 1-env batch (kernel ~ 10 us) back to back and K calls of
cygym_step on an env range of length 0 (no launch: pure host path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from cygym_amd import abi
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
topo, init, ck = make_topology(256, 1, seed=0, max_extra=0)
env = BatchedCyberDefenseEnv(topo, abi.EnvConfig(seed=0, **ck), 4096, init, device="cuda:0", max_groups=1, max_devs=32)
acts = []
for t in range(8):
    a = {k: torch.empty_like(v) for k, v in env.act.items()}; env.gen_actions(t, a); acts.append(a)
torch.cuda.synchronize()
K = 2000
for name, fn in (("step_range(0, 0): host path only, no launch", lambda i: env.step_range(0, 0, acts[i & 7])),
                 ("step(): host path + launch enqueue", lambda i: env.step(acts[i & 7]))):
    t0 = time.perf_counter()
    for i in range(K): fn(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{name}: {(t1 - t0) / K * 1e6:.2f} us per call on the host ({(t2 - t0) / K * 1e6:.2f} us per call until the GPU drained)")
