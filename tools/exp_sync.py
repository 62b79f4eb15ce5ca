"""Where do the ~39 us between `K x launch time` and the wall clock of a K = 20 timed region go?  (host side: first-launch
latency from an idle queue, the wait for completion, extra synchronisations).   python tools/exp_sync.py [spin]"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(hipDeviceScheduleSpin) ->", hip.hipSetDeviceFlags(1))
import numpy as np, torch
from cygym_amd import abi
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
M, N, K = 256, 4096, 20
topo, init, ck = make_topology(M, 1, seed=0, max_extra=0)
cfg = abi.EnvConfig(seed=0, auto_reset=1, lambda_events=0.0, **ck)
env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=32)
scripts = []
for t in range(5 + K):
    a = {k: torch.empty_like(v) for k, v in env.act.items()}; env.gen_actions(t, a); scripts.append(a)
for _ in range(50):
    for t in range(5 + K): env.step(scripts[t])
torch.cuda.synchronize()
def run(mode):
    ws, es = [], []
    for rep in range(15):
        env.load_state(init)
        for t in range(5): env.step(scripts[t])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.timer_start()
        for t in range(5, 5 + K): env.step(scripts[t])
        if mode == "bench":      # what bench.py did: event sync, then synchronize, then barrier (2 more synchronizes)
            ev = env.timer_stop(); torch.cuda.synchronize(); torch.cuda.synchronize(); torch.cuda.synchronize()
        else:                    # one synchronize, then read the events
            torch.cuda.synchronize(); t1 = time.perf_counter(); ev = env.timer_stop()
        if mode == "bench": t1 = time.perf_counter()
        ws.append((t1 - t0) * 1e6); es.append(ev * 1e3)
    print(f"{mode:6s}: wall {np.median(ws) / K:.2f} us/step, events {np.median(es) / K:.2f} us/step, fixed overhead {np.median(ws) - np.median(es):.1f} us per region")
run("bench"); run("one")
