"""Stage sub-stamps of a stamps build (GPU box): python tools/exp_stage.py <lib.so> [envs] [M] [ticks]
stamp0 (after env_setup) -> loads issued -> own loads back -> behind the barrier, cycles, medians / p90 over env-ticks."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cygym_amd import abi, _lib
_lib.SO = os.path.abspath(sys.argv[1])
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
M = int(sys.argv[3]) if len(sys.argv) > 3 else 256
T = int(sys.argv[4]) if len(sys.argv) > 4 else 30
W = 28
topo, init, ck = make_topology(M, {64: 4, 256: 1, 2048: 32}.get(M, 1), seed=0, max_extra=0)
cfg = abi.EnvConfig(seed=0, auto_reset=1, lambda_events=0.0, **ck)
env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=max(1, M // 8))
env.lib.cygym_set_debug.argtypes = [C.c_void_p, C.c_void_p]
dbg = torch.zeros((N, W), dtype=torch.int64, device="cuda:0")
env.lib.cygym_set_debug(env._h, C.c_void_p(dbg.data_ptr()))
rows = []
for t in range(T):
    env.gen_actions(t); dbg.zero_(); env.step(); torch.cuda.synchronize()
    if t >= 5: rows.append(dbg.cpu().numpy().copy())
d = np.concatenate(rows)
pre = d[:, 9] >> 8
q = lambda x: "p50 %6d  p90 %6d  p99 %6d" % tuple(np.percentile(x, [50, 90, 99]))
print(os.path.basename(sys.argv[1]), f"{N} x {M}")
print("  entry -> stamp0 (kernarg block, env_setup)   ", q(pre))
print("  stamp0 -> state loads issued                 ", q(d[:, 24] - d[:, 0]))
print("  issued -> all of this wave's loads back      ", q(d[:, 25] - d[:, 24]))
print("  back -> behind the barrier (slowest of 16)   ", q(d[:, 1] - d[:, 25]))
print("  stamp0 -> behind the barrier                 ", q(d[:, 1] - d[:, 0]))
