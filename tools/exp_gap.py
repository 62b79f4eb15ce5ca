"""Diagnostic (stamps build): what separates two CONSECUTIVE back-to-back launches?  K launches are enqueued without a
sync; the stamp buffer then holds the wave entry / end times (s_memrealtime, 10 ns) of the last launch (slots 19 / 20) and of
the last ODD-tick launch (slots 22 / 23).  With K even the last launch is an odd tick -> K odd makes the last launch an even
tick, whose predecessor is the odd-tick launch.
    TAIL_SO=... python tools/exp_gap.py [M] [N] [K]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cygym_amd import abi, _lib
_lib.SO = os.environ.get("TAIL_SO") or os.path.join(ROOT, "cygym_amd", "libcygym_hip_stamps.so")
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
M = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 41
topo, init, ck = make_topology(M, {64: 4, 256: 1, 2048: 32}.get(M, 1), seed=0, max_extra=0)
cfg = abi.EnvConfig(seed=0, auto_reset=1, lambda_events=0.0, **ck)
env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=max(1, M // 8))
env.lib.cygym_set_debug.argtypes = [C.c_void_p, C.c_void_p]
dbg = torch.zeros((N, 28), dtype=torch.int64, device="cuda:0")
env.lib.cygym_set_debug(env._h, C.c_void_p(dbg.data_ptr()))
scripts = []
for t in range(K):
    a = {k: torch.empty_like(v) for k, v in env.act.items()}; env.gen_actions(t, a); scripts.append(a)
res = []
for rep in range(6):
    env.load_state(init); torch.cuda.synchronize()
    for t in range(K): env.step(scripts[t], full_obs=not os.environ.get('EXP_NO_OBS'))
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    a0, a1 = d[:, 22], d[:, 23]      # the last odd tick (attacker turn: rng tick odd AFTER the increment = even tick index ... see print)
    b0, b1 = d[:, 19], d[:, 20]      # the last launch
    res.append(((a1.max() - a0.min()) * 10, (b0.min() - a1.max()) * 10, (b1.max() - b0.min()) * 10, (b0.min() - a0.min()) * 10))
r = np.array(res[1:], dtype=np.float64)
print(f"K={K}: previous launch span {r[:, 0].mean():.0f} ns | gap (its last wave end -> next launch's first wave entry) {r[:, 1].mean():.0f} ns | last launch span {r[:, 2].mean():.0f} ns | period (first entry to first entry) {r[:, 3].mean():.0f} ns")
