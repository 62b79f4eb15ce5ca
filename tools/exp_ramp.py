"""Diagnostic (stamps build): how a launch's waves start -- per XCD (HW_REG_XCC_ID) first / last wave entry relative to the
launch's first wave (s_memrealtime, 10 ns), and the order in which an XCD's workgroups start.
    TAIL_SO=cygym_amd/libcygym_exp_X.so python tools/exp_ramp.py [envs] [M] [ticks]"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cygym_amd import abi, _lib
_lib.SO = os.environ.get("TAIL_SO") or os.path.join(ROOT, "cygym_amd", "libcygym_hip_stamps.so")
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 20
WPB = int(os.environ.get("TAIL_WPB", "16"))
topo, init, ck = make_topology(M, {64: 4, 256: 1, 2048: 32}.get(M, 1), seed=0, max_extra=0)
cfg = abi.EnvConfig(seed=0, auto_reset=1, lambda_events=0.0, **ck)
env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=max(1, M // 8))
env.lib.cygym_set_debug.argtypes = [C.c_void_p, C.c_void_p]
dbg = torch.zeros((N, 28), dtype=torch.int64, device="cuda:0")
env.lib.cygym_set_debug(env._h, C.c_void_p(dbg.data_ptr()))
acc = []
for t in range(T):
    env.gen_actions(t); dbg.zero_(); env.step(); torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    if t < 5: continue
    rt0, rt1 = d[:, 19], d[:, 20]
    xcc = d[:, 21] & 0xF
    hw = d[:, 21] >> 8
    t0 = rt0.min()
    row = []
    for x in range(8):
        m = xcc == x
        row.append(((rt0[m].min() - t0) * 10, (rt0[m].max() - t0) * 10, (rt1[m].max() - t0) * 10, int(m.sum())))
    acc.append(row)
    if t == T - 1:
        wg = np.arange(N) // WPB
        print("last tick: workgroup id -> XCC of its first wave:", [int(xcc[w * WPB]) for w in range(16)], "...")
        x0 = xcc[0]
        ws = sorted(set(wg[xcc == x0].tolist()))
        st = [(int(w), int((rt0[wg == w].min() - t0) * 10), int((rt0[wg == w].max() - t0) * 10)) for w in ws]
        print(f"XCC {int(x0)}: (workgroup, first wave start ns, last wave start ns):", st)
        cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1   # HW_ID: [11:8] CU_ID, [12] SH_ID, [15:13] SE_ID (gfx9)
        print("waves per (se, cu) on that XCC:", sorted(set((int(a), int(b)) for a, b in zip(se[xcc == x0], cu[xcc == x0]))).__len__(), "distinct CUs")
a = np.array(acc, dtype=np.float64)   # [ticks][8][4]
print("per XCC: first wave start / last wave start / last wave end (ns after the launch's first wave), waves")
for x in range(8):
    print(f"  XCC {x}: {a[:, x, 0].mean():7.0f} {a[:, x, 1].mean():7.0f} {a[:, x, 2].mean():8.0f}  {int(a[0, x, 3])}")
