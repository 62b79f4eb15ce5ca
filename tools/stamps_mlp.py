"""Diagnostic: per-phase cycle stamps of cygym_actor_mlp_decode (build with -DCG_STAMPS; wave 0 of every workgroup)."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cygym_amd import abi, build as B
so = os.path.join(ROOT, "cygym_amd", "libcygym_hip_stamps.so")
if not (os.environ.get("CYGYM_STAMP_NOBUILD") and os.path.exists(so)):
    B.build_to(so, None, flags=["-DCG_STAMPS"], dev_mt=256)
from cygym_amd import _lib
_lib.SO = so
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
M, N, H = 256, 4096, int(os.environ.get("H", "64"))
topo, init, ck = make_topology(M, 1, seed=0, max_extra=0)
cfg = abi.EnvConfig(seed=0, lambda_events=0.0, **ck)
env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M)
env.lib.cygym_set_debug.argtypes = [C.c_void_p, C.c_void_p]
dbg = torch.zeros((N // 16, 16), dtype=torch.int64, device="cuda:0")
env.lib.cygym_set_debug(env._h, C.c_void_p(dbg.data_ptr()))
X = cfg.max_exploits
n_types, n_apps = 11, 4
n_out = n_types + M + X + n_apps
g = torch.Generator().manual_seed(0)
tm = torch.arange(n_types, dtype=torch.int32).cuda()
obs = torch.randn((N, 6 * M), generator=g).cuda()
W1 = (torch.randn((H, 6 * M), generator=g) * 0.02).cuda(); b1 = torch.zeros(H).cuda()
W = (torch.randn((n_out, H), generator=g) * 0.1).cuda(); b = (torch.randn((n_out,), generator=g) * 0.1 - 0.3).cuda()
hid = [(env.pack_linear(W1), b1, H)]
hd = (env.pack_linear(W, 64), b)
role = os.environ.get("ROLE")   # "defender": build the view on chip (obs_role)
for _ in range(5):
    env.actor_mlp_decode(None, None if role else obs, hid, hd, n_types, X, n_apps, tm, epsilon=1.0, obs_role=role)
torch.cuda.synchronize()
d = dbg.cpu().numpy().astype(np.int64)
names = ["requests", "stage0 wait+store", "stage0 mfma", "stage1 wait+store", "stage1 mfma", "stage2 wait+store", "stage2 mfma", "partials+sync",
         "finish hidden", "head mfma + sync", "decode"]
seg = np.diff(d[:, :12], axis=1)
if role:   # (stamps 0, 2, 7 .. 11 only)
    print('  end of layer-0 mfma of waves 3, 7, 11, 15 relative to wave 0 (mean):', [int((d[:, 12 + j] - d[:, 7]).mean()) for j in range(4)])
    d = d[:, [0, 1, 2, 7, 8, 9, 10, 11]]
    names = ["prologue (kernarg, pointers)", "view fill + sync", "layer-0 mfma (wave 0)", "partials + sync (waits for the slowest wave)", "finish hidden", "head mfma + sync", "decode"]
    seg = np.diff(d, axis=1)
    for i, n in enumerate(names):
        print(f"  {n:46s} {seg[:, i].mean():8.0f} {np.median(seg[:, i]):8.0f} {seg[:, i].max():8d}")
    print("  total per workgroup   ", int((d[:, -1] - d[:, 0]).mean()))
    sys.exit(0)
print("cycles (s_memtime) per phase, mean / p50 / max over workgroups; kernel span:", int(d[:, 11].max() - d[:, 0].min()))
for i, n in enumerate(names):
    print(f"  {n:22s} {seg[:, i].mean():8.0f} {np.median(seg[:, i]):8.0f} {seg[:, i].max():8d}")
print("  total per workgroup   ", int((d[:, 11] - d[:, 0]).mean()), " start spread", int(d[:, 0].max() - d[:, 0].min()))
