"""Microbenchmark: the actor head kernels back to back (HIP events), MFMA vs scalar variant."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from cygym_amd import abi
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
M, N, H = 256, 4096, 64
topo, init, ck = make_topology(M, 1, seed=0, max_extra=0)
cfg = abi.EnvConfig(seed=0, lambda_events=0.0, **ck)
env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=M)
X = cfg.max_exploits
n_types, n_apps = 11, 4
n_out = n_types + M + X + n_apps
g = torch.Generator().manual_seed(0)
hidden = torch.randn((N, H), generator=g).cuda()
W = (torch.randn((n_out, H), generator=g) * 0.1).cuda()
b = (torch.randn((n_out,), generator=g) * 0.1 - 0.3).cuda()
Wt = env.head_weights(W)
tm = torch.arange(n_types, dtype=torch.int32).cuda()
def run(name, fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / reps * 1e3:.2f} us per launch")
run("actor_head_decode (eps=1)", lambda: env.actor_head_decode(None, hidden, Wt, b, n_types, X, n_apps, tm, epsilon=1.0))
run("actor_head_decode (eps=0)", lambda: env.actor_head_decode(None, hidden, Wt, b, n_types, X, n_apps, tm))
vec = torch.addmm(b, hidden, W.t())
run("decode_actions only", lambda: env.decode_actions(None, vec, n_types, X, n_apps, tm))
run("addmm [4096x64]x[64x281]", lambda: torch.addmm(b, hidden, W.t()))
obs = torch.randn((N, 6 * M), generator=g).cuda(); W1 = torch.randn((H, 6 * M), generator=g).cuda(); b1 = torch.zeros(H).cuda()
run("addmm_activation [4096x1536]x[1536x64]", lambda: torch._addmm_activation(b1, obs, W1.t()))
env.gen_actions(0)
run("cygym_step (synthetic script tick 0, repeated)", lambda: env.step())
# (the whole actor in one launch, cygym_actor_mlp_decode: tools/exp_mlp.py)
