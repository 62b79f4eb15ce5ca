"""Microbenchmark of cygym_actor_mlp_decode (HIP events, back-to-back launches)."""
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
tm = torch.arange(n_types, dtype=torch.int32).cuda()
def run(name, fn, reps=300):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / reps * 1e3:.2f} us per launch", flush=True)
obs = torch.randn((N, 6 * M), generator=g).cuda()
W1 = (torch.randn((H, 6 * M), generator=g) * 0.02).cuda(); b1 = torch.zeros(H).cuda()
W = (torch.randn((n_out, H), generator=g) * 0.1).cuda(); b = (torch.randn((n_out,), generator=g) * 0.1 - 0.3).cuda()
hid = [(env.pack_linear(W1), b1, H)]
hd = (env.pack_linear(W, 64), b)
tag = os.environ.get("CYGYM_MLP_DEBUG", "0")
run(f"[dbg={tag}] defender 1536->64->out", lambda: env.actor_mlp_decode(None, obs, hid, hd, n_types, X, n_apps, tm, epsilon=1.0))
run(f"[dbg={tag}] defender, view built on chip (obs_role)", lambda: env.actor_mlp_decode(None, None, hid, hd, n_types, X, n_apps, tm, epsilon=1.0, obs_role="defender"))
Ka = 4 * M + X
obs_a = torch.randn((N, Ka + 2), generator=g).cuda()[:, :Ka]; Wa = (torch.randn((H, Ka), generator=g) * 0.02).cuda()
n_out_a = 3 + M + X
Wha = (torch.randn((n_out_a, H), generator=g) * 0.1).cuda(); bha = torch.zeros(n_out_a).cuda()
hid_a, hd_a = [(env.pack_linear(Wa), b1, H)], (env.pack_linear(Wha, 64), bha)
run(f"[dbg={tag}] attacker 1030->64->out (padded stride)", lambda: env.actor_mlp_decode(None, obs_a, hid_a, hd_a, 3, X, 0, None))
run(f"[dbg={tag}] attacker, view built on chip (obs_role)", lambda: env.actor_mlp_decode(None, None, hid_a, hd_a, 3, X, 0, None, obs_role="attacker"))
obs_d = obs_a.contiguous()
run(f"[dbg={tag}] attacker 1030->64->out (dense rows)", lambda: env.actor_mlp_decode(None, obs_d, hid_a, hd_a, 3, X, 0, None))
if tag == "0":
    W1b = (torch.randn((256, 6 * M), generator=g) * 0.02).cuda(); W2b = (torch.randn((256, 256), generator=g) * 0.05).cuda(); b256 = torch.zeros(256).cuda()
    Whb = (torch.randn((n_out, 256), generator=g) * 0.05).cuda()
    hid_b = [(env.pack_linear(W1b), b256, 256), (env.pack_linear(W2b), b256, 256)]
    hd_b = (env.pack_linear(Whb, 64), b)
    run("reference actor 1536->256->256->out", lambda: env.actor_mlp_decode(None, obs, hid_b, hd_b, n_types, X, n_apps, tm, tanh=True))
    run("torch: reference actor body (2 GEMMs)", lambda: torch._addmm_activation(b256, torch._addmm_activation(b256, obs, W1b.t()), W2b.t()))
    run("torch: addmm_activation [4096x1536]x[1536x64]", lambda: torch._addmm_activation(b1, obs, W1.t()))
