#!/usr/bin/env python3
"""tools/exp_ippo.py -- the batched IPPO rollout collector alone (cygym_amd/ippo_rollout.collect) at the `target` size:
    python tools/exp_ippo.py [--envs 4096] [--M 256] [--decisions 50] [--role defender]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cygym_amd import abi
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.ippo_rollout import collect, gae
from cygym_amd.topology import make_topology

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096); ap.add_argument("--M", type=int, default=256)
ap.add_argument("--decisions", type=int, default=50); ap.add_argument("--role", default="defender")
ap.add_argument("--hidden", type=int, default=64)
a = ap.parse_args()
M, N = a.M, a.envs
topo, init, ck = make_topology(M, 1, seed=0, max_extra=0)
cfg = abi.EnvConfig(seed=0, lambda_events=0.0, auto_reset=1, **ck)
X = cfg.max_exploits
K = 14 if a.role == "defender" else X + 3
F = 6 if a.role == "defender" else 4
batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=14, max_devs=M)


class PerDeviceNet(torch.nn.Module):
    """Independent per-device actors over the device's own view row + a centralised critic (the shape of the reference's
    IPPO network without the GAT: IPPO.py:100-260)."""
    def __init__(self):
        super().__init__()
        self.actor = torch.nn.Sequential(torch.nn.Linear(F, a.hidden), torch.nn.ReLU(), torch.nn.Linear(a.hidden, K))
        self.critic = torch.nn.Sequential(torch.nn.Linear(F * M, a.hidden), torch.nn.ReLU(), torch.nn.Linear(a.hidden, 1))
        self.exp = torch.nn.Linear(F * M, X)
        self.app = torch.nn.Linear(F * M, 4)

    def forward(self, state, vis):
        x = state[:, : F * M]
        return {"per_dev_type_logits": self.actor(x.reshape(-1, M, F)), "value": self.critic(x).squeeze(-1),
                "exp_logits": self.exp(x), "app_logits": self.app(x)}


net = PerDeviceNet().to("cuda:0").eval()
opp = "No Attack" if a.role == "defender" else "No Defense"
g = torch.Generator(device="cuda:0").manual_seed(0)
collect(batch, a.role, net, opp, 4, generator=g)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    ro = collect(batch, a.role, net, opp, a.decisions, generator=g)
    torch.cuda.synchronize(); dt = time.time() - t0
    ticks = 2 * a.decisions
    print(f"collect {a.role}: {a.decisions} decisions x {N} envs in {dt * 1e3:.1f} ms = {dt / ticks * 1e6:.0f} us per tick, {N * ticks / dt:.3e} env-steps/s, "
          f"{N * a.decisions / dt:.3e} decisions/s; mean groups {float(batch.act['n_groups'].float().mean()):.1f}")
with torch.no_grad():
    v_last = net(ro.last_state, ro.last_vis)["value"]
adv, ret = gae(ro.reward * 0.1, torch.cat([ro.value, v_last[None]]), ro.done)
print("gae:", tuple(adv.shape), float(adv.abs().mean()))
if os.environ.get("PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        collect(batch, a.role, net, opp, 10, generator=g)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
