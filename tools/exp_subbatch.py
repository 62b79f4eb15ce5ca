"""Experiment: per-tick stepping as S sub-batches on S streams (cygym_step_range) vs one full-batch launch per tick,
launched eagerly from Python or replayed from one HIP graph holding all K ticks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cygym_amd import abi
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology

def run(N, M, blocks, K=40, W=10, subs=(1, 2, 4, 8), reps=7):
    dev = torch.device("cuda:0")
    topo, init, ck = make_topology(M, blocks, seed=0, max_extra=0)
    cfg = abi.EnvConfig(seed=0, auto_reset=1, lambda_events=0.0, **ck)
    env = BatchedCyberDefenseEnv(topo, cfg, N, init, device=dev, max_groups=1, max_devs=max(1, M // 8))
    scripts = []
    for t in range(W + K):
        act = {k: torch.empty_like(v) for k, v in env.act.items()}
        env.gen_actions(t, act); scripts.append(act)
    for t in range(W): env.step(scripts[t])
    torch.cuda.synchronize()
    keep = {k: env.state[k].clone() for k in abi.BUFFER_FIELDS}
    for S in subs:
        per = (N + S - 1) // S
        streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
        def issue():
            cur = torch.cuda.current_stream(dev)
            if S == 1:
                for t in range(W, W + K): env.step(scripts[t])
                return
            for st in streams: st.wait_stream(cur)
            for t in range(W, W + K):
                for j, st in enumerate(streams):
                    with torch.cuda.stream(st):
                        env.step_range(j * per, max(0, min(per, N - j * per)), scripts[t])
            for st in streams: cur.wait_stream(st)
        g = torch.cuda.CUDAGraph()
        cap = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(cap):
            issue(); torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=cap):
                issue()
        torch.cuda.synchronize()
        for mode in ("eager", "graph"):
            times = []
            for r in range(reps):
                for k, v in keep.items(): env.state[k].copy_(v)
                torch.cuda.synchronize()
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record()
                if mode == "eager": issue()
                else: g.replay()
                e1.record()
                torch.cuda.synchronize()
                times.append((time.perf_counter() - t0, e0.elapsed_time(e1) / 1e3))
            wall = np.median([x[0] for x in times]); ev = np.median([x[1] for x in times])
            print(f"N={N} M={M} S={S} {mode}: {N*K/wall:.3e} env-steps/s  wall/tick {wall/K*1e6:.1f} us  events/tick {ev/K*1e6:.1f} us  chk {float(env.raw.sum()):.3f}", flush=True)
    env.close()

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "target"
    cfgs = {"target": (4096, 256, 1), "cfg2": (4096, 64, 4), "cfg3": (16384, 256, 1), "cfg5": (4096, 2048, 32)}
    for w in which.split(","):
        run(*cfgs[w])
