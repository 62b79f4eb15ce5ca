#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched CyGym tick on MI355X.

One "step" = one tick of every env of the batch, on synthetic input: the fixed-topology
generator (cygym_amd/topology.py) and the alternating defender/attacker action script
(SURVEY.md section 8d), pre-generated on device so that all inputs are resident in HBM
when the timed region starts.  The script is open loop by construction, so the K timed
steps are issued the way the library runs an open-loop rollout: cygym_rollout, K ticks in
one launch, state on chip between ticks, every tick's observation / reward / done written
to HBM (`value`, `roofline`).  The same K steps issued as K cygym_step launches -- what a
closed-loop policy would drive -- are timed too and reported under `per_tick_stepping`
(`--headline per_tick` swaps the two).

Contract (see the task description): `python bench.py --gpus N --steps K --warmup W`;
for N > 1 the driver launches one rank per GPU with torch.distributed.run.  Envs are
independent, so ranks shard the batch by env id (weak scaling: per-GPU envs fixed) and
there is NO collective on the step path; the only collectives are the barrier and the
MAX-over-ranks of the timing.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (envs per GPU, devices, blocks, description)
    "target": (4096, 256, 1, "north-star target: 4096 envs x 256 devices, Volt-Typhoon roles"),
    "cfg2": (4096, 64, 4, "BASELINE configs[1]: 4096 envs x 64 devices / 4 subnets"),
    "cfg3": (16384, 256, 1, "BASELINE configs[2]: 16384 envs x 256 devices"),
    "cfg5": (4096, 2048, 32, "BASELINE configs[4]: 4096 envs x 2048 devices / 32 subnets"),
}

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PMC_SUMMARY = "r01_v11_target_pmc_summary.json"


def algorithmic_bytes(M: int, E: int) -> float:
    """B(M,E) = M*(S_r+S_w) + M*O_w + 2*ceil(E/8) + A, S_r=S_w=8, O_w=24, A=M/8+16."""
    return M * 16.0 + M * 24.0 + 2.0 * ((E + 7) // 8) + (M / 8.0 + 16.0)


def layout_bytes(M: int, E: int) -> float:
    """What this build's layout must move per env-step at minimum: 3 live planes r+w (flags, busy, wl),
    comp_by read, obs write, env scalars r+w, action header + mean list; blocked bitmask only on
    attacker / edge ticks (counted at 1/2)."""
    return M * 3 * 2 + M + M * 24.0 + 2 * (16 * 4 + 3 * 8) + 48 + M / 8.0 + 0.5 * 4 * ((E + 31) // 32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="target", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="override envs per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--max-extra", type=int, default=0,
                    help="capacity of the per-env list of edges evolve_network may add.  0 (default): no list -- this is the "
                         "fixed-topology run of SURVEY.md 8d (lambda_events = 0 can never add an edge; bench.py checks that "
                         "no env wanted one), lean kernels; -1: the topology generator's default (room for two attacker "
                         "stars), i.e. the kernels that also follow added edges (about 1 %% slower at 4096 x 256)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--fused", type=int, default=-1,
                    help="ticks per cygym_rollout launch (-1 = all K steps in one launch, 0 = skip the rollout leg)")
    ap.add_argument("--headline", default="rollout", choices=["rollout", "per_tick"],
                    help="which way of issuing the K steps fills value / roofline (the other one is reported beside it)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cygym_amd import abi
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal aid (one-GPU box): CYGYM_BENCH_BACKEND=gloo CYGYM_BENCH_SAME_GPU=1 runs every rank on cuda:0
    # with CPU collectives, to exercise the multi-rank code path without a second GPU.
    backend = os.environ.get("CYGYM_BENCH_BACKEND", "nccl")
    if os.environ.get("CYGYM_BENCH_SAME_GPU") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"[bench] warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    dev = torch.device(f"cuda:{local_rank}")
    torch.cuda.set_device(dev)

    n_per_gpu, M, blocks, desc = WORKLOADS[args.workload]
    if args.envs:
        n_per_gpu = args.envs
    topo, init, ck = make_topology(M, blocks, seed=args.seed, max_extra=None if args.max_extra < 0 else args.max_extra)
    # fixed-topology roofline run: lambda_events = 0; everything else at reference defaults
    cfg = abi.EnvConfig(seed=args.seed, env_id_base=rank * n_per_gpu, auto_reset=1, lambda_events=0.0, **ck)
    L = max(1, M // 8)
    env = BatchedCyberDefenseEnv(topo, cfg, n_per_gpu, init, device=dev, max_groups=1, max_devs=L)

    K, W = args.steps, args.warmup
    # pre-generate the action script for every tick: inputs resident in HBM before timing
    scripts = []
    for t in range(W + K):
        act = {k: torch.empty_like(v) for k, v in env.act.items()}
        env.gen_actions(t, act)
        scripts.append(act)
    torch.cuda.synchronize(dev)

    for t in range(W):
        env.step(scripts[t])
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    env.timer_start()
    for t in range(W, W + K):
        env.step(scripts[t])
    ev_ms = env.timer_stop()          # HIP events on the stream the kernels were launched on
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([wall, ev_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tt[0]), float(tt[1])

    total_envs = n_per_gpu * world
    launch_s = (ev_ms / 1e3) / K      # average launch duration over the timed region (incl. inter-kernel gaps)
    B = algorithmic_bytes(M, topo.E)
    ret_sum = float(env.raw.sum())
    per_tick = {"what": "K launches of cygym_step (what a closed-loop policy drives)",
                "value": total_envs * K / wall, "unit": "env-steps/s", "ms_per_step": wall / K * 1e3,
                "roofline": roofline_block(n_per_gpu * B, launch_s, "step_kernel<.., FUSED=0>", 1),
                "last_raw_reward_sum": ret_sum}
    rollout = None
    if args.fused:
        rollout = fused_leg(env, init, scripts, W, K, args.fused, world, dev, n_per_gpu, B, backend)
    head = rollout if (rollout is not None and args.headline == "rollout") else per_tick
    other = per_tick if head is rollout else rollout

    out = {
        "metric": "env-steps/sec", "value": head["value"], "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {desc}", "envs_per_gpu": n_per_gpu, "devices": M, "edges": topo.E,
                   "exploits": topo.X, "lambda_events": 0.0, "max_extra_edges": topo.max_extra,
                   "stepping": head["what"],
                   "parallelism": f"env-batch split x{world}, no step-path collective"},
        "roofline": head["roofline"],
        "check": {"last_raw_reward_sum": head["last_raw_reward_sum"],
                  "same_trajectory_both_ways": (rollout is None) or (rollout["last_raw_reward_sum"] == ret_sum)},
    }
    out["roofline"].update({"bytes_per_env_step": B, "layout_bytes_per_env_step": layout_bytes(M, topo.E)})
    if other is not None:
        out["per_tick_stepping" if other is per_tick else "fused_rollout"] = other

    # HBM traffic of each kernel from the committed PMC passes of this same command
    # (profiles/: separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per the gfx950 note)
    pmc = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    if args.workload == "target" and not args.envs and os.path.exists(pmc):
        try:
            c = json.load(open(pmc))
            for leg, key in ((per_tick, "per_tick"), (rollout, "fused")):
                if leg is None or key not in c:
                    continue
                kb = 2.0 * c[key]["FETCH_SIZE"]["mean_per_launch"] + c[key]["WRITE_SIZE"]["mean_per_launch"]
                ticks_pmc = float(c[key].get("ticks_per_launch", 1))
                r = leg["roofline"]
                r["traffic"] = kb * 1024.0 / ticks_pmc * r["ticks_per_launch"]
                r["traffic_unit"] = "bytes per launch"
                r["traffic_source"] = f"profiles/{PMC_SUMMARY} (rocprofv3 --pmc passes of this command, tools/rocprof_summary.py)"
        except Exception:
            pass
    if rank == 0:
        # also price the kernel against a device-copy bandwidth measured on this box
        bw = measured_copy_gbs(dev)
        out["roofline"]["measured_copy_peak"] = bw
        out["roofline"]["frac_of_measured_copy"] = out["roofline"]["achieved"] / bw if bw else None
    # the fixed-topology run must never have wanted an edge it could not add (CG_E_TOPO_OVF, cygym_spec.h)
    from cygym_amd import spec as S
    out["check"]["envs_that_needed_an_unavailable_edge"] = int(((env.state["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF) != 0).sum())
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(topo, init, cfg, M, L, scripts, W, args.cpu_seconds)
    if rank == 0:
        print(json.dumps(out))
    env.close()
    if world > 1:
        dist.destroy_process_group()


def fused_leg(env, init, scripts, W, K, T, world, dev, n_per_gpu, B, backend="nccl"):
    """cygym_rollout: the same K ticks on the same script, T ticks per launch (open-loop), every tick's
    observation still written to HBM.  Restarts from the initial state, so its last reward sum must equal
    the per-tick leg's (same trajectory)."""
    import torch
    import torch.distributed as dist
    T = K if T < 0 else min(T, K)
    env.load_state(init)
    for t in range(W):   # the W untimed warm-up steps
        env.step(scripts[t])
    n_launch = (K + T - 1) // T
    chunks = []
    for c in range(n_launch):
        lo, hi = W + c * T, min(W + K, W + (c + 1) * T)
        act = {k: torch.stack([scripts[t][k] for t in range(lo, hi)]).contiguous() for k in scripts[0]}
        _, out = env.alloc_rollout(hi - lo)
        chunks.append((act, out))
    # warm the rollout kernel with a throw-away pass of the same launches, then put the state back at tick W
    keep = {k: env.state[k].clone() for k in abi_buffer_fields()}
    for act, out in chunks:
        env.rollout(act, out)
    for k, v in keep.items():
        env.state[k].copy_(v)
    del keep
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    env.timer_start()
    for act, out in chunks:
        env.rollout(act, out)
    ev_ms = env.timer_stop()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([wall, ev_ms], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        wall, ev_ms = float(tt[0]), float(tt[1])
    n_launch = len(chunks)
    return {"what": f"cygym_rollout: {T} ticks per launch (open loop, pre-staged action script; state on chip between "
                    "ticks; every tick's obs / reward / done written to HBM)",
            "value": n_per_gpu * world * K / wall, "unit": "env-steps/s", "ms_per_step": wall / K * 1e3,
            "roofline": roofline_block(n_per_gpu * B * K / n_launch, (ev_ms / 1e3) / n_launch, "step_kernel<.., FUSED=1>", K / n_launch),
            "last_raw_reward_sum": float(chunks[-1][1]["raw"][-1].sum())}


def abi_buffer_fields():
    from cygym_amd import abi
    return abi.BUFFER_FIELDS


def roofline_block(bytes_per_launch, launch_s, kernel, ticks_per_launch):
    achieved = bytes_per_launch / launch_s / 1e9
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "kernel": kernel, "launch_us": launch_s * 1e6, "ticks_per_launch": ticks_per_launch,
            "algorithmic_bytes_per_launch": bytes_per_launch}


def measured_copy_gbs(dev, mib=1024, reps=8):
    """Device-to-device copy bandwidth (read + write bytes / time) of a 1 GiB buffer, HIP events."""
    import torch
    n = mib << 20
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize(dev)
    ms = e0.elapsed_time(e1)
    del a, b
    return 2.0 * n * reps / (ms / 1e3) / 1e9 if ms > 0 else None


def cpu_baseline(topo, init, cfg, M, L, scripts, W, budget_s):
    """The CPU oracle (C restatement) on a bounded sample of the same workload: the first n envs per thread,
    replaying the same pre-generated script from the initial state until the time budget is spent.  Two legs,
    half the budget each: one thread, then one thread per host core of this process's CPU share (envs are
    independent, so each thread owns a contiguous env range -- the same sharding the GPU ranks use)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import driver as od
    n_all = scripts[0]["mode"].shape[0]
    per = min(1024, n_all)
    if n_all // per < 16:   # give every host core of the share a contiguous env range
        per = max(64, n_all // 16)
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    threads = max(1, min(share, 16, n_all // per))

    def leg(n_thr, seconds):
        n = per * n_thr
        ob = od.OracleBatch(topo, cfg, n)
        acts = [{k: np.ascontiguousarray(v[:n].cpu().numpy()) for k, v in act.items()} for act in scripts]
        steps, reps = 0, 0
        t0 = time.perf_counter()

        def run(j):   # ctypes drops the GIL inside cgo_step
            for a in acts:
                ob.step(a, j * per, (j + 1) * per)

        with ThreadPoolExecutor(n_thr) as pool:
            while True:
                ob.load_state(init)
                list(pool.map(run, range(n_thr)))
                steps += n * len(acts)
                reps += 1
                if time.perf_counter() - t0 > seconds:
                    break
        dt = time.perf_counter() - t0
        return steps / dt, n, reps, steps, dt

    v1, n1, r1, s1, d1 = leg(1, budget_s / 2)
    out = {"value": v1, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"oracle/cygym_oracle.c, 1 thread: first {n1} envs x {len(scripts)} ticks of the same script, "
                     f"replayed {r1}x ({s1} env-steps in {d1:.1f} s)",
           "host_cores_available": share}
    if threads > 1:
        vt, nt, rt, st, dt = leg(threads, budget_s / 2)
        out["single_thread"] = {"value": v1, "cores": 1, "sample": out["sample"]}
        out.update({"value": vt, "cores": threads,
                    "sample": f"oracle/cygym_oracle.c, {threads} threads x {per} envs each: first {nt} envs x "
                              f"{len(scripts)} ticks of the same script, replayed {rt}x ({st} env-steps in {dt:.1f} s)"})
    return out


if __name__ == "__main__":
    main()
