#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched CyGym tick on MI355X.

One "step" = one tick of every env of the batch, on synthetic input: the fixed-topology generator
(cygym_amd/topology.py) and the alternating defender/attacker action script, pre-generated
on device so that all inputs are resident in HBM when a timed region starts.

Two ways of issuing the same K ticks are timed, from the same state, and must end with the same rewards:

  * `per_tick_stepping` -- K x cygym_step: one launch per tick (per sub-batch), i.e. what `env.step()` drives and
    what a closed-loop policy can use.  THIS is `value` / `roofline` (the reference's callers are closed-loop:
    do_agent.py:206-272, IPPO.py:503-620).  By default the batch is stepped as `--sub-batches` S contiguous
    sub-batches, each on its own HIP stream (cygym_step_range): a sub-batch's next tick starts when ITS slowest env
    is done, not the whole batch's, which is how a closed-loop driver pipelines policy evaluation and stepping.
    The plain one-launch-per-tick figure (S = 1) is measured in the same run and reported beside it
    (`single_launch`); its `launch_us` is the figure rocprofv3's kernel trace reports for the full-batch kernel.
  * `fused_rollout` -- cygym_rollout: K ticks in ONE launch (open-loop scripts only; state on chip between ticks;
    every tick's observation / reward / done still written to HBM).  `--headline rollout` swaps the two.

Every timed region times exactly K steps and is repeated `--reps` times from the same state (restored, untimed,
between repetitions); the MEDIAN repetition is reported (max over ranks first).  HIP events on the launch stream
give `roofline.launch_us`.

Contract (see the task description): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver
launches one rank per GPU with torch.distributed.run.  Envs are independent, so ranks shard the batch by env id
(weak scaling: per-GPU envs fixed) and there is NO collective on the step path; the collectives are the barrier,
the MAX-over-ranks of the timings and one end-of-run all_gather of per-env returns (checked against a
single-rank recomputation).  Rank 0 prints ONE JSON line.
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
    "cfg4": (16384, 256, 1, "BASELINE configs[3], per-GPU shard: 131072 envs x 256 devices over 8 GPUs = 16384 per GPU"),
    "cfg5": (4096, 2048, 32, "BASELINE configs[4]: 4096 envs x 2048 devices / 32 subnets"),
}
# Sub-batches of the pipelined per-tick leg.  Measured (tools/exp_subbatch.py): it pays only when the batch
# oversubscribes the chip (more envs than resident waves: cfg3 / cfg4 +16 %, cfg5 +19 %); at 4096 envs every env has
# its own resident wave and a second stream only adds launches (-7 %), so those workloads step with one launch per tick.
DEFAULT_SUB = {"target": 1, "cfg2": 1, "cfg3": 2, "cfg4": 2, "cfg5": 4}
# Extra-edge capacity per workload (--max-extra overrides): 0 = fixed topology, lean kernels; -1 = the generator's default
# capacity, full-feature kernels.  Only cfg5's topology can ask for an edge with lambda_events = 0 (see --max-extra).
DEFAULT_MAX_EXTRA = {"target": 0, "cfg2": 0, "cfg3": 0, "cfg4": 0, "cfg5": -1}

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PMC_SUMMARY = "r04_pmc_summary.json"   # profiles/: per workload and kernel class, bytes per launch from the --pmc passes
REFERENCE_PYTHON = "r04_reference_python.json"   # profiles/: the reference's own step() timed in the build container


def kernel_source_hash() -> str:
    """sha256 over the kernel sources (cygym_amd/csrc/*, include/*.h), 16 hex digits: the PMC summary under profiles/ carries the
    hash of the kernels it was collected on, and `roofline.traffic` is only filled in from it when this run's kernels are the same."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "cygym_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def algorithmic_bytes(M: int, E: int) -> float:
    """B(M,E) = M*(S_r+S_w) + M*O_w + 2*ceil(E/8) + A, S_r=S_w=8, O_w=24, A=M/8+16."""
    return M * 16.0 + M * 24.0 + 2.0 * ((E + 7) // 8) + (M / 8.0 + 16.0)


def state_term_bytes(M: int) -> float:
    """The M*(S_r+S_w) term of B: per-device state read + written every tick.  The fused rollout keeps the state on
    chip between ticks and never moves it, so its fraction is also reported without this term."""
    return M * 16.0


def layout_bytes(M: int, E: int) -> float:
    """What this build's layout must move per env-step at minimum: 3 live planes r+w (flags, busy, wl),
    comp_by read, obs write, env scalars r+w, action header + mean list; blocked bitmask only on
    attacker / edge ticks (counted at 1/2)."""
    return M * 3 * 2 + M + M * 24.0 + 2 * (16 * 4 + 3 * 8) + 48 + M / 8.0 + 0.5 * 4 * ((E + 31) // 32)


class Dist:
    """The few collectives the bench needs (none of them on the step path)."""

    def __init__(self):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # Rehearsal aid (one-GPU box): CYGYM_BENCH_SAME_GPU=1 runs every rank on cuda:0 -- together with
        # CYGYM_BENCH_BACKEND=gloo (CPU collectives): RCCL refuses two ranks on one device ("Duplicate GPU detected",
        # tried in round 3), so the nccl branch needs one GPU per rank.
        self.backend = os.environ.get("CYGYM_BENCH_BACKEND", "nccl")
        if os.environ.get("CYGYM_BENCH_SAME_GPU") == "1":
            self.local_rank = 0
        self.dev = torch.device(f"cuda:{self.local_rank}")
        torch.cuda.set_device(self.dev)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
            else:
                dist.init_process_group(self.backend, rank=self.rank, world_size=self.world)

    def identity(self):
        """What every rank saw, gathered on all ranks: rank, local rank, device index, the device's uuid / PCI bus id, the
        backend and the world size IT reports -- so that "N ranks on N distinct devices over RCCL" is checkable from the
        bench line.  Distinct devices are REQUIRED unless CYGYM_BENCH_SAME_GPU=1 (the one-GPU rehearsal)."""
        torch, dist = self.torch, self.dist
        p = torch.cuda.get_device_properties(self.dev)
        parts = [f"name={p.name}"]
        for attr in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"):   # (whichever this torch build exposes)
            v = getattr(p, attr, None)
            if v is not None:
                parts.append(f"{attr}={v}")
        dev_id = "|".join(parts)
        me = {"rank": self.rank, "local_rank": self.local_rank, "device_index": self.dev.index, "device": dev_id or "unknown",
              "backend": (dist.get_backend() if self.world > 1 else "none"), "world_size_reported": (dist.get_world_size() if self.world > 1 else 1)}
        ranks = [me]
        if self.world > 1:
            ranks = [None] * self.world
            dist.all_gather_object(ranks, me)
        distinct = len({(r["device_index"], r["device"]) for r in ranks}) == len(ranks)
        same_ok = os.environ.get("CYGYM_BENCH_SAME_GPU") == "1"
        if self.world > 1 and not distinct and not same_ok:
            raise RuntimeError(f"ranks share a device: {ranks} (set CYGYM_BENCH_SAME_GPU=1 only for the one-GPU rehearsal)")
        return {"ranks": ranks, "distinct_devices": distinct, "same_gpu_rehearsal": same_ok,
                "collectives": "barrier + MAX all_reduce of the timings + one all_gather of per-env returns; none on the step path"}

    def barrier(self):
        self.torch.cuda.synchronize(self.dev)
        if self.world > 1:
            self.dist.barrier()
            self.torch.cuda.synchronize(self.dev)

    def max_over_ranks(self, values):
        if self.world == 1:
            return list(values)
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.dev if self.backend == "nccl" else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(x) for x in t]

    def close(self):
        if self.world > 1:
            self.dist.destroy_process_group()


def timed_reps(D: Dist, env, keep, reps, issue):
    """`reps` repetitions of one timed region: restore the state (untimed), barrier + synchronize, issue, barrier +
    synchronize.  Returns (median wall seconds, median HIP-event seconds), each the max over ranks per repetition.
    The wall clock and the HIP events are taken in SEPARATE repetitions (`reps` of each): the two event records are
    stream operations of their own, and inside the wall-clocked region they cost ~0.2 us per step at K = 20."""
    torch = D.torch
    walls, evs = [], []
    for i in range(2 * reps):
        for k, v in keep.items():
            env.state[k].copy_(v)
        D.barrier()
        if i & 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()                # HIP events on the stream the kernels are launched on (torch's current stream)
            issue()
            e1.record()
            D.barrier()
            evs.append(e0.elapsed_time(e1) / 1e3)
            continue
        t0 = time.perf_counter()
        issue()
        # ONE wait for the region: torch.cuda.synchronize (+ the ranks' barrier).  Round 3 waited on the stop event first
        # (hipEventSynchronize: a sleeping wait, ~40 us to wake up) and then synchronised three more times: 50 us of host
        # latency per region, 2.5 us per step at K = 20, inside `ms_per_step` (tools/exp_sync.py).
        D.barrier()
        walls.append(time.perf_counter() - t0)
    walls = D.max_over_ranks(walls)
    evs = D.max_over_ranks(evs)
    return float(np.median(walls)), float(np.median(evs)), walls


def run_workload(D: Dist, name, n_per_gpu, K, W, reps, sub, fused_T, seed, max_extra, want_rollout=True):
    """All legs of one workload on this rank's shard.  Returns (record, env-side objects for the CPU baseline)."""
    import torch
    from cygym_amd import abi
    from cygym_amd import spec as S
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.topology import make_topology
    _, M, blocks, desc = WORKLOADS[name]
    dev = D.dev
    topo, init, ck = make_topology(M, blocks, seed=seed, max_extra=None if max_extra < 0 else max_extra)
    # fixed-topology roofline run: lambda_events = 0; everything else at reference defaults
    cfg = abi.EnvConfig(seed=seed, env_id_base=D.rank * n_per_gpu, auto_reset=1, lambda_events=0.0, **ck)
    L = max(1, M // 8)
    env = BatchedCyberDefenseEnv(topo, cfg, n_per_gpu, init, device=dev, max_groups=1, max_devs=L)
    N = n_per_gpu
    scripts = []   # the action script of every tick, generated on device: inputs resident in HBM before timing
    for t in range(W + K):
        act = {k: torch.empty_like(v) for k, v in env.act.items()}
        env.gen_actions(t, act)
        scripts.append(act)
    # clock ramp: a quarter of a second of untimed stepping first (a workload that follows an idle stretch -- the CPU
    # baseline, a process start -- otherwise runs its few hundred microseconds of timed region at idle clocks: the
    # short cfg2 legs came out up to 2x low), then back to the initial state
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        for t in range(W + K):
            env.step(scripts[t])
        torch.cuda.synchronize(dev)
    env.load_state(init)
    for t in range(W):
        env.step(scripts[t])
    torch.cuda.synchronize(dev)
    keep = {k: env.state[k].clone() for k in abi.BUFFER_FIELDS}   # the state at tick W: every timed region starts here
    B = algorithmic_bytes(M, topo.E)
    total_envs = N * D.world
    main = torch.cuda.current_stream(dev)

    def single():
        for t in range(W, W + K):
            env.step(scripts[t])

    sub = max(1, min(int(sub), N))
    streams = [torch.cuda.Stream(device=dev) for _ in range(sub)] if sub > 1 else []
    per = (N + sub - 1) // sub

    def _pipelined(cur):
        for st in streams:
            st.wait_stream(cur)
        for t in range(W, W + K):
            for j, st in enumerate(streams):
                with torch.cuda.stream(st):
                    env.step_range(j * per, max(0, min(per, N - j * per)), scripts[t])
        for st in streams:
            cur.wait_stream(st)

    def pipelined():
        graph.replay()

    graph, cap = None, None
    if sub > 1:   # the S x K launches are captured once, below (fork / join of the side streams inside the capture)
        cap = torch.cuda.Stream(device=dev)
        graph = torch.cuda.CUDAGraph()

    def leg(issue, launches_per_tick, what):
        wall, ev, walls = timed_reps(D, env, keep, reps, issue)
        r = roofline_block(N * B, ev / K, "step_kernel<WPB, M, FUSED=0, ..>", 1, wall / K)
        r["launches_per_tick"] = launches_per_tick
        return {"what": what, "value": total_envs * K / wall, "unit": "env-steps/s", "ms_per_step": wall / K * 1e3,
                "sub_batches": launches_per_tick, "reps": reps, "rep_spread": [min(walls) / K * 1e3, max(walls) / K * 1e3],
                "roofline": r, "last_raw_reward_sum": float(env.raw.sum())}

    # (plain env.step() calls: the same K launches replayed from one HIP graph measured the same -- 15.44-15.45 against 15.39-15.50 us
    # per step at K = 20 -- the host enqueues a step in less time than the GPU runs one)
    one = leg(single, 1, "K launches of cygym_step, one full-batch launch per tick")
    if sub > 1:
        with torch.cuda.stream(cap):
            _pipelined(cap)                     # warm-up outside the capture
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(graph, stream=cap):
                _pipelined(cap)
        torch.cuda.synchronize(dev)
        per_tick = leg(pipelined, sub, f"K ticks of cygym_step_range on {sub} sub-batches of {per} envs, one HIP stream each, replayed "
                       "from one HIP graph (closed-loop capable: every tick of every sub-batch is its own launch)")
        per_tick["single_launch"] = {k: one[k] for k in ("value", "ms_per_step", "roofline", "rep_spread")}
        same_sub = per_tick["last_raw_reward_sum"] == one["last_raw_reward_sum"]
    else:
        per_tick, same_sub = one, True

    rollout = None
    if want_rollout and fused_T != 0:
        T = K if fused_T < 0 else min(fused_T, K)
        chunks = []
        for c in range((K + T - 1) // T):
            lo, hi = W + c * T, min(W + K, W + (c + 1) * T)
            act = {k: torch.stack([scripts[t][k] for t in range(lo, hi)]).contiguous() for k in scripts[0]}
            _, out = env.alloc_rollout(hi - lo)
            chunks.append((act, out))

        def fused():
            for act, out in chunks:
                env.rollout(act, out, check=False)    # asynchronous form; the status word is read once, after the timed region
        wall, ev, walls = timed_reps(D, env, keep, reps, fused)
        n_launch = len(chunks)
        r = roofline_block(N * B * K / n_launch, ev / n_launch, "step_kernel<WPB, M, FUSED=1, ..>", K / n_launch, wall / n_launch)
        r["frac_without_state_term"] = (N * (B - state_term_bytes(M)) * K / n_launch) / (max(ev, wall) / n_launch) / 1e9 / HBM_PEAK_GBS
        rollout = {"what": f"cygym_rollout: {T} ticks per launch (open loop, pre-staged action script; state on chip between "
                           "ticks; every tick's obs / reward / done written to HBM)",
                   "value": total_envs * K / wall, "unit": "env-steps/s", "ms_per_step": wall / K * 1e3, "reps": reps,
                   "rep_spread": [min(walls) / K * 1e3, max(walls) / K * 1e3], "roofline": r,
                   "last_raw_reward_sum": float(chunks[-1][1]["raw"][-1].sum())}
        del chunks

    x_used = (env.state["ienv"][:, S.I_FLAGS].to(torch.int64) & 0xFFFFFFFF) >> 16   # live entries of the extra-edge lists
    rec = {"workload": f"{name}: {desc}", "envs_per_gpu": N, "devices": M, "edges": topo.E, "exploits": topo.X,
           "max_extra_edges": topo.max_extra, "envs_with_added_edges": int((x_used > 0).sum()),
           "bytes_per_env_step": B, "layout_bytes_per_env_step": layout_bytes(M, topo.E),
           "per_tick_stepping": per_tick, "fused_rollout": rollout,
           "check": {"last_raw_reward_sum": per_tick["last_raw_reward_sum"],
                     "same_trajectory_all_ways": same_sub and (rollout is None or rollout["last_raw_reward_sum"] == per_tick["last_raw_reward_sum"]),
                     # the fixed-topology run must never have wanted an edge it could not add (CG_E_TOPO_OVF)
                     "envs_that_needed_an_unavailable_edge": int(((env.state["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF) != 0).sum())}}
    attach_traffic(rec, name, N)
    return rec, (env, topo, init, cfg, M, L, scripts)


def attach_traffic(rec, name, n_envs):
    """HBM traffic of each kernel from the committed PMC passes of this same workload (profiles/: separate --pmc
    FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), scaled to this
    run's launch shape; `frac_traffic` = those bytes / launch time / peak."""
    pmc = os.path.join(ROOT, "profiles", PMC_SUMMARY)
    if not os.path.exists(pmc):
        for leg in (rec["per_tick_stepping"], rec["fused_rollout"]):
            if leg is not None:
                leg["roofline"]["traffic_source"] = f"none: profiles/{PMC_SUMMARY} is missing (tools/profile_all.sh collects it)"
        return
    try:
        summary = json.load(open(pmc))
        have, want = summary.get("kernel_source_hash"), kernel_source_hash()
        if have != want:   # counters of OTHER kernels say nothing about this run: leave traffic / frac_traffic null and say why
            for leg in (rec["per_tick_stepping"], rec["fused_rollout"]):
                if leg is not None:
                    leg["roofline"]["traffic_source"] = (f"none: profiles/{PMC_SUMMARY} was collected on kernels {have}, this run's are {want} "
                                                         "(re-run tools/profile_all.sh)")
            return
        c = summary.get(name)
        # (the summary counts WAVES per launch: the env count rounded up to whole workgroups of up to 16 waves)
        if not c or not (n_envs <= c.get("envs_per_launch", -1) < n_envs + 16):
            return
        for leg, key in ((rec["per_tick_stepping"], "per_tick"), (rec["fused_rollout"], "fused")):
            if leg is None or key not in c:
                continue
            per_tick_bytes = (2.0 * c[key]["FETCH_SIZE_KiB_per_launch"] + c[key]["WRITE_SIZE_KiB_per_launch"]) * 1024.0 \
                / float(c[key]["ticks_per_launch"])
            for r in [leg["roofline"]] + ([leg["single_launch"]["roofline"]] if "single_launch" in leg else []):
                r["traffic"] = per_tick_bytes * r["ticks_per_launch"]
                r["traffic_unit"] = "bytes per tick of the whole batch" if key == "per_tick" else "bytes per launch"
                t_us = max(r["launch_us"], r.get("launch_us_wall") or 0.0)   # the slower clock, like the headline fraction
                r["frac_traffic"] = r["traffic"] / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBS
                r["traffic_source"] = f"profiles/{PMC_SUMMARY} (rocprofv3 --pmc passes of this same command on kernels {want}, tools/rocprof_summary.py)"
    except Exception as e:   # a malformed summary must not break the bench line
        print(f"[bench] warning: could not read {pmc}: {e}", file=sys.stderr)


def roofline_block(bytes_per_launch, launch_s, kernel, ticks_per_launch, wall_s=None):
    """`launch_s`: average time per launch from HIP events around the timed region (on the launch stream); `wall_s`: the same
    region on the host clock (what `ms_per_step` is made of: it also holds the final synchronisation).  The HEADLINE fraction is
    the lower of the two; both are given."""
    ev_frac = bytes_per_launch / launch_s / 1e9 / HBM_PEAK_GBS
    wall_frac = bytes_per_launch / wall_s / 1e9 / HBM_PEAK_GBS if wall_s else None
    frac = min(ev_frac, wall_frac) if wall_frac is not None else ev_frac
    return {"bound": "hbm", "achieved": frac * HBM_PEAK_GBS, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac,
            "frac_hip_events": ev_frac, "frac_on_ms_per_step": wall_frac,
            "traffic": None, "frac_traffic": None, "kernel": kernel, "launch_us": launch_s * 1e6,
            "launch_us_wall": wall_s * 1e6 if wall_s else None,
            "ticks_per_launch": ticks_per_launch, "algorithmic_bytes_per_launch": bytes_per_launch}


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


def gathered_returns_check(D: Dist, env, scripts, topo, init, cfg, L, n_check=64):
    """The one optional collective: all_gather of the per-env returns into global env order
    (cygym_amd/sharding.gather_by_env; RCCL with the nccl backend).  Rank 0 recomputes the first `n_check` envs as
    a single-rank batch (same global env ids, same script) and compares."""
    import dataclasses
    import torch
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.sharding import gather_by_env
    local = env.raw.clone()
    if D.backend != "nccl" and D.world > 1:
        local = local.cpu()
    allret = gather_by_env(local, env.N * D.world)
    ok = None
    if D.rank == 0:
        n = min(n_check, env.N)
        ref = BatchedCyberDefenseEnv(topo, dataclasses.replace(cfg, env_id_base=0), n, init, device=D.dev, max_groups=1, max_devs=L)
        for act in scripts:
            ref.step({k: v[:n].contiguous() for k, v in act.items()})
        torch.cuda.synchronize(D.dev)
        ok = bool(torch.equal(ref.raw.cpu(), allret[:n].cpu())) and int(allret.shape[0]) == env.N * D.world
        ref.close()
    return ok


def reference_python_block(M):
    """The reference's own Python step(), as timed in the BUILD CONTAINER (oracle/harness/time_reference.py: the reference
    cannot travel to the GPU box), from profiles/: machine and core count named there.  A baseline, not the target."""
    path = os.path.join(ROOT, "profiles", REFERENCE_PYTHON)
    try:
        d = json.load(open(path))
        row = d["by_devices"].get(str(M))
        if row is None:
            return None
        return {"value": row["steps_per_s_8_processes"], "unit": "env-steps/s", "cores": 8, "kind": "reference",
                "single_process": row["steps_per_s_1_process"], "machine": d["machine"], "measured": d["measured"],
                "sample": f"unmodified reference step() at {M} devices, alternating random defender / attacker actions; 8 processes "
                          "x 6 s each and one process x 6 s (oracle/harness/time_reference.py)",
                "note": "NOT the GPU box's host: the reference stays in the build container; " + d["what"],
                "source": f"profiles/{REFERENCE_PYTHON}"}
    except Exception as e:
        print(f"[bench] warning: could not read {path}: {e}", file=sys.stderr)
        return None


def cpu_baseline(topo, init, cfg, M, L, scripts, W, budget_s, max_threads=16):
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
    # `max_threads` (--cpu-threads, default 16): the GPU box grants a one-GPU job a 16-CPU share (its process guard kills jobs that
    # start more workers than that), whatever sched_getaffinity reports -- so "all cores" here means all cores of the share
    threads = max(1, min(share, max_threads, n_all // per))

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
           "host_cores_visible": share, "cores_cap": max_threads,
           "cores_cap_reason": "the GPU box's CPU share for a one-GPU job (16); sched_getaffinity still lists every core of the host"}
    ref = reference_python_block(M)
    if ref is not None:
        out["reference_python"] = ref
    if threads > 1:
        vt, nt, rt, st, dt = leg(threads, budget_s / 2)
        out["single_thread"] = {"value": v1, "cores": 1, "sample": out["sample"]}
        out.update({"value": vt, "cores": threads,
                    "sample": f"oracle/cygym_oracle.c, {threads} threads x {per} envs each: first {nt} envs x "
                              f"{len(scripts)} ticks of the same script, replayed {rt}x ({st} env-steps in {dt:.1f} s)"})
    return out


def closed_loop_block(D: Dist, per_tick_value, seed, n_per_gpu, M=256, blocks=1, hidden=64):
    """The closed-loop consumer timed end to end (cygym_amd/rollout_grid.simulate_grid at the `target` size): per tick ONE
    actor launch for the acting role (cygym_actor_mlp_decode: the whole Linear-ReLU-Linear network on the matrix cores, its
    observation built on chip from the envs' flag planes, + decode_action + the scatter into the action tensors = the
    reference's actor forward and do_agent.decode_action for the batch; a population of same-shaped actors shares the launch)
    and one cygym_step launch that also adds the episode returns -- the shape of do_agent.py:206-272 / :2035-2073 with the
    actors evaluated for all cells at once.  `one_launch` times the same loop with the tick and the NEXT role's actor as one
    kernel (cygym_step_actor: simulate_grid's default in the eager loop).  Grid = |D| x |A| x n_mc cells = one env each.  `torch_body` times the same loop
    with the actor's first layer as a torch GEMM on the role view the tick writes and only the last layer in the decode
    launch (cygym_actor_head_decode): what the fused actor launch replaces.

    Workload notes: fixed topology like the rest of the bench (lambda_events = 0, no ownership reshuffle at the start:
    with max_extra_edges = 0 a reshuffled attacker star cannot be re-linked); the actors are randomly initialised, so
    their action TYPES are drawn epsilon-greedily with epsilon = 1 (decode_action's own exploration, do_agent.py:972-973:
    uniform over the role's types, like the synthetic script) and their device heads are calibrated to list about M / 16
    devices per action (the script's mean); device lists, exploit and app indices are decoded from the actor outputs.

    Reported: the HIP-graph loop (ticks 6.. replayed from one captured defender+attacker pair), the eager loop, and the
    eager per-tick split observe / policy+scatter / step from HIP events (eager: includes host launch gaps)."""
    import torch
    from cygym_amd import abi
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.policies import ActorPolicy, calibrate_device_head, mlp_actor
    from cygym_amd.rollout_grid import simulate_grid
    from cygym_amd.topology import make_topology
    dev = D.dev
    topo, init, ck = make_topology(M, blocks, seed=seed, max_extra=0)
    cfg = abi.EnvConfig(seed=seed, env_id_base=D.rank * n_per_gpu, lambda_events=0.0, **ck)
    X = cfg.max_exploits
    def_types = [1, 4, 5, 6, 7, 8, 9, 11, 12, 13, 2]      # the action types of the bench's synthetic script (no training)
    att_types = [1, 2, 3]

    def grid(nD, nA, batch):
        Dp = [ActorPolicy(mlp_actor(6 * M, len(def_types) + M + X + 4, (hidden,), seed=100 + i, device=dev), len(def_types), X, 4,
                          type_map=def_types, epsilon=1.0) for i in range(nD)]
        Ap = [ActorPolicy(mlp_actor(4 * M + X, len(att_types) + M + X, (hidden,), seed=200 + j, device=dev), len(att_types), X, 0,
                          type_map=att_types, epsilon=1.0) for j in range(nA)]
        for p, role in [(p, 1) for p in Dp] + [(p, 2) for p in Ap]:
            calibrate_device_head(p, batch.observe(role), M, 1.0 / 16.0)
        return Dp, Ap

    out = {"what": "simulate_grid, closed loop: cygym_actor_mlp_decode (whole actor on the matrix cores, role view built on chip from the "
                   "flag planes, + decode + scatter) -> cygym_step (+ episode returns) per tick: 2 launches, replayed from a HIP graph; "
                   "`one_launch` times the same loop with the tick and the next role's actor as ONE kernel (cygym_step_actor); all tensors "
                   "on the device; no ownership reshuffle",
           "envs": n_per_gpu, "devices": M,
           "policy": f"per role: Linear(obs, {hidden})-ReLU-Linear({hidden}, action vector) fp32, random weights, epsilon-greedy types "
                     f"(epsilon = 1), device head calibrated to ~M/16 devices; obs = 6M (defender) / 4M + {X} (attacker) floats",
           "unit": "env-steps/s"}
    for name, (nD, nA) in (("grid_1x1", (1, 1)), ("grid_2x2", (2, 2))):
        n_mc = n_per_gpu // (nD * nA)
        N = nD * nA * n_mc
        batch = BatchedCyberDefenseEnv(topo, cfg, N, init, device=dev, max_groups=1, max_devs=M)
        Dp, Ap = grid(nD, nA, batch)
        simulate_grid(batch, Dp, Ap, n_mc, 16, randomize=False, graph=True, local_only=True)       # warm-up: allocator, GEMM heuristics, clocks
        rec = {"cells": f"{nD} x {nA} x {n_mc}"}
        runs = {}
        for mode, graph, T in (("graph", True, 106), ("graph_long", True, 306), ("eager", False, 106)):
            best = None
            for _ in range(3):
                tm = {}
                D.barrier()
                simulate_grid(batch, Dp, Ap, n_mc, T, randomize=False, graph=graph, timers=tm, local_only=True)
                t = D.max_over_ranks([tm["loop_s"]])[0]
                best = t if best is None else min(best, t)
            runs[mode] = (best, T)
        tg, Tg = runs["graph"]
        tl, Tl = runs["graph_long"]
        te, Te = runs["eager"]
        steady = (tl - tg) / (Tl - Tg)
        rec["graph"] = {"value": N * D.world * Tg / tg, "ms_per_tick": tg / Tg * 1e3, "ticks": Tg,
                        "steady_state": {"value": N * D.world / steady, "ms_per_tick": steady * 1e3,
                                         "what": f"(loop time at {Tl} ticks - loop time at {Tg} ticks) / {Tl - Tg}: replays only"}}
        rec["eager"] = {"value": N * D.world * Te / te, "ms_per_tick": te / Te * 1e3, "ticks": Te}
        tm = {"split": True}
        simulate_grid(batch, Dp, Ap, n_mc, 60, randomize=False, graph=False, timers=tm, local_only=True)
        rec["eager"]["split_us_per_tick"] = {k: tm[k] / 60 * 1e6 for k in ("observe", "policy+scatter", "step")}
        rec["mean_device_list"] = float(batch.act["dev_cnt"].float().mean())
        rec["launches_per_tick"] = ("1 cygym_actor_mlp_decode, 1 cygym_step" if nD * nA == 1 else
                                    "the acting role's actors as ONE population in env order (n_groups): 1 cygym_actor_mlp_decode, "
                                    "1 cygym_step -- independent of the number of strategies")
        one = {}
        for T in (106, 306):      # the same loop with the tick and the next role's actor as ONE launch
            best = None
            for _ in range(3):
                tm = {}
                D.barrier()
                simulate_grid(batch, Dp, Ap, n_mc, T, randomize=False, graph=True, timers=tm, merge_launches=True, local_only=True)
                t = D.max_over_ranks([tm["loop_s"]])[0]
                best = t if best is None else min(best, t)
            one[T] = best
        st1 = (one[306] - one[106]) / 200
        rec["one_launch"] = {"what": "cygym_step_actor per tick (tick + next role's actor in one kernel), replayed from a HIP graph",
                             "steady_state": {"value": N * D.world / st1, "ms_per_tick": st1 * 1e3}}
        if nD * nA == 1:      # the loop this replaces: first layer as a torch GEMM on the role view, last layer in the decode launch
            for p in Dp + Ap:
                p.fuse_mlp = False
            simulate_grid(batch, Dp, Ap, n_mc, 16, randomize=False, graph=True, local_only=True)
            tb = {}
            for T in (106, 306):
                best = None
                for _ in range(3):
                    tm = {}
                    D.barrier()
                    simulate_grid(batch, Dp, Ap, n_mc, T, randomize=False, graph=True, timers=tm, local_only=True)
                    t = D.max_over_ranks([tm["loop_s"]])[0]
                    best = t if best is None else min(best, t)
                tb[T] = best
            st_tb = (tb[306] - tb[106]) / 200
            rec["torch_body"] = {"what": "actor body as a torch GEMM (+ReLU epilogue) on the role view written by the tick, last layer + decode in "
                                         "cygym_actor_head_decode: 3 launches per tick",
                                 "steady_state": {"value": N * D.world / st_tb, "ms_per_tick": st_tb * 1e3}}
            for p in Dp + Ap:
                p.fuse_mlp = True
        rec["frac_of_per_tick_stepping"] = rec["graph"]["steady_state"]["value"] / per_tick_value if per_tick_value else None
        rec["check_unpinned_or_truncated"] = bool(batch.take_status() & (0x200 | abi.DECODE_TRUNCATED))
        out[name] = rec
        batch.close()
        del batch, Dp, Ap
    out["value"] = out["grid_1x1"]["graph"]["steady_state"]["value"]
    out["frac_of_per_tick_stepping"] = out["grid_1x1"]["frac_of_per_tick_stepping"]
    out["frac_note"] = ("value / per_tick_stepping of this run's synthetic script (one launch per tick).  What the loop adds to a tick at this "
                        "size: the actor launch, ~19-21 us for Linear(obs, 64)-ReLU-Linear(64, action vector) on 4096 rows (fp32 matrix "
                        "instructions at their issue limit ~6 us + their weight fragments through the vector-memory path ~3 us, not overlapped; "
                        "view build, layer hand-offs, last layer and the one-wave-per-row decode ~9 us; DESIGN.md section 8), and ~2.5 us of "
                        "graph-node gap per kernel")
    return out


def ippo_collect_block(D: Dist, seed, n_per_gpu, M=256, blocks=1, hidden=64, decisions=40):
    """The batched IPPO / MAPPO rollout collector timed end to end (cygym_amd/ippo_rollout.collect: the loop of IPPO.py:503-624 for
    the batch): per decision one forward of a per-device MLP actor + centralised critic (torch), ONE launch that samples a
    Categorical per visible device / exploit / app, sums the log-probabilities and groups the devices by action type
    (cygym_sample_group_actions), and two ticks (the learner's grouped step, the baseline opponent's)."""
    import time
    import torch
    from cygym_amd import abi
    from cygym_amd.batched_env import BatchedCyberDefenseEnv
    from cygym_amd.ippo_rollout import collect
    from cygym_amd.topology import make_topology
    dev = D.dev
    topo, init, ck = make_topology(M, blocks, seed=seed, max_extra=0)
    cfg = abi.EnvConfig(seed=seed, env_id_base=D.rank * n_per_gpu, lambda_events=0.0, auto_reset=1, **ck)
    X, K, F = cfg.max_exploits, 14, 6
    batch = BatchedCyberDefenseEnv(topo, cfg, n_per_gpu, init, device=dev, max_groups=14, max_devs=M)
    torch.manual_seed(seed)
    actor = torch.nn.Sequential(torch.nn.Linear(F, hidden), torch.nn.ReLU(), torch.nn.Linear(hidden, K)).to(dev).eval()
    critic = torch.nn.Sequential(torch.nn.Linear(F * M, hidden), torch.nn.ReLU(), torch.nn.Linear(hidden, 1)).to(dev).eval()
    heads = torch.nn.Linear(F * M, X + 4).to(dev).eval()

    def net(state, vis):
        x = state[:, : F * M]
        h = heads(x)
        return {"per_dev_type_logits": actor(x.reshape(-1, M, F)), "value": critic(x).squeeze(-1), "exp_logits": h[:, :X].contiguous(),
                "app_logits": h[:, X:].contiguous()}

    collect(batch, "defender", net, "No Attack", 4)
    best = None
    for _ in range(3):
        D.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        collect(batch, "defender", net, "No Attack", decisions)
        torch.cuda.synchronize(dev)
        t = D.max_over_ranks([time.perf_counter() - t0])[0]
        best = t if best is None else min(best, t)
    batch.close()
    return {"what": "ippo_rollout.collect, defender learner vs the 'No Attack' baseline: per decision 1 net forward (torch) + 1 "
                    "cygym_sample_group_actions + 2 cygym_step; Step records kept as [T, N, ...] device tensors",
            "envs": n_per_gpu, "devices": M, "decisions": decisions, "ticks": 2 * decisions,
            "value": n_per_gpu * D.world * 2 * decisions / best, "unit": "env-steps/s",
            "decisions_per_s": n_per_gpu * D.world * decisions / best, "ms_per_decision": best / decisions * 1e3,
            "policy": f"per-device actor Linear(6, {hidden})-ReLU-Linear({hidden}, 14) + critic / exploit / app heads on the 6M view, fp32, random weights"}


def brief(rec):
    """The sub-record of a secondary workload in the `configs` block."""
    out = {"workload": rec["workload"], "envs_per_gpu": rec["envs_per_gpu"], "devices": rec["devices"],
           "max_extra_edges": rec.get("max_extra_edges"), "envs_with_added_edges": rec.get("envs_with_added_edges"),
           "bytes_per_env_step": rec["bytes_per_env_step"], "check": rec["check"]}
    for key in ("per_tick_stepping", "fused_rollout"):
        leg = rec[key]
        if leg is None:
            continue
        r = leg["roofline"]
        out[key] = {"value": leg["value"], "ms_per_step": leg["ms_per_step"], "frac": r["frac"], "frac_traffic": r.get("frac_traffic"),
                    "traffic": r.get("traffic"), "launch_us": r["launch_us"], "ticks_per_launch": r["ticks_per_launch"]}
        if key == "per_tick_stepping":
            out[key]["sub_batches"] = leg["sub_batches"]
            if "single_launch" in leg:
                out[key]["single_launch"] = {"value": leg["single_launch"]["value"], "frac": leg["single_launch"]["roofline"]["frac"],
                                             "launch_us": leg["single_launch"]["roofline"]["launch_us"]}
        else:
            out[key]["frac_without_state_term"] = r.get("frac_without_state_term")
    return out


def spin_wait():
    """Host wait policy: spin instead of sleeping in hipDeviceSynchronize (hipDeviceScheduleSpin; must be set before the
    runtime creates its context, i.e. before torch touches the GPU).  A sleeping wait wakes up tens of microseconds after a
    20-step region of ~340 us has finished."""
    if os.environ.get("CYGYM_BENCH_NO_SPIN") == "1":
        return "default"
    try:
        import ctypes
        rc = ctypes.CDLL("libamdhip64.so").hipSetDeviceFlags(1)   # hipDeviceScheduleSpin
        return "spin" if rc == 0 else f"default (hipSetDeviceFlags -> {rc})"
    except OSError as e:
        return f"default ({e})"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="target", choices=sorted(WORKLOADS))
    ap.add_argument("--envs", type=int, default=0, help="override envs per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--reps", type=int, default=11, help="repetitions of every timed K-step region (median reported)")
    ap.add_argument("--sub-batches", type=int, default=0,
                    help="per-tick stepping: sub-batches / HIP streams the batch is pipelined over (0 = the workload's default, "
                         "1 = one full-batch launch per tick only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the closed-loop grid consumer leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the short runs of the other single-GPU BASELINE configs")
    ap.add_argument("--max-extra", type=int, default=None,
                    help="capacity of the per-env list of edges evolve_network may add.  0: no list -- the fixed-topology run "
                         "of SURVEY.md 8d (with lambda_events = 0 only the attacker-star check can want an edge; bench.py "
                         "reports how many envs did), lean kernels; -1: the topology generator's default (room for two "
                         "attacker stars), i.e. the full-feature kernels.  Default: 0, except for cfg5 (2048 devices), "
                         "whose sparse topology has ONE star hub: once a defender takes it out the next evolve re-links "
                         "the star with added edges, so that workload runs with the list (-1)")
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--cpu-threads", type=int, default=16, help="threads of the CPU baseline's all-cores leg (the GPU box's CPU share for one GPU is 16)")
    ap.add_argument("--fused", type=int, default=-1,
                    help="ticks per cygym_rollout launch (-1 = all K steps in one launch, 0 = skip the rollout leg)")
    ap.add_argument("--headline", default="per_tick", choices=["per_tick", "rollout"],
                    help="which way of issuing the K steps fills value / roofline (the other one is reported beside it)")
    args = ap.parse_args()

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world_env:
        # one rank per GPU: a run whose rank count differs from --gpus would report the wrong job size under the right label
        print(f"[bench] error: --gpus {args.gpus} but WORLD_SIZE={world_env}: launch with `python -m torch.distributed.run --nnodes=1 "
              f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 bench.py --gpus {args.gpus} ...`", file=sys.stderr)
        sys.exit(2)
    wait_policy = spin_wait()
    D = Dist()
    K, W = args.steps, args.warmup
    name = args.workload
    n_per_gpu = args.envs or WORKLOADS[name][0]
    sub = args.sub_batches or DEFAULT_SUB[name]
    max_extra_of = lambda w: DEFAULT_MAX_EXTRA[w] if args.max_extra is None else args.max_extra
    rec, (env, topo, init, cfg, M, L, scripts) = run_workload(D, name, n_per_gpu, K, W, args.reps, sub, args.fused, args.seed,
                                                               max_extra_of(name))
    per_tick, rollout = rec["per_tick_stepping"], rec["fused_rollout"]
    head = rollout if (rollout is not None and args.headline == "rollout") else per_tick
    other_key, other = ("per_tick_stepping", per_tick) if head is rollout else ("fused_rollout", rollout)

    out = {
        "metric": "env-steps/sec", "value": head["value"], "unit": "env-steps/s", "n_gpus": D.world, "steps": K, "warmup": W,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": rec["workload"], "envs_per_gpu": n_per_gpu, "devices": M, "edges": topo.E,
                   "exploits": topo.X, "lambda_events": 0.0, "max_extra_edges": topo.max_extra,
                   "stepping": head["what"], "reps": args.reps, "host_wait": wait_policy,
                   "parallelism": f"env-batch split x{D.world}, no step-path collective"},
        "roofline": dict(head["roofline"]),
        "check": rec["check"],
    }
    out["roofline"].update({"bytes_per_env_step": rec["bytes_per_env_step"], "layout_bytes_per_env_step": rec["layout_bytes_per_env_step"]})
    if head is per_tick and "single_launch" in per_tick:
        out["single_launch_per_tick"] = per_tick["single_launch"]
    if other is not None:
        out[other_key] = other
    if D.rank == 0:
        # also price the kernel against a device-copy bandwidth measured on this box
        bw = measured_copy_gbs(D.dev)
        out["roofline"]["measured_copy_peak"] = bw
        out["roofline"]["frac_of_measured_copy"] = out["roofline"]["achieved"] / bw if bw else None
    ident = D.identity()       # (a collective when N > 1: every rank calls it)
    if D.world > 1:
        out["ranks"] = ident
    if D.world > 1:
        # end-of-rollout gather of per-env returns over RCCL, checked against a single-rank recomputation:
        # restart from the initial state and run the first W + 4 ticks on every rank
        n_ticks = min(len(scripts), W + 4)
        env.load_state(init)
        for t in range(n_ticks):
            env.step(scripts[t])
        out["check"]["gathered_returns_match_single_rank"] = gathered_returns_check(D, env, scripts[:n_ticks], topo, init, cfg, L)
    if D.rank == 0 and D.world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(topo, init, cfg, M, L, scripts, W, args.cpu_seconds, args.cpu_threads)
    st = env.take_status()
    out["check"]["status_word"] = {"unpinned_scan": bool(st & 0x200), "busy_saturated": bool(st & 0x40)}
    env.close()
    del env, scripts
    if not args.no_closed_loop and name == "target" and not args.envs:
        per_tick_one = per_tick["single_launch"]["value"] if "single_launch" in per_tick else per_tick["value"]
        out["closed_loop_grid"] = closed_loop_block(D, per_tick_one, args.seed, n_per_gpu)
        out["ippo_rollout_collect"] = ippo_collect_block(D, args.seed, n_per_gpu)

    # the other BASELINE configs as short sub-records (parity-test sizes, here timed on the current kernels):
    # N = 1: cfg2 / cfg3 / cfg5; N > 1: cfg4 = the per-GPU shard of the 131072-env config
    if not args.no_configs and not args.envs and name == "target":
        others = ["cfg2", "cfg3", "cfg5"] if D.world == 1 else ["cfg4"]
        out["configs"] = {}
        for w in others:
            kk = min(K, 20 if w != "cfg5" else 10)
            r, objs = run_workload(D, w, WORKLOADS[w][0], kk, min(W, 5), min(args.reps, 5), DEFAULT_SUB[w], -1, args.seed, max_extra_of(w))
            objs[0].close()
            del objs
            b = brief(r)
            b["steps"] = kk
            out["configs"][w] = b
    if D.rank == 0:
        print(json.dumps(out))
    D.close()


if __name__ == "__main__":
    main()
