"""Diagnostic (stamps build, -DCG_STAMPS): which envs set the launch time of the per-tick kernel?

A launch of 4096 envs x 256 devices is ONE residency round (one wave per env, 16 waves per CU), so it lasts as long as its
slowest env.  Every earlier stamp report gave MEANS per phase; this one records, for every env of every tick, its lifetime
(s_memtime cycles from wave entry to the end of the write-back) and what it was doing, then describes the MAXIMUM:
histograms per action type, the make-up of the slowest 1 % (action type, list length, spread sweeps / conflicts / source
count / hub rows, block passes), the env that finished last in each launch, and the phase split of the slowest envs.

    python tools/tail_hist.py [envs] [M] [ticks] > profiles/r04_tail_hist.txt      (GPU box; writes the .json next to it)
"""
import ctypes as C, json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cygym_amd import abi, build as B
so = os.environ.get("TAIL_SO") or os.path.join(ROOT, "cygym_amd", "libcygym_hip_stamps.so")   # TAIL_SO: a prebuilt experiment variant (tools/exp_build.sh)
if not ((os.environ.get("CYGYM_STAMP_NOBUILD") or os.environ.get("TAIL_SO")) and os.path.exists(so)):
    B.build_to(so, None, flags=["-DCG_STAMPS"], dev_mt=int(os.environ["CYGYM_STAMP_MT"]) if "CYGYM_STAMP_MT" in os.environ else None)
from cygym_amd import _lib
_lib.SO = so
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 205
OUT = os.environ.get("TAIL_JSON", os.path.join(ROOT, "gpurun_out", "tail_hist.json"))
W = 28   # CG_DBG_W
WPB_ENV = int(os.environ.get("TAIL_WPB", "16"))   # envs per workgroup of the launch shape (16 at 4096 x 256)
topo, init, ck = make_topology(M, {64: 4, 256: 1, 2048: 32}.get(M, 1), seed=0, max_extra=int(os.environ.get("CYGYM_STAMP_MAX_EXTRA", "0")))
cfg = abi.EnvConfig(seed=0, auto_reset=1, lambda_events=0.0, **ck)   # bench.py's `target` workload
env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=max(1, M // 8))
env.lib.cygym_set_debug.argtypes = [C.c_void_p, C.c_void_p]
dbg = torch.zeros((N, W), dtype=torch.int64, device="cuda:0")
env.lib.cygym_set_debug(env._h, C.c_void_p(dbg.data_ptr()))
deg = np.diff(np.asarray(topo.out_ptr))

rows = []   # per tick: dict of arrays
for t in range(T):
    env.gen_actions(t)
    dbg.zero_()
    env.step()
    torch.cuda.synchronize()
    d = dbg.cpu().numpy().copy()
    cnt = env.act["dev_cnt"].cpu().numpy()[:, 0].copy()
    rows.append((t, d, cnt))

BIN = 2000
def hist(x, lo=0, hi=70000):
    h, _ = np.histogram(np.clip(x, lo, hi - 1), bins=np.arange(lo, hi + BIN, BIN))
    return h.tolist()

report = {"workload": f"{N} envs x {M} devices, bench.py `target` script (seed 0), ticks 0..{T - 1}, stamps build (each stamp costs a few hundred cycles)",
          "bin_cycles": BIN, "parities": {}}
for par, pname in ((0, "defender"), (1, "attacker")):
    sel_all = [(t, d, c) for (t, d, c) in rows if (t & 1) == par and t >= 5]
    # launches in which the periodic workload arrivals are due (volt_typhoon_env.py:575-596: every env of the batch has the same
    # step_num, so they all generate arrivals in the same tick): reported on their own, the histograms describe the other ticks
    is_arr = lambda d: float(np.median(d[:, 3] - d[:, 2])) > 5000.0   # noqa: E731
    arr = [(t, d, c) for (t, d, c) in sel_all if is_arr(d)]
    sel = [(t, d, c) for (t, d, c) in sel_all if not is_arr(d)]
    life = np.concatenate([(d[:, 7] - d[:, 0]) + (d[:, 9] >> 8) for (_, d, _) in sel])   # + cold kernarg round trip before stamp 0
    at = np.concatenate([d[:, 8] for (_, d, _) in sel])
    allph = np.concatenate([np.diff(d[:, :8], axis=1) for (_, d, _) in sel])
    names = ["stage", "action", "work+arrivals", "counts", "obs", "evolve+busyc", "writeback"]
    P = {"ticks": len(sel), "env_ticks": int(life.size),
         "arrival_ticks": {"ticks": [int(t) for (t, _, _) in arr],
                           "lifetime_mean": float(np.mean([((d[:, 7] - d[:, 0]) + (d[:, 9] >> 8)).mean() for (_, d, _) in arr])) if arr else None,
                           "lifetime_max": int(max(((d[:, 7] - d[:, 0]) + (d[:, 9] >> 8)).max() for (_, d, _) in arr)) if arr else None,
                           "work_arrivals_phase_mean": float(np.mean([(d[:, 3] - d[:, 2]).mean() for (_, d, _) in arr])) if arr else None},
         "lifetime": {"mean": float(life.mean()), "p50": float(np.median(life)), "p90": float(np.percentile(life, 90)),
                      "p99": float(np.percentile(life, 99)), "p99.9": float(np.percentile(life, 99.9)), "max": int(life.max())},
         "hist_all": hist(life), "by_atype": {}}
    for a in sorted(set(at.tolist())):
        m = at == a
        P["by_atype"][int(a)] = {"n": int(m.sum()), "share": float(m.mean()), "mean": float(life[m].mean()), "p99": float(np.percentile(life[m], 99)),
                                 "max": int(life[m].max()), "hist": hist(life[m]),
                                 "phase_mean": {n: float(allph[m, i].mean()) for i, n in enumerate(names)}}
    # the launch: span from the first wave's entry to the last wave's end, and who finished last.  s_memtime counters are
    # not synchronised across the chip, so spans come from s_memrealtime (100 MHz, one counter for the whole chip: 10 ns =
    # ~24 shader cycles of resolution), stamped at wave entry and after the write-back; cycles per tick of it from the envs'
    # own (s_memtime lifetime / realtime lifetime) ratio.
    spans, last_at, last_life, last_start, p99_end, start_spread, per_tick = [], [], [], [], [], [], []
    slow_rows = []
    for (t, d, c) in sel_all:
        rt0, rt1 = d[:, 19].astype(np.int64), d[:, 20].astype(np.int64)
        lf = (d[:, 7] - d[:, 0]) + (d[:, 9] >> 8)
        cyc_per_rt = float(np.median(lf[rt1 > rt0] / (rt1 - rt0)[rt1 > rt0]))   # shader cycles per 10 ns
        t0 = rt0.min()
        i = int(np.argmax(rt1))
        work = d[:, 3] - d[:, 2]
        arr_env = work > 6000      # this env generated workload arrivals in this tick
        rec = {"tick": int(t), "span_ns": int(rt1.max() - t0) * 10, "clock_ghz": cyc_per_rt / 10.0, "start_spread_ns": int(rt0.max() - t0) * 10,
               "last_env_atype": int(d[i, 8]), "last_env_lifetime": int(lf[i]), "last_env_start_ns": int(rt0[i] - t0) * 10,
               "last_env_had_arrivals": bool(arr_env[i]), "life_max": int(lf.max()), "life_p99": float(np.percentile(lf, 99)),
               "envs_with_arrivals": int(arr_env.sum()), "life_max_without_arrivals": int(lf[~arr_env].max()) if (~arr_env).any() else 0,
               "xcds_seen": int(len(set((d[:, 21] & 0xF).tolist()))), "arrival_launch": bool(is_arr(d))}
        per_tick.append(rec)
        if is_arr(d):
            continue
        spans.append(rec["span_ns"]); last_at.append(rec["last_env_atype"]); last_life.append(rec["last_env_lifetime"]); last_start.append(rec["last_env_start_ns"])
        start_spread.append(rec["start_spread_ns"]); p99_end.append(float(np.percentile(rt1 - t0, 99)) * 10)
        k = max(1, N // 100)
        for j in np.argsort(lf)[-k:]:
            slow_rows.append((t, int(j), int(d[j, 8]), int(lf[j]), d[j].copy(), int(c[j])))
    P["per_tick"] = per_tick
    P["launch"] = {"span_ns_mean": float(np.mean(spans)), "span_ns_max": int(np.max(spans)), "span_ns_min": int(np.min(spans)),
                   "span_ns_p50": float(np.median(spans)), "end_p99_ns_mean": float(np.mean(p99_end)),
                   "wave_start_spread_ns_mean": float(np.mean(start_spread)),
                   "clock_ghz_mean": float(np.mean([r["clock_ghz"] for r in per_tick])),
                   "last_env_atype_counts": {int(a): int((np.array(last_at) == a).sum()) for a in sorted(set(last_at))},
                   "last_env_lifetime_mean": float(np.mean(last_life)), "last_env_start_offset_ns_mean": float(np.mean(last_start)),
                   "launches_whose_last_env_had_arrivals": int(sum(r["last_env_had_arrivals"] for r in per_tick if not r["arrival_launch"])),
                   "envs_with_arrivals_per_launch_mean": float(np.mean([r["envs_with_arrivals"] for r in per_tick])),
                   "note": "s_memrealtime (10 ns ticks); span = first wave entry .. last wave end over the whole launch; arrival launches excluded"}
    # the slowest 1 % of every launch
    sa = np.array([r[2] for r in slow_rows]); sl = np.array([r[3] for r in slow_rows])
    S = {"n": len(slow_rows), "lifetime_mean": float(sl.mean()), "lifetime_min": int(sl.min()),
         "atype_counts": {int(a): int((sa == a).sum()) for a in sorted(set(sa.tolist()))}, "detail": {}}
    for a in sorted(set(sa.tolist())):
        sub = [r for r in slow_rows if r[2] == a]
        dd = np.stack([r[4] for r in sub]); ph = np.diff(dd[:, :8], axis=1)
        det = {"n": len(sub), "lifetime_mean": float(np.mean([r[3] for r in sub])), "entry_to_first_param_mean": float((dd[:, 9] >> 8).mean()),
               "phase_mean": {n: float(ph[:, i].mean()) for i, n in enumerate(names)}}
        if par == 1 and a == 1:   # spread: sub-phases and counters
            det["spread"] = {"setup": float((dd[:, 10] - dd[:, 1]).mean()), "sweeps": float((dd[:, 11] - dd[:, 10]).mean()),
                             "apply+logcnt": float((dd[:, 12] - dd[:, 11]).mean()), "ring": float((dd[:, 13] - dd[:, 12]).mean()),
                             "n_sweeps_hist": np.bincount(dd[:, 15].astype(int), minlength=6).tolist(),
                             "n_src_mean": float(dd[:, 16].mean()), "n_src_max": int(dd[:, 16].max()),
                             "full_row_sources_mean": float((dd[:, 17] & 0xFFFF).mean()), "coop_rows_mean": float((dd[:, 17] >> 16).mean()),
                             "conflicting_lanes_mean": float((dd[:, 18] & 0xFFFF).mean()), "rescans_in_later_sweeps_mean": float((dd[:, 18] >> 16).mean())}
        if par == 0 and a in (6, 9):
            det["block"] = {"pre": float((dd[:, 10] - dd[:, 1]).mean()), "loop": float((dd[:, 11] - dd[:, 10]).mean()), "loop_max": int((dd[:, 11] - dd[:, 10]).max()),
                            "rest": float((dd[:, 2] - dd[:, 11]).mean()), "passes_hist": np.bincount(np.minimum(dd[:, 15].astype(int), 15), minlength=16).tolist(),
                            "entries_mean": float(dd[:, 14].mean()), "list_len_mean": float(np.mean([r[5] for r in sub])),
                            "not_simple_share": float((dd[:, 13] & 1).mean())}
        S["detail"][int(a)] = det
    P["slowest_1pct"] = S
    # all spread / block envs (not only the tail): lifetime against the counters, to see what the tail shares
    if par == 1:
        dall = np.concatenate([d[d[:, 8] == 1] for (_, d, _) in sel])
        lf = (dall[:, 7] - dall[:, 0]) + (dall[:, 9] >> 8)
        ns = dall[:, 15].astype(int)
        P["spread_all"] = {"n": int(len(dall)), "by_sweeps": {int(k): {"n": int((ns == k).sum()), "life_mean": float(lf[ns == k].mean()), "life_max": int(lf[ns == k].max()),
                           "sweep_cycles_mean": float((dall[ns == k, 11] - dall[ns == k, 10]).mean())} for k in sorted(set(ns.tolist()))},
                           "corr_life_nsrc": float(np.corrcoef(lf, dall[:, 16])[0, 1]), "corr_life_conflicts": float(np.corrcoef(lf, dall[:, 18] & 0xFFFF)[0, 1]),
                           "n_src_p50": float(np.median(dall[:, 16])), "n_src_p99": float(np.percentile(dall[:, 16], 99)),
                           "setup_mean": float((dall[:, 10] - dall[:, 1]).mean()), "sweeps_mean": float((dall[:, 11] - dall[:, 10]).mean()),
                           "apply_logcnt_mean": float((dall[:, 12] - dall[:, 11]).mean()), "ring_mean": float((dall[:, 13] - dall[:, 12]).mean())}
    else:
        for a in (6, 9):
            dall = np.concatenate([d[d[:, 8] == a] for (_, d, _) in sel])
            if not len(dall):
                continue
            lf = (dall[:, 7] - dall[:, 0]) + (dall[:, 9] >> 8)
            npass = dall[:, 15].astype(int)
            P[f"block_all_{a}"] = {"n": int(len(dall)), "by_passes": {int(k): {"n": int((npass == k).sum()), "life_mean": float(lf[npass == k].mean()), "life_max": int(lf[npass == k].max()),
                                   "loop_mean": float((dall[npass == k, 11] - dall[npass == k, 10]).mean())} for k in sorted(set(npass.tolist()))},
                                   "corr_life_entries": float(np.corrcoef(lf, dall[:, 14])[0, 1]), "corr_life_passes": float(np.corrcoef(lf, npass)[0, 1])}
    report["parities"][pname] = P

os.makedirs(os.path.dirname(OUT), exist_ok=True)
json.dump(report, open(OUT, "w"), indent=1)

if os.environ.get("TAIL_BRIEF"):   # one line per parity: experiment loops (tools/exp_tail.sh)
    for pname, P in report["parities"].items():
        la, heavy = P["launch"], ({1: "spread"} if pname == "attacker" else {6: "block", 9: "unblock"})
        msg = f"{os.path.basename(so):32s} {pname:8s} span {la['span_ns_mean']:7.0f} ns (p50 {la['span_ns_p50']:.0f}) clk {la['clock_ghz_mean']:.2f} last-env life {la['last_env_lifetime_mean']:.0f} | all p50 {P['lifetime']['p50']:.0f} p99 {P['lifetime']['p99']:.0f}"
        for a, nm in heavy.items():
            if a in P["by_atype"]:
                v = P["by_atype"][a]
                msg += f" | {nm} mean {v['mean']:.0f} p99 {v['p99']:.0f} action {v['phase_mean']['action']:.0f}"
        if "spread_all" in P:
            sa = P["spread_all"]
            msg += f" [setup {sa['setup_mean']:.0f} sweeps {sa['sweeps_mean']:.0f} apply+cnt {sa['apply_logcnt_mean']:.0f} ring {sa['ring_mean']:.0f}]"
        ph = P["by_atype"][3 if pname == "attacker" else 8]["phase_mean"]
        msg += " | noop: " + " ".join(f"{k[:5]} {x:.0f}" for k, x in ph.items())
        print(msg)
    sys.exit(0)
# ---- readable summary ----
print(report["workload"])
for pname, P in report["parities"].items():
    L = P["lifetime"]
    print(f"\n== {pname} ticks ({P['ticks']} launches, {P['env_ticks']} env-ticks) ==")
    print(f"(launches with workload arrivals due, reported apart: {P['arrival_ticks']})")
    print(f"env lifetime cycles: mean {L['mean']:.0f}  p50 {L['p50']:.0f}  p90 {L['p90']:.0f}  p99 {L['p99']:.0f}  p99.9 {L['p99.9']:.0f}  max {L['max']}")
    la = P["launch"]
    print(f"launch span ns (first wave entry .. last wave end, s_memrealtime): mean {la['span_ns_mean']:.0f}  p50 {la['span_ns_p50']:.0f}  min {la['span_ns_min']}  max {la['span_ns_max']};  99th-percentile env end {la['end_p99_ns_mean']:.0f};  wave-start spread {la['wave_start_spread_ns_mean']:.0f};  shader clock {la['clock_ghz_mean']:.2f} GHz")
    print(f"last-finishing env: action types {la['last_env_atype_counts']}, its lifetime {la['last_env_lifetime_mean']:.0f} cycles, its start offset {la['last_env_start_offset_ns_mean']:.0f} ns; "
          f"launches whose last env generated arrivals: {la['launches_whose_last_env_had_arrivals']}; envs with arrivals per launch {la['envs_with_arrivals_per_launch_mean']:.1f}")
    print("per tick: tick span_ns last_atype last_life last_start_ns arrivals? | life_max life_p99 envs_with_arrivals life_max_without_arrivals")
    for r in P["per_tick"][:40]:
        print(f"   {r['tick']:4d} {r['span_ns']:6d} {r['last_env_atype']:3d} {r['last_env_lifetime']:6d} {r['last_env_start_ns']:5d} {int(r['last_env_had_arrivals'])} | {r['life_max']:6d} {r['life_p99']:8.0f} {r['envs_with_arrivals']:5d} {r['life_max_without_arrivals']:6d}  xcds {r['xcds_seen']}")
    print("per action type:  n  share  mean  p99  max | phase means")
    for a, v in P["by_atype"].items():
        print(f"  atype {a:3d}: {v['n']:7d} {v['share']:.3f} {v['mean']:8.0f} {v['p99']:8.0f} {v['max']:8d} | " + " ".join(f"{k} {x:.0f}" for k, x in v["phase_mean"].items()))
    print(f"histogram of all lifetimes ({BIN}-cycle bins from 0): {P['hist_all']}")
    S = P["slowest_1pct"]
    print(f"slowest 1 % of every launch: n {S['n']}, lifetime mean {S['lifetime_mean']:.0f} (min {S['lifetime_min']}), action types {S['atype_counts']}")
    for a, det in S["detail"].items():
        print(f"  atype {a}: n {det['n']} lifetime {det['lifetime_mean']:.0f} entry->param {det['entry_to_first_param_mean']:.0f} | " + " ".join(f"{k} {x:.0f}" for k, x in det["phase_mean"].items()))
        for k in ("spread", "block"):
            if k in det:
                print(f"     {k}: {det[k]}")
    for k in ("spread_all", "block_all_6", "block_all_9"):
        if k in P:
            print(f"  {k}: {json.dumps(P[k])}")
