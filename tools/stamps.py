"""Diagnostic: per-phase cycle shares of the tick kernel (build with -DCG_STAMPS).
Usage on the GPU box: python tools/stamps.py [envs] [M]"""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from cygym_amd import abi, build as B
so = os.path.join(ROOT, "cygym_amd", "libcygym_hip_stamps.so")
if not (os.environ.get("CYGYM_STAMP_NOBUILD") and os.path.exists(so)):   # (prebuilt in the build container: saves GPU-box minutes)
    B.build_to(so, None, flags=["-DCG_STAMPS", *os.environ.get("CYGYM_EXTRA_FLAGS", "").split()], dev_mt=int(os.environ["CYGYM_STAMP_MT"]) if "CYGYM_STAMP_MT" in os.environ else None)
from cygym_amd import _lib
_lib.SO = so
from cygym_amd.batched_env import BatchedCyberDefenseEnv
from cygym_amd.topology import make_topology
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = int(sys.argv[2]) if len(sys.argv) > 2 else 256
topo, init, ck = make_topology(M, {64: 4, 256: 1, 2048: 32}.get(M, 1), seed=0, max_extra=int(os.environ.get("CYGYM_STAMP_MAX_EXTRA", "0")))   # 0: the lean kernels bench.py runs
cfg = abi.EnvConfig(seed=0, auto_reset=1, lambda_events=0.0, **ck)
env = BatchedCyberDefenseEnv(topo, cfg, N, init, device="cuda:0", max_groups=1, max_devs=max(1, M // 8))
env.lib.cygym_set_debug.argtypes = [C.c_void_p, C.c_void_p]
dbg = torch.zeros((N, 28), dtype=torch.int64, device="cuda:0")   # CG_DBG_W of csrc/cg_params.hpp
env.lib.cygym_set_debug(env._h, C.c_void_p(dbg.data_ptr()))
names = ["stage", "action", "work+arrivals", "counts", "obs", "evolve+busyc", "writeback"]
if os.environ.get("CYGYM_STAMP_ROLLOUT"):   # phases of the LAST tick of a T-tick cygym_rollout, per action type
    T = int(os.environ["CYGYM_STAMP_ROLLOUT"])
    for T_run in (T, T + 1):   # last tick defender / attacker
        env.load_state(init)
        act, out = env.alloc_rollout(T_run)
        env.gen_actions_rollout(0, act)
        env.rollout(act, out)
        torch.cuda.synchronize()
        d = dbg.cpu().numpy()
        st = d[:, :8]
        seg = np.diff(st[:, :7], axis=1)
        tot = st[:, 6] - st[:, 0]
        at = d[:, 8]
        print(f"rollout T={T_run} last tick mode {int(d[0, 9] & 0xFF)}: tick cycles mean {tot.mean():.0f} p50 {np.median(tot):.0f} p99 {np.percentile(tot, 99):.0f} max {tot.max()}")
        print("   mean per phase:", {n: int(seg[:, i].mean()) for i, n in enumerate(names[:6])})
        for a in sorted(set(at.tolist())):
            m = at == a
            print(f"   atype {int(a):3d}: n={int(m.sum()):5d} tick mean {tot[m].mean():8.0f} max {tot[m].max():8d}  top {seg[m, 0].mean():7.0f} action {seg[m, 1].mean():8.0f} work {seg[m, 2].mean():6.0f} counts {seg[m, 3].mean():6.0f} obs {seg[m, 4].mean():6.0f} evolve {seg[m, 5].mean():6.0f}")
        if int(d[0, 9] & 0xFF) == 1 and (at == 1).any():
            sub = d[at == 1]
            print("   DIAG spread: sweep0 lane part", int(sub[:, 12].mean()), "sweep0 coop part", int(sub[:, 13].mean()), "later sweeps", int(sub[:, 14].mean()), "rounds total", int((sub[:, 11] - sub[:, 10]).mean()))
            print("   spread sub-phases (mean cycles): setup", int((sub[:, 10] - sub[:, 1]).mean()), "rounds", int((sub[:, 11] - sub[:, 10]).mean()),
                  "logcnt", int((sub[:, 12] - sub[:, 11]).mean()), "ring", int((sub[:, 13] - sub[:, 12]).mean()),
                  "apply", int((sub[:, 14] - sub[:, 13]).mean()), " n_rounds mean", sub[:, 15].mean(), "max", sub[:, 15].max())
        if int(d[0, 9] & 0xFF) == 0:
            for aa in (6, 9):
                if (at == aa).any():
                    sub = d[at == aa]
                    print(f"   action {aa}: pre {int((sub[:, 10] - sub[:, 1]).mean())} loop {int((sub[:, 11] - sub[:, 10]).mean())} (max {int((sub[:, 11] - sub[:, 10]).max())}) rest {int((sub[:, 2] - sub[:, 11]).mean())}"
                          f" passes mean {sub[:, 15].mean():.2f} max {sub[:, 15].max()} entries mean {sub[:, 14].mean():.1f}")
    sys.exit(0)
for t in range(40):
    env.gen_actions(t)
    env.step()
    if t < 30:
        continue
    torch.cuda.synchronize()
    d = dbg.cpu().numpy()
    st = d[:, :8]
    seg = np.diff(st, axis=1)
    tot = st[:, 7] - st[:, 0]
    at = d[:, 8]
    print(f"tick {t} mode {int(d[0, 9] & 0xFF)}: wave lifetime cycles mean {tot.mean():.0f} p50 {np.median(tot):.0f} p99 {np.percentile(tot, 99):.0f} max {tot.max()}  "
          f"span(first start..last end) {st[:, 7].max() - st[:, 0].min()}")
    print("   mean per phase:", {n: int(seg[:, i].mean()) for i, n in enumerate(names)})
    ent = d[:, 9] >> 8
    print(f"   entry -> first parameter use (cold kernarg round trip): mean {ent.mean():.0f} min {ent.min()} max {ent.max()} cycles")
    if int(d[0, 9] & 0xFF) == 1:
        m = at == 1
        if m.any():
            sub = d[m]
            if os.environ.get("CYGYM_SRC"):
                r2 = sub[:, 15] >= 2
                print("   DIAG rounds: round0", int((sub[:, 12] - sub[:, 10]).mean()), "round1", int((sub[r2, 13] - sub[r2, 12]).mean()),
                      "rest-of-rounds", int((sub[r2, 11] - sub[r2, 13]).mean()), "n>=2 frac", r2.mean())
            print("   spread sub-phases (mean cycles): setup", int((sub[:, 10] - sub[:, 1]).mean()), "rounds", int((sub[:, 11] - sub[:, 10]).mean()),
                  "logcnt", int((sub[:, 12] - sub[:, 11]).mean()), "ring", int((sub[:, 13] - sub[:, 12]).mean()),
                  "apply", int((sub[:, 14] - sub[:, 13]).mean()), " n_rounds mean", sub[:, 15].mean(), "max", sub[:, 15].max())
    if int(d[0, 9] & 0xFF) == 0:
        for aa in (6, 9):
            m = at == aa
            if m.any():
                sub = d[m]
                print(f"   action {aa}: pre {int((sub[:, 10] - sub[:, 1]).mean())} loop {int((sub[:, 11] - sub[:, 10]).mean())} (max {int((sub[:, 11] - sub[:, 10]).max())}) rest {int((sub[:, 2] - sub[:, 11]).mean())}"
                      f" passes mean {sub[:, 15].mean():.2f} max {sub[:, 15].max()} entries mean {sub[:, 14].mean():.1f}")
    if os.environ.get("CYGYM_STAMP_TAIL"):   # the slowest envs of this launch: what were they doing?
        dl = env.act["dev_idx"].cpu().numpy(); dc = env.act["dev_cnt"].cpu().numpy()[:, 0]
        deg = np.diff(np.asarray(topo.out_ptr))
        for i in np.argsort(tot)[-4:][::-1]:
            lst = dl[i, : dc[i]]
            print(f"   TAIL env {i}: atype {int(at[i])} total {tot[i]} phases {seg[i].tolist()} loop {d[i, 11] - d[i, 10]} passes {d[i, 15]} n_act {d[i, 14]} flags {d[i, 13]} "
                  f"list len {len(lst)} distinct {len(set(lst.tolist()))} max out-degree in list {deg[lst].max() if len(lst) else 0} long rows in list {(deg[lst] > 8).sum() if len(lst) else 0}")
    for a in sorted(set(at.tolist())):
        m = at == a
        print(f"   atype {int(a):3d}: n={int(m.sum()):5d} total mean {tot[m].mean():8.0f} max {tot[m].max():8d}  action-phase mean {seg[m, 1].mean():8.0f} max {seg[m, 1].max():8d}")
