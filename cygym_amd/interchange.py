"""Snapshot interchange: ingest a network the REFERENCE built.

`from_reference_env(env)` reads a live `Volt_Typhoon_CyberDefenseEnv` object -- after `initialize_environment()`
(volt_typhoon_env.py:1485-1900) and `_rebuild_graph_cache()` (:456-473, what DoubleOracle.restore does,
do_agent.py:893) -- purely through its attributes (duck typing: nothing of the reference is imported here) and
returns what `BatchedCyberDefenseEnv` consumes: the shared topology + static per-device columns, the initial
struct-of-arrays state, and the scalar knobs.  So an experiment that pickles `initial_net_DO_its*.pkl`
(init_experiments.py:54-61) can be re-hosted on the GPU: unpickle with the reference on the path, flatten here.

Also mirrored, for callers that build networks without the reference (cygym_amd/topology.py):
  * the scaling knobs of initialize_environment (:1580-1591): `scaling_knobs`
  * the zero-day exploit bookkeeping (:1504-1563; per-rollout private exploit, volt_typhoon_do.py:1332-1415):
    `zero_day_bookkeeping` / `zero_day_mask`

The oracle harness (oracle/harness/ref_harness.py) exports its fixtures through these very functions, so every
golden fixture -- and `make_golden.py --check` -- exercises them against the reference's own objects.
"""
from __future__ import annotations

import math

import numpy as np

from . import abi
from . import spec as S


def scaling_knobs(num_of_device: int, scaling_vulnerability: bool = True, sv_dc_ratio: float = 50,
                  sv_attacker_fraction: float = 0.05, sv_apps_base: int = 3, sv_apps_per_device: float = 0.0):
    """(n_dc, n_owned, add_apps) as initialize_environment computes them (volt_typhoon_env.py:1580-1591;
    defaults :86-89)."""
    if not scaling_vulnerability:
        return 3, 5, 3
    n_dc = max(1, int(math.ceil(num_of_device / max(1.0, float(sv_dc_ratio)))))
    n_owned = max(1, int(round(num_of_device * float(sv_attacker_fraction))))
    add_apps = max(1, int(sv_apps_base + math.floor(num_of_device * float(sv_apps_per_device))))
    return n_dc, n_owned, add_apps


def zero_day_bookkeeping(n_exploits: int, k_known: int = 1, j_private: int = 0, private_pick=None):
    """The exploit-index sets of zero-day mode (volt_typhoon_env.py:1504-1563): exploits [0, k_known) are common
    knowledge, [k_known, k_known + j_private) form the unknown pool, of which the attacker privately holds
    `private_pick` (indices; default: the whole pool, which is what `_rng.choice(pool, size=j_private,
    replace=False)` returns as a set).  Returns dict(common, pool, private, prior_pi, owned_mask): `owned_mask` is
    EnvConfig.zero_day_owned_mask -- an exploit index outside it is replaced by a random owned one in the
    attacker's spread (:1131-1146)."""
    k_known = k_known if isinstance(k_known, int) and k_known >= 0 else 1
    j_private = j_private if isinstance(j_private, int) and j_private >= 0 else 0
    common = list(range(min(k_known, n_exploits)))
    pool = list(range(k_known, min(k_known + j_private, n_exploits)))
    private = list(pool) if private_pick is None else [int(i) for i in private_pick if int(i) in pool]
    prior = {i: 1.0 / len(pool) for i in pool} if pool else {}
    mask = 0
    for i in set(common) | set(private):
        mask |= 1 << i
    return dict(common=common, pool=pool, private=private, prior_pi=prior, owned_mask=mask)


def exploit_index_map(env):
    ids = [e.id for e in env.simulator.exploits]
    return {eid: i for i, eid in reversed(list(enumerate(ids)))}


def flatten_static(env):
    """Topology + static per-device columns, as the kernels consume them."""
    net = env.simulator.subnet.net
    ids = list(net.keys())
    M = len(ids)
    assert ids == list(range(M)), "device ids must be 0..M-1 in dict order"
    exps = env.simulator.exploits
    X = len(exps)
    dstatic = np.zeros(M, np.uint8)
    vuln = np.zeros(M, np.uint8)
    napps = np.zeros(M, np.uint8)
    os_val = np.zeros(M, np.float32)
    version = np.zeros(M, np.float32)
    anomaly = np.zeros(M, np.float32)
    for i, d in net.items():
        if getattr(d, "device_type", None) == "DomainController":
            dstatic[i] |= S.D_DC
        if d.wtype == "server":
            dstatic[i] |= S.D_SERVER
        napps[i] = min(255, len(d.apps))
        for e, ex in enumerate(exps):
            hit = any(v.id in ex.target for app in d.apps.values() for v in app.vulnerabilities.values())
            if hit:
                vuln[i] |= (1 << e)
        os_val[i] = env.os_to_float(d.OS)
        try:
            version[i] = float(d.version)
        except Exception:
            version[i] = -1.0
        a = d.anomaly_score
        anomaly[i] = -1.0 if a is None else float(a)
    out_ptr = np.zeros(M + 1, np.int32)
    out_col = []
    for u in range(M):
        nb = env._outnbrs.get(u, [])
        out_col.extend(int(v) for v in nb)
        out_ptr[u + 1] = len(out_col)
    out_col = np.asarray(out_col, np.int32)
    # in-CSR in the order of env._innbrs, each entry mapped to an out-CSR slot
    in_ptr = np.zeros(M + 1, np.int32)
    in_col, in_eid = [], []
    used = {}
    for v in range(M):
        for u in env._innbrs.get(v, []):
            u = int(u)
            k = used.get((u, v), 0)
            row = out_col[out_ptr[u]:out_ptr[u + 1]]
            pos = [j for j, w in enumerate(row) if w == v]
            assert k < len(pos), f"in-edge ({u},{v}) has no matching out entry"
            in_col.append(u)
            in_eid.append(int(out_ptr[u]) + pos[k])
            used[(u, v)] = k + 1
        in_ptr[v + 1] = len(in_col)
    return dict(M=M, X=X, dstatic=dstatic, vuln=vuln, napps=napps, os_val=os_val, version=version,
                anomaly=anomaly, out_ptr=out_ptr, out_col=out_col, in_ptr=in_ptr,
                in_col=np.asarray(in_col, np.int32), in_eid=np.asarray(in_eid, np.int32))


def flatten_config(env):
    return dict(
        turbo=int(bool(getattr(env, "turbo", False))),
        turbo_fraction_clients=float(getattr(env, "turbo_fraction_clients", 0.05)),
        turbo_fraction_servers=float(getattr(env, "turbo_fraction_servers", 0.02)),
        turbo_max_clients=int(getattr(env, "turbo_max_clients", 200)), turbo_max_servers=int(getattr(env, "turbo_max_servers", 40)),
        turbo_ramp_steps=int(getattr(env, "turbo_ramp_steps", 200)),
        turbo_train_max_logs=int(getattr(env, "turbo_train_max_logs", 256)), turbo_train_stride=int(getattr(env, "turbo_train_stride", 2)),
        num_of_device=int(env.numOfDevice), min_network_size=int(env.Min_network_size),
        max_exploits=int(env.MaxExploits), evolve_period=int(env._evolve_period),
        work_scale=float(env.work_scale), comp_scale=float(env.comp_scale), def_scale=float(env.def_scale),
        gamma=float(env.γ), lambda_events=float(env.lambda_events), p_add=float(env.p_add),
        p_attacker=float(env.p_attacker),
        workload_cap=(-1 if env.workload_cap is None else int(env.workload_cap)),
        workload_period_base=int(env.workload_period_base), workload_period_max=int(env.workload_period_max),
        scaling_vulnerability=int(bool(env.scaling_vulnerability)), fast_scan=int(bool(env.fast_scan)),
        n_att_actions=int(env.attacker_action_space.n), n_def_actions=int(env.defender_action_space.n),
        zero_day=int(bool(env.zero_day)), default_high=int(env.default_high),
        baseline={"Nash": 0, "No Defense": 1, "Preset": 2, "No Attack": 3}[env.base_line],
        zero_day_owned_mask=zero_day_mask(env),
    )


def zero_day_mask(env):
    if not getattr(env, "zero_day", False):
        return 0
    m = 0
    for i in set(env.common_exploit_indices) | set(env.private_exploit_indices):
        m |= (1 << int(i))
    return m


def flatten_dynamic(env, static):
    """Per-env mutable state as the SoA planes of include/cygym_spec.h."""
    net = env.simulator.subnet.net
    M = static["M"]
    emap = exploit_index_map(env)
    flags = np.zeros(M, np.uint8)
    busy = np.zeros(M, np.int32)
    wl = np.zeros(M, np.int32)
    comp_by = np.zeros(M, np.uint8)
    st_flags = np.zeros(M, np.uint8)
    st_busy = np.zeros(M, np.int32)
    st_wl = np.zeros(M, np.int32)
    st_comp_by = np.zeros(M, np.uint8)
    active_ids = getattr(env, "_active_ids", None)
    busy_set = env._busy_devices
    busy_ids = {d.id for d in busy_set} if not isinstance(busy_set, dict) else {d.id for d in busy_set.keys()}
    for i, d in net.items():
        f = 0
        if d.isCompromised: f |= S.F_COMP
        if d.attacker_owned: f |= S.F_OWNED
        if d.Known_to_attacker: f |= S.F_KNOWN
        if d.reachable_by_attacker: f |= S.F_REACH
        if d.Not_yet_added: f |= S.F_NYA
        if active_ids is not None and i in active_ids: f |= S.F_EVOACT
        if i in busy_ids: f |= S.F_BUSYC
        w = d.workload
        if w is not None:
            pt = int(w.processing_time or 0)
            wl[i] = pt
            if getattr(w, "adversarial", False): f |= S.F_WLADV
        flags[i] = f
        busy[i] = int(d.busy_time or 0)
        for eid in d.compromised_by:
            comp_by[i] |= (1 << emap[eid])
        st = env._device_ckpts.get(i)
        if st is not None:
            sf = S.S_VALID
            if st["isCompromised"]: sf |= S.F_COMP
            if st["Known_to_attacker"]: sf |= S.F_KNOWN
            if st["reachable_by_attacker"]: sf |= S.F_REACH
            if st["Not_yet_added"]: sf |= S.F_NYA
            sw = st["workload"]
            if sw:
                st_wl[i] = int(sw["processing_time"])
                if sw["adversarial"]: sf |= S.F_WLADV
            st_flags[i] = sf
            st_busy[i] = int(st["busy_time"])
            for eid in st["compromised_by"]:
                st_comp_by[i] |= (1 << emap[eid])
    E = len(static["out_col"])
    blocked = np.zeros(E, np.uint8)
    for (u, v) in env._blocked:
        lo, hi = static["out_ptr"][u], static["out_ptr"][u + 1]
        for j in range(lo, hi):
            if static["out_col"][j] == v:
                blocked[j] = 1
    # edges evolve_network added since the export (CyberDefenseEnv.py:738-843): the multiset difference
    # between the live neighbour cache and the exported CSR, as the env's extra-edge list
    K = int(static.get("max_extra", 0))
    xe = extra_edges(env, static)
    extra = abi.pack_extra(xe, [(u, v) in env._blocked for (u, v) in xe], K) if K > 0 else np.zeros(0, np.uint32)
    if K == 0:
        xe = []
    logs = env.simulator.logger.logs
    ring = np.full((S.LOG_RING, 2), -1, np.int32)
    tail = logs[-S.LOG_RING:]
    base = len(logs) - len(tail)
    for k, l in enumerate(tail):
        ring[(base + k) % S.LOG_RING] = (int(l["from_device"]), int(l["to_device"]))
    hist = np.full((S.HIST_RING, 2), 0xFFFF, np.uint16)     # the long history Detector.train fits on
    htail = logs[-S.HIST_RING:]
    hbase = len(logs) - len(htail)
    if htail:
        idx = (hbase + np.arange(len(htail))) % S.HIST_RING
        hist[idx, 0] = [int(l["from_device"]) for l in htail]
        hist[idx, 1] = [int(l["to_device"]) for l in htail]
    ienv = np.zeros(S.I_COUNT, np.int64)
    ienv[S.I_STEP_NUM] = env.step_num
    ienv[S.I_DEF_STEP] = env.defender_step
    ienv[S.I_ATT_STEP] = env.attacker_step
    ienv[S.I_WORK_DONE] = env.work_done
    ienv[S.I_CKPT_CNT] = env.checkpoint_count
    ienv[S.I_REVERT_CNT] = env.revert_count
    ienv[S.I_SCAN_CNT] = env.scan_cnt
    ienv[S.I_COMP_CNT] = env.compromised_devices_cnt
    ienv[S.I_EDGES_BLOCKED] = env.edges_blocked
    ienv[S.I_EDGES_ADDED] = env.edges_added
    ef = 0
    if env.checkpoint is not None: ef |= S.E_HAS_CKPT
    if active_ids is not None: ef |= S.E_EVO_INIT
    det = env.simulator.detector
    if det.trained: ef |= S.E_DET_TRAIN
    if det.random_detection: ef |= S.E_DET_RANDOM
    if getattr(env, "_prev_att_potential", None) is not None: ef |= S.E_PREV_SET
    ienv[S.I_FLAGS] = ef | (len(xe) << S.E_NX_SHIFT)
    ienv[S.I_LOG_TOTAL] = len(logs)
    disc = 0
    for e, ex in enumerate(env.simulator.exploits):
        if getattr(ex, "discovered", False):
            disc |= (1 << e)
    ienv[S.I_DISCOVERED] = disc
    fenv = np.zeros(S.D_COUNT, np.float64)
    fenv[S.D_DEF_COST] = env.defensive_cost
    fenv[S.D_CLEAN_COST] = env.clearning_cost
    pp = getattr(env, "_prev_att_potential", None)
    fenv[S.D_PREV_ATT_POT] = 0.0 if pp is None else float(pp)
    # a trained detector travels with the env: its fitted forest, flattened (cygym_spec.h).  Header word 5 (the
    # request the installed forest answers) equals word 3 (no request of this env's ticks is outstanding).
    forest = np.zeros(S.FOREST_WORDS, np.uint32)
    if det.trained:
        from . import detector as D
        forest[:] = D.flatten_forest(det.model)
        forest[3] = forest[5] = 0
    # Device.anomaly_score as it stands (the per-log scan path rewrites it, volt_typhoon_env.py:1033-1038); -1 = None
    anomaly = np.full(M, -1.0, np.float32)
    for i, d in net.items():
        a = getattr(d, "anomaly_score", None)
        anomaly[i] = -1.0 if a is None else float(a)
    return dict(flags=flags, busy=busy, wl=wl, comp_by=comp_by, st_flags=st_flags, st_busy=st_busy,
                st_wl=st_wl, st_comp_by=st_comp_by, blocked=blocked, ring=ring, ienv=ienv, fenv=fenv,
                extra=extra, hist=hist, forest=forest, anomaly=anomaly)


def extra_edges(env, static):
    """Sorted (u, v) list of the edges present in env._outnbrs but not in the exported CSR.  Also checks
    the two ordering facts the flat restatement relies on: a rebuilt neighbour row is the exported row
    merged with the added edges by ascending neighbour id, and env._active_ids iterates ascending."""
    from collections import Counter
    M = static["M"]
    op, oc = static["out_ptr"], static["out_col"]
    ip, ic = static["in_ptr"], static["in_col"]
    added = []
    for u in range(M):
        base = [int(v) for v in oc[op[u]:op[u + 1]]]
        cur = [int(v) for v in env._outnbrs.get(u, [])]
        diff = Counter(cur) - Counter(base)
        assert not (Counter(base) - Counter(cur)), f"edges of {u} disappeared"
        for v, c in diff.items():
            assert c == 1 and v not in base, f"added edge ({u},{v}) duplicates an existing one"
            added.append((u, int(v)))
        if diff:
            xs = sorted(diff)
            merged, j = [], 0
            for v in base:
                while j < len(xs) and xs[j] < v:
                    merged.append(xs[j]); j += 1
                merged.append(v)
            merged.extend(xs[j:])
            assert merged == cur, f"row {u}: merged order {merged} != cache {cur}"
    added.sort()
    if added:
        for d in {v for _, v in added}:
            base = [int(u) for u in ic[ip[d]:ip[d + 1]]]
            xs = sorted(u for (u, v) in added if v == d)
            merged, j = [], 0
            for u in base:
                while j < len(xs) and xs[j] < u:
                    merged.append(xs[j]); j += 1
                merged.append(u)
            merged.extend(xs[j:])
            assert merged == [int(u) for u in env._innbrs.get(d, [])], f"in-row {d} order"
    return added


def from_reference_env(env, max_extra: int = 0):
    """(TopologyArrays, init_state dict with leading dim 1, EnvConfig keyword dict) of a reference env object."""
    static = flatten_static(env)
    static["max_extra"] = int(max_extra)
    cfg = flatten_config(env)
    dyn = flatten_dynamic(env, static)
    topo = abi.TopologyArrays(
        M=static["M"], X=static["X"], dstatic=static["dstatic"], vuln=static["vuln"], napps=static["napps"],
        os_val=static["os_val"], version=static["version"], anomaly=static["anomaly"], out_ptr=static["out_ptr"],
        out_col=static["out_col"], in_ptr=static["in_ptr"], in_col=static["in_col"], in_eid=static["in_eid"],
        max_extra=int(max_extra)).normalised()
    init = {k: np.asarray(v)[None] for k, v in dyn.items()}
    kw = {k: v for k, v in cfg.items() if k not in ("baseline",)}
    kw["baseline"] = {v: k for k, v in abi.BASELINES.items()}[cfg["baseline"]]
    return topo, init, kw
