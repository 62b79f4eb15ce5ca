"""Synthetic fixed-topology networks for benchmarks and large-size tests
(SURVEY.md section 8d).  This is the build's own generator: the reference's network
construction (CDSimulator.py:407-597, volt_typhoon_env.py:1485-1900) is out of
scope; what is mirrored here are its *scaling knobs* so that the synthetic network
has the same roles and densities:

  n_dc     = ceil(M / 50) highest-degree devices      (volt_typhoon_env.py:1583, :1634)
  n_owned  = max(1, round(0.05 * M)) attacker-owned    (:1584)
  dense attacker edges (owned -> every other device) for M < 500
                                                       (CDSimulatorComponents.py:722-748)
  sparse attacker edges (owned -> every DC + k random) for M >= 500, k = round(log2(M)/2)
                                                       (volt_typhoon_env.py:1344-1462, :1472)
  star edges hub <-> other owned devices pre-materialised (CyberDefenseEnv.py:738-774)
  initial compromise ratio 0.4 over active devices     (volt_typhoon_env.py:45, :1846-1851)
  non-DC devices serve 'server' workloads, DCs 'client' (:1680-1684)

Rows of the CSR are sorted by neighbour id (python-igraph's neighbour order).
"""
from __future__ import annotations

import math

import numpy as np

from . import abi
from . import spec as S


def make_topology(M: int, n_blocks: int = 1, seed: int = 0, n_exploits: int = 2, vuln_frac: float = 0.5,
                  n_active: int | None = None, max_extra: int | None = None):
    """Returns (TopologyArrays, init_state dict with leading dim 1, EnvConfig defaults dict).
    max_extra: capacity of the per-env list of edges evolve_network may add (None: room for two full
    attacker stars, CyberDefenseEnv.py:738-774)."""
    rs = np.random.RandomState(seed * 7919 + M)
    n_active = M if n_active is None else int(n_active)
    blocks = np.array_split(np.arange(M), max(1, n_blocks))
    edges = set()
    indeg = np.zeros(M, np.int64)
    # directed Barabasi-Albert (m = 2) inside each block: new node -> 2 earlier nodes
    for blk in blocks:
        for j in range(1, len(blk)):
            v = blk[j]
            prev = blk[:j]
            w = indeg[prev] + 1.0
            k = min(2, j)
            tgt = rs.choice(prev, size=k, replace=False, p=w / w.sum())
            for u in tgt:
                edges.add((int(v), int(u)))
                indeg[u] += 1
    # sparse inter-block links
    if len(blocks) > 1:
        for bi, blk in enumerate(blocks):
            for _ in range(max(1, len(blk) // 8)):
                v = int(rs.choice(blk))
                ob = blocks[(bi + 1 + rs.randint(len(blocks) - 1)) % len(blocks)]
                u = int(rs.choice(ob))
                if u != v:
                    edges.add((v, u))
                    indeg[u] += 1
    outdeg = np.zeros(M, np.int64)
    for v, u in edges:
        outdeg[v] += 1
    deg = indeg + outdeg
    from .interchange import scaling_knobs
    n_dc, n_owned, _ = scaling_knobs(M)   # initialize_environment's knobs at the reference defaults (:1580-1591)
    order = np.argsort(-deg, kind="stable")
    dcs = order[:n_dc]
    owned = rs.choice(M, size=min(n_owned, M), replace=False)
    if M < 500:
        for a in owned:
            for u in range(M):
                if u != a:
                    edges.add((int(a), u))
    else:
        k = max(1, int(round(math.log2(max(2, M)) / 2)))
        non_dc = np.setdiff1d(np.arange(M), dcs)
        for a in owned:
            for u in dcs:
                if u != a:
                    edges.add((int(a), int(u)))
            for u in rs.choice(non_dc, size=min(k, len(non_dc)), replace=False):
                if u != a:
                    edges.add((int(a), int(u)))
        hub = int(np.sort(owned)[0])
        for a in owned:
            if a != hub:
                edges.add((hub, int(a)))
                edges.add((int(a), hub))
    el = np.array(sorted(edges), np.int64).reshape(-1, 2)
    out_ptr = np.zeros(M + 1, np.int32)
    out_ptr[1:] = np.cumsum(np.bincount(el[:, 0], minlength=M))
    out_col = el[:, 1].astype(np.int32)
    in_ptr, in_col, in_eid = abi.build_in_csr(M, out_ptr, out_col)

    dstatic = np.full(M, S.D_SERVER, np.uint8)
    dstatic[dcs] = S.D_DC
    vuln = np.zeros(M, np.uint8)
    vuln[rs.rand(M) < vuln_frac] |= 1
    non_dc_ids = np.setdiff1d(np.arange(M), dcs)
    vuln[rs.choice(non_dc_ids, size=min(5, len(non_dc_ids)), replace=False)] |= 1   # FortiOS 3.1 holders
    if n_exploits > 1:
        vuln[dcs] |= 2
    napps = np.full(M, 7, np.uint8)
    napps[dcs] = 5
    topo = abi.TopologyArrays(
        M=M, X=n_exploits, dstatic=dstatic, vuln=vuln, napps=napps,
        os_val=np.arange(M, dtype=np.float32), version=rs.choice([1.0, 2.0, 3.0], size=M).astype(np.float32),
        anomaly=np.zeros(M, np.float32), out_ptr=out_ptr, out_col=out_col,
        in_ptr=in_ptr, in_col=in_col, in_eid=in_eid).normalised()
    if max_extra is None:   # two stars of 2*(n_owned-1) edges, rounded up to a multiple of 4, at least 16
        max_extra = max(16, (4 * max(0, n_owned - 1) + 12 + 3) & ~3)
    topo.max_extra = int(max_extra)

    flags = np.zeros(M, np.uint8)
    active = np.zeros(M, bool)
    active[:n_active] = True
    active[dcs] = True
    active[owned] = True
    flags[~active] |= S.F_NYA
    flags[owned] |= (S.F_COMP | S.F_OWNED | S.F_KNOWN)
    for a in owned:   # one reachable neighbour per attacker-owned device (:1738-1841)
        row = out_col[out_ptr[a]:out_ptr[a + 1]]
        if len(row):
            flags[int(rs.choice(row))] |= S.F_REACH
    lucky = active & (rs.rand(M) < 0.4)
    flags[lucky] |= (S.F_COMP | S.F_KNOWN)
    wl = np.zeros(M, np.uint8)
    idle = np.where(active)[0]
    boot = rs.choice(idle, size=min(12, len(idle)), replace=False)
    wl[boot] = rs.randint(1, 6, size=len(boot))
    from .batched_env import initial_state_numpy  # local import: torch only needed there
    init = initial_state_numpy(topo, flags=flags, wl=wl)
    cfg = dict(num_of_device=n_active, n_att_actions=n_exploits + 3)
    return topo, init, cfg
