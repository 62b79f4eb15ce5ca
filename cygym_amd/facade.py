"""Read-only object facade over one env of the batch.

Some reference agents walk the simulator's object graph instead of the flat
observation: `env._get_ordered_devices()` (IPPO.py:86 build_visibility_mask),
`env.simulator.subnet.net` (HMARL.py:126-137, 339, 470, 539), `.graph.get_edgelist()`
(meta_hierarchical_br.py:74-96), `env.simulator.exploits`, `logger.get_logs()`.
These classes answer those reads from the struct-of-arrays state; they hold a
host copy of the env's planes taken when the facade object was created, and are
read-only (the tick is the only writer).
"""
from __future__ import annotations

import numpy as np

from . import spec as S


class WorkloadView:
    __slots__ = ("processing_time", "adversarial", "wtype", "assigned")

    def __init__(self, processing_time, adversarial, wtype):
        self.processing_time = int(processing_time)
        self.adversarial = bool(adversarial)
        self.wtype = wtype
        self.assigned = True


class DeviceView:
    """Attribute names of CDSimulatorComponents.Device (:217-243)."""
    __slots__ = ("id", "isCompromised", "attacker_owned", "Known_to_attacker", "reachable_by_attacker",
                 "Not_yet_added", "busy_time", "workload", "device_type", "wtype", "anomaly_score",
                 "compromised_by", "version", "n_apps")

    def getId(self):
        return self.id

    def __repr__(self):
        return f"DeviceView(id={self.id}, comp={self.isCompromised}, owned={self.attacker_owned}, nya={self.Not_yet_added})"


class ExploitView:
    __slots__ = ("id", "index", "discovered")

    def getId(self):
        return self.id


class GraphView:
    """The cached adjacency (_outnbrs / _innbrs) with python-igraph's read accessors: the shared base CSR
    plus the edges evolve_network added to THIS env (extra-edge list), rows in neighbour-id order."""

    def __init__(self, topo, blocked_bits, extra=()):
        self._t = topo
        self._blocked = blocked_bits
        self._extra = sorted((int(u), int(v)) for (u, v, _b) in extra)
        self._extra_blocked = {(int(u), int(v)) for (u, v, b) in extra if b}

    def vcount(self):
        return self._t.M

    def ecount(self):
        return self._t.E + len(self._extra)

    def get_edgelist(self):
        src = np.repeat(np.arange(self._t.M), np.diff(self._t.out_ptr))
        return [(int(u), int(v)) for u, v in zip(src, self._t.out_col)] + list(self._extra)

    def neighbors(self, v, mode="all"):
        t = self._t
        out = [int(x) for x in t.out_col[t.out_ptr[v]:t.out_ptr[v + 1]]]
        inn = [int(x) for x in t.in_col[t.in_ptr[v]:t.in_ptr[v + 1]]]
        if self._extra:
            out = sorted(out + [b for (a, b) in self._extra if a == v])
            inn = sorted(inn + [a for (a, b) in self._extra if b == v])
        m = str(mode).lower()
        return out if m == "out" else inn if m == "in" else sorted(out + inn)

    def get_adjlist(self, mode="out"):
        return [self.neighbors(v, mode) for v in range(self._t.M)]

    def degree(self, v=None, mode="all"):
        if v is None:
            return [len(self.neighbors(i, mode)) for i in range(self._t.M)]
        if isinstance(v, (list, tuple, range)):
            return [len(self.neighbors(i, mode)) for i in v]
        return len(self.neighbors(v, mode))

    def get_eid(self, u, v, directed=True, error=True):
        """Edge id of (u -> v): base CSR slot, or E + position in the env's extra-edge list; -1 when absent."""
        t = self._t
        for k in range(int(t.out_ptr[u]), int(t.out_ptr[u + 1])):
            if int(t.out_col[k]) == int(v):
                return k
        if (int(u), int(v)) in self._extra:
            return t.E + self._extra.index((int(u), int(v)))
        if error:
            raise ValueError(f"no such edge ({u}, {v})")
        return -1

    def blocked_edges(self):
        src = np.repeat(np.arange(self._t.M), np.diff(self._t.out_ptr))
        return {(int(src[k]), int(self._t.out_col[k])) for k in np.nonzero(self._blocked)[0]} | set(self._extra_blocked)


class SubnetView:
    def __init__(self, net, graph):
        self.net = net
        self.graph = graph
        self.partitions = None

    def create_partitions(self, partition_size: int):
        """Subnet.create_partitions (CDSimulatorComponents.py:556-582): disjoint parts of about `partition_size` devices,
        stored in `.partitions` as lists of device ids (read by the hierarchical agents, hierarchical_br.py:142-147,
        :435-438).  The reference asks METIS (pymetis.part_graph) for nparts = ceil(n / partition_size) parts; METIS's
        exact cut is a property of that library build and is not pinned here (pymetis is not installable in this
        image): this is a deterministic balanced partition of the same graph with the same part count -- breadth-first
        order from the lowest unvisited id over the undirected adjacency, cut into nparts runs whose sizes differ by
        at most one -- so connected devices land together, like a graph partitioner's parts."""
        import math
        n = self.graph.vcount()
        if n == 0:
            raise ValueError("Cannot partition an empty graph")
        nparts = min(max(1, math.ceil(n / partition_size)), n)
        seen, order = [False] * n, []
        for root in range(n):
            if seen[root]:
                continue
            seen[root] = True
            queue = [root]
            while queue:
                v = queue.pop(0)
                order.append(v)
                for w in self.graph.neighbors(v, mode="all"):
                    if not seen[w]:
                        seen[w] = True
                        queue.append(w)
        q, r = divmod(n, nparts)
        parts, pos = [], 0
        for p in range(nparts):
            size = q + (1 if p < r else 0)
            parts.append(sorted(order[pos:pos + size]))
            pos += size
        self.partitions = parts


class LoggerView:
    def __init__(self, logs):
        self.logs = logs

    def get_logs(self):
        return self.logs


class DetectorView:
    def __init__(self, eflags):
        self.trained = bool(eflags & S.E_DET_TRAIN)
        self.random_detection = bool(eflags & S.E_DET_RANDOM)


class SimulatorView:
    """What `env.simulator` exposes to the reference's agents."""

    def __init__(self, view):
        from . import abi
        b, i = view._b, view._i
        st = {k: b.state[k][i].cpu().numpy() for k in ("live", "blocked", "ienv", "extra")}
        flags, busy, wl, cby = st["live"]
        topo = b.topo
        net = {}
        for d in range(topo.M):
            dv = DeviceView()
            f = int(flags[d])
            dv.id = d
            dv.isCompromised = bool(f & S.F_COMP)
            dv.attacker_owned = bool(f & S.F_OWNED)
            dv.Known_to_attacker = bool(f & S.F_KNOWN)
            dv.reachable_by_attacker = bool(f & S.F_REACH)
            dv.Not_yet_added = bool(f & S.F_NYA)
            dv.busy_time = int(busy[d])
            dv.device_type = "DomainController" if topo.dstatic[d] & S.D_DC else None
            dv.wtype = "server" if topo.dstatic[d] & S.D_SERVER else "client"
            dv.anomaly_score = float(topo.anomaly[d])
            dv.version = float(topo.version[d])
            dv.n_apps = int(topo.napps[d])
            dv.workload = WorkloadView(wl[d], f & S.F_WLADV, dv.wtype) if wl[d] > 0 else None
            dv.compromised_by = {e for e in range(topo.X) if (int(cby[d]) >> e) & 1}
            net[d] = dv
        bits = abi.unpack_blocked(st["blocked"].view(np.uint32)[None], topo.E)[0]
        n_extra = (int(st["ienv"][S.I_FLAGS]) & 0xFFFFFFFF) >> S.E_NX_SHIFT
        extra = abi.unpack_extra(st["extra"].view(np.uint32), n_extra, topo.max_extra) if n_extra else ()
        self.subnet = SubnetView(net, GraphView(topo, bits, extra))
        disc = int(st["ienv"][S.I_DISCOVERED])
        self.exploits = []
        for e in range(topo.X):
            ex = ExploitView()
            ex.id = e
            ex.index = e
            ex.discovered = bool((disc >> e) & 1)
            self.exploits.append(ex)
        self.logger = LoggerView(view._logs())
        self.detector = DetectorView(int(st["ienv"][S.I_FLAGS]))
        self.system_time = 0

    def getExploitsSize(self):
        return len(self.exploits)

    def getSubnetSize(self):
        return len(self.subnet.net)
