"""Minimal stand-in for `python-igraph` (absent from this image).

Covers the calls the reference environment makes on its graph object (vertex /
edge attribute sequences, neighbors, get_adjlist, degree, get_eid, add/delete
edges, incident, Barabasi).  Written for the oracle harness; contains no
reference code.

Conventions taken from python-igraph's documented behaviour:
  * integer vertex arguments are vertex indices, strings are looked up by "name";
  * neighbors()/get_adjlist() return indices sorted ascending (multi-edges give
    repeated entries); mode "all" on a directed graph is the sorted merge of the
    out- and in-lists;
  * delete_edges() renumbers the remaining edges compactly, keeping their order.
The Barabasi generator is this module's own (seeded from Python's `random`): real
igraph's C generator cannot be reproduced, and the harness exports whatever graph
was built, so only determinism matters.
"""
from __future__ import annotations

import random as _random


def _mode(mode):
    m = str(mode).lower()
    if m in ("out", "1"):
        return "out"
    if m in ("in", "2"):
        return "in"
    return "all"


class Vertex:
    __slots__ = ("_g", "index")

    def __init__(self, g, index):
        self._g = g
        self.index = index

    def __getitem__(self, key):
        return self._g._vattr[key][self.index]

    def __setitem__(self, key, value):
        col = self._g._vattr.setdefault(key, [None] * self._g._n)
        col[self.index] = value

    def attributes(self):
        return {k: col[self.index] for k, col in self._g._vattr.items()}

    def get(self, key, default=None):
        col = self._g._vattr.get(key)
        if col is None:
            return default
        v = col[self.index]
        return default if v is None else v


class VertexSeq:
    def __init__(self, g):
        self._g = g

    def __len__(self):
        return self._g._n

    def __iter__(self):
        return (Vertex(self._g, i) for i in range(self._g._n))

    def __getitem__(self, key):
        if isinstance(key, str):
            return list(self._g._vattr[key])
        if isinstance(key, slice):
            return [Vertex(self._g, i) for i in range(*key.indices(self._g._n))]
        i = int(key)
        if i < 0 or i >= self._g._n:
            raise IndexError("vertex index out of range")
        return Vertex(self._g, i)

    def __setitem__(self, key, values):
        if not isinstance(key, str):
            raise TypeError("vertex sequences are assigned per attribute name")
        if isinstance(values, (list, tuple)):
            if len(values) != self._g._n:
                raise ValueError("attribute list length mismatch")
            self._g._vattr[key] = list(values)
        else:
            self._g._vattr[key] = [values] * self._g._n
        if key == "name":
            self._g._name_index = None

    def attributes(self):
        return list(self._g._vattr.keys())

    def find(self, *args, **kw):
        if "name" in kw:
            idx = self._g._lookup_name(kw["name"])
            if idx is None:
                raise ValueError(f"no such vertex: {kw['name']!r}")
            return Vertex(self._g, idx)
        if args:
            return self[args[0]]
        raise ValueError("unsupported find()")


class Edge:
    __slots__ = ("_g", "index")

    def __init__(self, g, index):
        self._g = g
        self.index = index

    @property
    def tuple(self):
        return self._g._edges[self.index]

    @property
    def source(self):
        return self._g._edges[self.index][0]

    @property
    def target(self):
        return self._g._edges[self.index][1]

    def __getitem__(self, key):
        return self._g._eattr[key][self.index]

    def __setitem__(self, key, value):
        col = self._g._eattr.setdefault(key, [None] * len(self._g._edges))
        col[self.index] = value

    def attributes(self):
        return {k: col[self.index] for k, col in self._g._eattr.items()}


class _EdgeSlice:
    def __init__(self, g, idxs):
        self._g = g
        self._idxs = list(idxs)

    def __setitem__(self, key, values):
        col = self._g._eattr.setdefault(key, [None] * len(self._g._edges))
        if isinstance(values, (list, tuple)):
            for i, v in zip(self._idxs, values):
                col[i] = v
        else:
            for i in self._idxs:
                col[i] = values

    def __getitem__(self, key):
        col = self._g._eattr[key]
        return [col[i] for i in self._idxs]


class EdgeSeq:
    def __init__(self, g):
        self._g = g

    def __len__(self):
        return len(self._g._edges)

    def __iter__(self):
        return (Edge(self._g, i) for i in range(len(self._g._edges)))

    def __getitem__(self, key):
        if isinstance(key, str):
            return list(self._g._eattr[key])
        if isinstance(key, slice):
            return _EdgeSlice(self._g, range(*key.indices(len(self._g._edges))))
        i = int(key)
        if i < 0 or i >= len(self._g._edges):
            raise IndexError("edge index out of range")
        return Edge(self._g, i)

    def __setitem__(self, key, values):
        if not isinstance(key, str):
            raise TypeError("edge sequences are assigned per attribute name")
        m = len(self._g._edges)
        if isinstance(values, (list, tuple)):
            if len(values) != m:
                raise ValueError("attribute list length mismatch")
            self._g._eattr[key] = list(values)
        else:
            self._g._eattr[key] = [values] * m

    def attributes(self):
        return list(self._g._eattr.keys())


class Graph:
    def __init__(self, n=0, edges=None, directed=False):
        self._n = 0
        self._directed = bool(directed)
        self._edges = []          # list of (u, v)
        self._vattr = {}
        self._eattr = {}
        self._out = None          # per-vertex sorted [(nbr, eid)]
        self._in = None
        self._name_index = None
        if n:
            self.add_vertices(int(n))
        if edges:
            self.add_edges(edges)

    # ---- construction ----
    @classmethod
    def Barabasi(cls, n, m=1, directed=False, **kw):
        g = cls(directed=directed)
        g.add_vertices(int(n))
        weight = [1] * n          # in-degree + 1
        edges = []
        for v in range(1, n):
            k = min(int(m), v)
            chosen = set()
            total = sum(weight[:v])
            guard = 0
            while len(chosen) < k and guard < 10000:
                guard += 1
                r = _random.random() * total
                acc = 0.0
                pick = v - 1
                for u in range(v):
                    acc += weight[u]
                    if r < acc:
                        pick = u
                        break
                chosen.add(pick)
            for u in sorted(chosen):
                edges.append((v, u))
                weight[u] += 1
        g.add_edges(edges)
        return g

    def is_directed(self):
        return self._directed

    def vcount(self):
        return self._n

    def ecount(self):
        return len(self._edges)

    @property
    def vs(self):
        return VertexSeq(self)

    @property
    def es(self):
        return EdgeSeq(self)

    def add_vertices(self, n):
        if isinstance(n, int):
            k = n
            names = None
        else:
            names = list(n)
            k = len(names)
        for col in self._vattr.values():
            col.extend([None] * k)
        if names is not None:
            col = self._vattr.setdefault("name", [None] * self._n)
            if len(col) < self._n + k:
                col.extend([None] * (self._n + k - len(col)))
            for i, nm in enumerate(names):
                col[self._n + i] = nm
            self._name_index = None
        self._n += k
        self._out = self._in = None

    def add_vertex(self, name=None, **kw):
        self.add_vertices(1 if name is None else [name])
        for k, v in kw.items():
            Vertex(self, self._n - 1)[k] = v

    def _lookup_name(self, name):
        if self._name_index is None:
            idx = {}
            for i, nm in enumerate(self._vattr.get("name", [])):
                if nm is not None and nm not in idx:
                    idx[nm] = i
            self._name_index = idx
        return self._name_index.get(name)

    def _vid(self, v):
        if isinstance(v, Vertex):
            return v.index
        if isinstance(v, str):
            idx = self._lookup_name(v)
            if idx is None:
                raise ValueError(f"no such vertex: {v!r}")
            return idx
        i = int(v)
        if i < 0 or i >= self._n:
            raise ValueError(f"vertex index out of range: {i}")
        return i

    def add_edge(self, source, target, **kw):
        self.add_edges([(source, target)])
        for k, v in kw.items():
            Edge(self, len(self._edges) - 1)[k] = v

    def add_edges(self, es):
        new = [(self._vid(u), self._vid(v)) for (u, v) in es]
        self._edges.extend(new)
        for col in self._eattr.values():
            col.extend([None] * len(new))
        self._out = self._in = None

    def delete_edges(self, es):
        if isinstance(es, int):
            es = [es]
        kill = {int(e.index if isinstance(e, Edge) else e) for e in es}
        if not kill:
            return
        keep = [i for i in range(len(self._edges)) if i not in kill]
        self._edges = [self._edges[i] for i in keep]
        for k, col in self._eattr.items():
            self._eattr[k] = [col[i] for i in keep]
        self._out = self._in = None

    # ---- adjacency ----
    def _build(self):
        if self._out is not None:
            return
        out = [[] for _ in range(self._n)]
        inn = [[] for _ in range(self._n)]
        for eid, (u, v) in enumerate(self._edges):
            out[u].append((v, eid))
            inn[v].append((u, eid))
            if not self._directed:
                out[v].append((u, eid))
                inn[u].append((v, eid))
        for lst in out:
            lst.sort()
        for lst in inn:
            lst.sort()
        self._out, self._in = out, inn

    def neighbors(self, vertex, mode="all"):
        self._build()
        i = self._vid(vertex)
        m = _mode(mode)
        if not self._directed:
            return [n for n, _ in self._out[i]]
        if m == "out":
            return [n for n, _ in self._out[i]]
        if m == "in":
            return [n for n, _ in self._in[i]]
        return sorted([n for n, _ in self._out[i]] + [n for n, _ in self._in[i]])

    def successors(self, vertex):
        return self.neighbors(vertex, mode="out")

    def predecessors(self, vertex):
        return self.neighbors(vertex, mode="in")

    def get_adjlist(self, mode="out"):
        return [self.neighbors(i, mode=mode) for i in range(self._n)]

    def incident(self, vertex, mode="out"):
        self._build()
        i = self._vid(vertex)
        m = _mode(mode)
        if not self._directed or m == "all":
            return sorted({e for _, e in self._out[i]} | {e for _, e in self._in[i]})
        if m == "out":
            return [e for _, e in self._out[i]]
        return [e for _, e in self._in[i]]

    def degree(self, vertices=None, mode="all", loops=True):
        self._build()
        m = _mode(mode)

        def one(v):
            i = self._vid(v)
            if not self._directed:
                return len(self._out[i])
            if m == "out":
                return len(self._out[i])
            if m == "in":
                return len(self._in[i])
            return len(self._out[i]) + len(self._in[i])

        if vertices is None:
            return [one(i) for i in range(self._n)]
        if isinstance(vertices, (list, tuple, set, range)):
            return [one(v) for v in vertices]
        return one(vertices)

    def indegree(self, vertices=None):
        return self.degree(vertices, mode="in")

    def outdegree(self, vertices=None):
        return self.degree(vertices, mode="out")

    def get_eid(self, v1, v2, directed=True, error=True):
        self._build()
        try:
            a, b = self._vid(v1), self._vid(v2)
        except ValueError:
            if error:
                raise
            return -1
        best = -1
        for n, e in self._out[a]:
            if n == b and (best < 0 or e < best):
                best = e
        if best < 0 and (not directed or not self._directed):
            for n, e in self._out[b]:
                if n == a and (best < 0 or e < best):
                    best = e
        if best < 0 and error:
            raise ValueError(f"no such edge: {a} -> {b}")
        return best

    def are_connected(self, v1, v2):
        return self.get_eid(v1, v2, error=False) >= 0

    def get_edgelist(self):
        return list(self._edges)

    def copy(self):
        import copy as _copy
        return _copy.deepcopy(self)


__all__ = ["Graph", "Vertex", "Edge", "VertexSeq", "EdgeSeq"]
