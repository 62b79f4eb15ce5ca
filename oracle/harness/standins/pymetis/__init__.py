"""Stand-in for `pymetis` (absent).  Partitions are never read by step(); the
harness only needs part_graph() to return something of the right shape."""


def part_graph(nparts, adjacency=None, **kw):
    n = len(adjacency) if adjacency is not None else 0
    nparts = max(1, int(nparts))
    size = max(1, -(-n // nparts))
    membership = [min(nparts - 1, i // size) for i in range(n)]
    return 0, membership
