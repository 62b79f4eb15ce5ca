"""N > 1 path on CPU: two gloo ranks each step their shard (with the CPU oracle as the
compute stand-in -- tests may use it) and gather per-env returns; the result must equal
one process stepping the whole batch.  Proves shard-independence of the RNG keying and
exercises the only collective of the design."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cygym_amd import abi, sharding
from cygym_amd.actions import gen_actions_numpy
from cygym_amd.topology import make_topology

N_TOTAL, M, TICKS, SEED = 37, 64, 40, 4      # odd batch: uneven shards


def _run_shard(cfg, n_local):
    from oracle import driver as od
    topo, init, _ = make_topology(M, 4, seed=SEED, n_active=56)
    ob = od.OracleBatch(topo, cfg, n_local)
    ob.load_state(init)
    ret = np.zeros(n_local)
    for t in range(TICKS):
        act = gen_actions_numpy(cfg.seed, cfg.env_id_base, n_local, M, topo.X, t, M // 8)
        _, raw, _, _ = ob.step(act)
        ret += raw
    return ret, ob.state["flags"].copy()


def _base_cfg():
    _, _, ck = make_topology(M, 4, seed=SEED, n_active=56)
    return abi.EnvConfig(seed=SEED, env_id_base=1000, **ck)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, n_local = sharding.shard_config(_base_cfg(), N_TOTAL, rank, world)
    ret, flags = _run_shard(cfg, n_local)
    g_ret = sharding.gather_by_env(torch.from_numpy(ret), N_TOTAL)
    g_flags = sharding.gather_by_env(torch.from_numpy(flags), N_TOTAL)
    if rank == 0:
        np.save(out + ".ret.npy", g_ret.numpy())
        np.save(out + ".flags.npy", g_flags.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    for n, w in [(37, 2), (131072, 8), (5, 8), (64, 3)]:
        spans = [sharding.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [e - b for b, e in spans]
        assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    ret_full, flags_full = _run_shard(_base_cfg(), N_TOTAL)
    np.testing.assert_allclose(np.load(out + ".ret.npy"), ret_full, rtol=0, atol=1e-9)
    np.testing.assert_array_equal(np.load(out + ".flags.npy"), flags_full)


# ---- the grid consumer sharded by cell: 4 ranks, `cell_offset` / `n_total`, gathered payoff matrices ----
GRID_D = ["No Defense", [(1, [0], [3, 9, 12], 0), (7, [0], [5], 0), (6, [0], [1, 2, 3, 4], 0)]]
GRID_A = ["No Attack", [(1, [0], [], 0)], [(2, [0], [], 0), (1, [1], [], 0)]]
GRID_MC, GRID_T = 5, 16                                   # 2 x 3 x 5 = 30 cells over 4 ranks: shards of 8, 8, 7, 7


def _grid_shard(rank, world, closed_loop):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from grid_util import OracleGrid
    from cygym_amd.rollout_grid import payoff_grid, simulate_grid
    topo, init, _ = make_topology(M, 4, seed=SEED, n_active=56)
    cells = len(GRID_D) * len(GRID_A) * GRID_MC
    begin, end = sharding.shard_range(cells, rank, world)
    cfg, n_local = sharding.shard_config(_base_cfg(), cells, rank, world)      # env_id_base = first cell of the shard
    og = OracleGrid(topo, cfg, n_local, init, 1, 8)
    fn = simulate_grid if closed_loop else payoff_grid
    return fn(og, GRID_D, GRID_A, GRID_MC, GRID_T, randomize=True, n_total=cells, cell_offset=begin)


def _grid_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    res = [_grid_shard(rank, world, cl) for cl in (False, True)]
    np.save(out + f".r{rank}.npy", np.stack([np.stack(r) for r in res]))       # [2 ways, (U_def, U_att), |D|, |A|]
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_four_rank_grid_shards_equal_single_process(tmp_path):
    """payoff_grid and simulate_grid with the cells sharded over 4 gloo ranks (`cell_offset`, per-rank env_id_base,
    gather_by_env before averaging): every rank ends with the full payoff matrices of the single-process run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "grid")
    mp.spawn(_grid_worker, args=(4, port, out), nprocs=4, join=True)
    full = np.stack([np.stack(_grid_shard(0, 1, cl)) for cl in (False, True)])
    np.testing.assert_allclose(full[0], full[1], rtol=0, atol=1e-9)           # open-loop script == closed-loop SequencePolicy
    for r in range(4):
        np.testing.assert_allclose(np.load(out + f".r{r}.npy"), full, rtol=0, atol=1e-9, err_msg=f"rank {r}")
