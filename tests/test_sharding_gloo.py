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
