"""Multi-GPU: independent envs sharded by env id, one process per GPU.

There is no collective on the step path: rank g owns the contiguous env-id range
[begin, end) and keys its Philox draws with the GLOBAL env id (EnvConfig.env_id_base =
begin), so results do not depend on the number of ranks.  The only communication is the
optional end-of-rollout gather of per-env returns / counters -- a few bytes per env,
latency-bound, so a direct all_gather (RCCL over xGMI with the "nccl" backend, or gloo
on CPU) rather than anything bucketed.
"""
from __future__ import annotations

import dataclasses

import torch
import torch.distributed as dist

from . import abi


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous, balanced split of env ids: the first (n_total % world) ranks get one extra env."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(int(n_total), int(world))
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def shard_config(cfg: abi.EnvConfig, n_total: int, rank: int, world: int):
    """(EnvConfig for this rank, number of local envs)."""
    begin, end = shard_range(n_total, rank, world)
    return dataclasses.replace(cfg, env_id_base=cfg.env_id_base + begin), end - begin


def gather_by_env(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """all_gather of a per-env tensor ([n_local, ...]) into global env order ([n_total, ...]).
    Shards may differ by one env, so they are padded to the largest and trimmed afterwards."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_range(n_total, r, world) for r in range(world)]
    n_max = max(e - b for b, e in sizes)
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[: e - b] for p, (b, e) in zip(parts, sizes)], dim=0)
