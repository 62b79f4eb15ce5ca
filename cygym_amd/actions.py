"""numpy mirror of gen_actions_kernel (cygym_amd/csrc/cg_aux_kernels.hpp): the synthetic
action script of bench.py -- alternating defender / attacker
turns, defender type uniform over {1,4,5,6,7,8,9,11,12,13,2} on k ~ U[1, M/8]
distinct devices, attacker uniform over {1,2,3}, exploit uniform over X.
Used by tests (GPU script == this) and to feed the CPU baseline the same actions."""
from __future__ import annotations

import numpy as np

from . import rng as R
from . import spec as S

DEF_TYPES = np.array([1, 4, 5, 6, 7, 8, 9, 11, 12, 13, 2], np.int64)


def gen_actions_numpy(seed: int, env_id_base: int, n_envs: int, M: int, X: int, tick: int, max_devs: int):
    env = np.arange(n_envs, dtype=np.uint64) + np.uint64(env_id_base)
    k0, k1 = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    r = R.philox4x32_10_np(env, tick, S.SITE_ACTGEN, 0, k0, k1)
    q = R.philox4x32_10_np(env, tick, S.SITE_ACTGEN, 1, k0, k1)
    mulhi = lambda u, n: ((u.astype(np.uint64) * np.uint64(n)) >> np.uint64(32)).astype(np.int64)  # noqa: E731
    m = tick & 1
    act = dict(
        mode=np.full(n_envs, m, np.int32), n_groups=np.zeros(n_envs, np.int32),
        atype=np.zeros((n_envs, 1), np.int32), n_exploit=np.ones((n_envs, 1), np.int32),
        exploit=np.full((n_envs, 1, S.MAX_EXPLOITS), -1, np.int32), app=np.zeros((n_envs, 1), np.int32),
        dev_cnt=np.zeros((n_envs, 1), np.int32), dev_idx=np.zeros((n_envs, max_devs), np.int16))
    act["exploit"][:, 0, 0] = mulhi(r[1], max(1, X))
    act["app"][:, 0] = mulhi(r[2], 4)
    if m == S.MODE_DEFENDER:
        act["atype"][:, 0] = DEF_TYPES[mulhi(r[0], 11)]
        kmax = min(max(1, M // 8), max_devs)
        k = 1 + mulhi(r[3], kmax)
        a = mulhi(q[0], M)
        if M > 1 and (M & (M - 1)) == 0:
            stride = 2 * mulhi(q[1], M // 2) + 1
        else:
            stride = np.ones(n_envs, np.int64)
        j = np.arange(max_devs, dtype=np.int64)[None, :]
        dev = (a[:, None] + j * stride[:, None]) % M
        dev = np.where(j < k[:, None], dev, 0)
        act["dev_idx"][:] = dev.astype(np.int16)
        act["dev_cnt"][:, 0] = k
    else:
        act["atype"][:, 0] = 1 + mulhi(r[0], 3)
    return act
