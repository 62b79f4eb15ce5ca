"""Counter-based Philox4x32-10 and the draw conventions of the batched tick.

This is the host-side (pure Python / numpy) statement of the RNG contract in
include/cygym_spec.h.  The HIP kernels and the C oracle implement the same
functions; the oracle harness feeds the reference's `random.*` / `numpy.random.*`
call sites ("RNG sites") from these functions so that the
reference and the kernels consume identical draws.

A draw is addressed by (seed, global env id, rng tick, site, a, b) -- never by
"how many draws came before" -- so lanes can evaluate draws independently.
"""
from __future__ import annotations

import math
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Scalar Philox4x32-10. Returns a 4-tuple of python ints."""
    c0 &= MASK; c1 &= MASK; c2 &= MASK; c3 &= MASK; k0 &= MASK; k1 &= MASK
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, \
                         ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
        k0 = (k0 + W0) & MASK
        k1 = (k1 + W1) & MASK
    return c0, c1, c2, c3


def philox4x32_10_np(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox (numpy uint64 arithmetic); inputs broadcast."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & MASK for x in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint64(k0 & MASK); k1 = np.uint64(k1 & MASK)
    m = np.uint64(MASK); s = np.uint64(32)
    for _ in range(10):
        p0 = np.uint64(M0) * c0
        p1 = np.uint64(M1) * c2
        c0, c1, c2, c3 = ((p1 >> s) ^ c1 ^ k0) & m, p1 & m, ((p0 >> s) ^ c3 ^ k1) & m, p0 & m
        k0 = (k0 + np.uint64(W0)) & m
        k1 = (k1 + np.uint64(W1)) & m
    return c0, c1, c2, c3


def draw(seed: int, env: int, tick: int, site: int, a: int = 0, b: int = 0) -> int:
    """One 32-bit draw (word 0) for (env, tick, site, a, b)."""
    return philox4x32_10(env, tick, site, (a & 0xFFFF) | ((b & 0xFFFF) << 16),
                         seed & MASK, (seed >> 32) & MASK)[0]


def draw_np(seed: int, env, tick, site, a=0, b=0) -> np.ndarray:
    a = np.asarray(a, dtype=np.uint64); b = np.asarray(b, dtype=np.uint64)
    c3 = (a & np.uint64(0xFFFF)) | ((b & np.uint64(0xFFFF)) << np.uint64(16))
    return philox4x32_10_np(env, tick, site, c3, seed & MASK, (seed >> 32) & MASK)[0]


def index(u: int, n: int) -> int:
    """Index in [0, n) by multiply-high."""
    return (u * n) >> 32


def randint(u: int, lo: int, hi: int) -> int:
    """random.randint(lo, hi) (inclusive)."""
    return lo + index(u, hi - lo + 1)


def bernoulli_threshold(p: float) -> int:
    """thr such that `random.random() < p`  <=>  u32 < thr (u32/2^32 as uniform)."""
    if p <= 0.0:
        return 0
    if p >= 1.0:
        return 1 << 32
    return math.ceil(p * 4294967296.0)


def cdf_lookup(u: int, thr) -> int:
    return sum(1 for t in thr if u >= t)


def poisson_table(lam: float, n: int = 16):
    """u64 thresholds ceil(cdf_k * 2^32), k = 0..n-1, for np.random.poisson(lam)
    (inverse-CDF on one uniform).  Built once on the host so that CPU and GPU
    never evaluate exp() themselves."""
    out = []
    if lam <= 0.0:
        return [1 << 32] * n
    p = math.exp(-lam)
    cdf = p
    for k in range(n):
        out.append(min(1 << 32, math.ceil(cdf * 4294967296.0)))
        p = p * lam / (k + 1)
        cdf += p
    # the last threshold closes the table: anything beyond maps to n-1
    out[-1] = 1 << 32
    return out


def triangular_ceil_table(mode: float, high: float, n: int = 8):
    """Thresholds for int(ceil(np.random.triangular(0, mode, high))) in 1..ceil(high).
    value = 1 + #{j : u >= thr[j]};  thr[j] = ceil(CDF(j+1) * 2^32), padded with 2^32."""
    out = []
    top = int(math.ceil(high))
    for x in range(1, top):
        if x <= mode:
            c = (x * x) / (high * mode)
        else:
            c = 1.0 - ((high - x) ** 2) / (high * (high - mode))
        out.append(min(1 << 32, math.ceil(c * 4294967296.0)))
    while len(out) < n:
        out.append(1 << 32)
    return out[:n]
