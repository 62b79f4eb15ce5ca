"""Host side of the trained detector (CDSimulator.py:681-723).

The reference's `Detector` wraps scikit-learn's `IsolationForest(n_estimators=2, max_samples=256)`:
`train(logs)` fits it on the `[from_device, to_device]` pairs of the last <= 2000 comm-log entries
(volt_typhoon_env.py:955-961), `batch_predict(points)` labels a point "A" when `model.predict == -1`.

In this build the split is:
  * **train** is a host callback.  The tick of defender action 10 only records the request (forest header +
    CG_E_DET_PENDING, cygym_spec.h); `fit_forest` below runs the very estimator the reference runs -- scikit-learn
    is the reference's own third-party dependency, not something this repo restates -- on the env's history
    ring, and `flatten_forest` turns the two fitted trees into the flat u32 layout the tick kernel walks.
  * **batch_predict** runs on the device (cg_defender.hpp) / in the oracle (cygym_oracle.c): a walk over the flat
    trees, one f64 add and one compare per point -- no pow, no log: the leaf term `apl[n]` is a 257-entry table
    computed here with numpy (`apl_table`), and the decision threshold is folded into one f64 `S*` per forest
    (`score_threshold`), found by bisection over the same numpy expressions scikit-learn evaluates.

`fit_forest` seeds the numpy stream the fit draws from with the Philox draw addressed (env, tick, CG_SITE_DET_FIT):
the reference draws from the process-global numpy stream there (IsolationForest(random_state=None)), which the
oracle harness seeds with the same draw before the reference's step -- so a fixture's forest can be refitted
bit for bit by this module (same scikit-learn version).
"""
from __future__ import annotations

import numpy as np

from . import rng as R
from . import spec as S


def apl_table() -> np.ndarray:
    """sklearn.ensemble._iforest._average_path_length(n) for n = 0..256, f64 [257].
    Same expression, evaluated by numpy (the table travels to device / oracle as data)."""
    n = np.arange(S.DET_APL_N, dtype=np.float64)
    out = np.zeros(S.DET_APL_N, np.float64)
    out[2] = 1.0
    m = n > 2
    out[m] = 2.0 * (np.log(n[m] - 1.0) + np.euler_gamma) - 2.0 * (n[m] - 1.0) / n[m]
    return out


def _is_anomaly(s: float, denominator: float, offset: float) -> bool:
    """IsolationForest.predict for a point whose summed depth is `s` (sklearn _compute_score_samples,
    score_samples, decision_function, predict -- same numpy operations, on arrays like sklearn does)."""
    depths = np.array([s], dtype=np.float64)
    den = np.float64(denominator)
    scores = 2 ** (-np.divide(depths, den, out=np.ones_like(depths), where=den != 0))
    return bool(((-scores) - offset < 0)[0])


def score_threshold(max_samples: int, n_estimators: int = 2, offset: float = -0.5) -> float:
    """S*: the smallest f64 summed depth that is NOT an anomaly (anomaly <=> depths < S*).
    Bisection over the (monotone) bit patterns of non-negative doubles."""
    apl = apl_table()
    denominator = n_estimators * float(apl[min(int(max_samples), S.DET_APL_N - 1)])
    if denominator == 0.0:          # a single training sample: every score is 0.5 -> nothing is an anomaly
        return 0.0
    lo = np.array([0.0]).view(np.uint64)[0]          # anomaly(lo) may be True
    hi = np.array([1024.0]).view(np.uint64)[0]       # never an anomaly
    f = lambda bits: _is_anomaly(float(np.array([bits], np.uint64).view(np.float64)[0]), denominator, offset)  # noqa: E731
    if not f(lo):
        return 0.0
    assert not f(hi)
    lo, hi = int(lo), int(hi)
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if f(mid):
            lo = mid
        else:
            hi = mid
    return float(np.array([hi], np.uint64).view(np.float64)[0])


_SSTAR_CACHE: dict = {}


def flatten_forest(model) -> np.ndarray:
    """A fitted sklearn IsolationForest -> u32 [FOREST_WORDS] (layout: cygym_spec.h).  Header words 3..5 (the
    request / answer ticks) are left 0 for the caller."""
    ests = model.estimators_
    if len(ests) != S.FOREST_TREES:
        raise ValueError(f"expected {S.FOREST_TREES} trees, got {len(ests)}")
    if any(list(f) != [0, 1] for f in model.estimators_features_):
        raise ValueError("feature sub-sampling is not part of the reference's detector")
    words = np.zeros(S.FOREST_WORDS, np.uint32)
    key = (int(model.max_samples_), len(ests), float(model.offset_))
    if key not in _SSTAR_CACHE:      # (a 64-step bisection over numpy expressions: once per distinct training size)
        _SSTAR_CACHE[key] = score_threshold(*key)
    words[0:2] = np.array([_SSTAR_CACHE[key]], np.float64).view(np.uint32)
    counts = []
    for t, est in enumerate(ests):
        tr = est.tree_
        n = int(tr.node_count)
        if n > S.FOREST_NODES - 1:
            raise ValueError(f"tree {t} has {n} nodes")
        depth = np.asarray(tr.compute_node_depths(), np.int64)[:n]
        left = np.asarray(tr.children_left, np.int64)[:n]
        right = np.asarray(tr.children_right, np.int64)[:n]
        leaf = left == -1
        ns = np.asarray(tr.n_node_samples, np.int64)[:n]
        if np.any(leaf & ~((0 <= ns) & (ns < 512) & (1 <= depth) & (depth <= 15))):
            raise ValueError("leaf does not fit the node encoding")
        ft = np.asarray(tr.feature, np.int64)[:n]
        fl = np.floor(np.where(leaf, 0.0, np.asarray(tr.threshold, np.float64)[:n])).astype(np.int64)
        if np.any(~leaf & ~(((ft == 0) | (ft == 1)) & (0 <= fl) & (fl < 4096))):
            raise ValueError("internal node does not fit the node encoding")
        base = S.FOREST_HDR + t * S.FOREST_NODES
        words[base: base + n] = np.where(leaf, (1 << 31) | (depth << 9) | ns,
                                         (ft << 30) | (fl << 18) | (left << 9) | right).astype(np.uint32)
        counts.append(n)
    words[2] = counts[0] | (counts[1] << 16)
    words[7] = int(model.max_samples_)      # what the decision_function value of the slow scan path is scaled by
    return words


def predict_flat(words: np.ndarray, points, apl: np.ndarray | None = None) -> np.ndarray:
    """Numpy walk over a flattened forest: True where IsolationForest.predict would say -1.  The host-side twin
    of the device walk, used to check `flatten_forest` against sklearn's own predict."""
    apl = apl_table() if apl is None else apl
    words = np.asarray(words, np.uint32)
    sstar = float(words[0:2].view(np.float64)[0])
    out = np.zeros(len(points), bool)
    for p, (a, b) in enumerate(points):
        depths = 0.0
        for t in range(S.FOREST_TREES):
            base = S.FOREST_HDR + t * S.FOREST_NODES
            w = int(words[base])
            for _ in range(16):
                if w >> 31:
                    break
                x = b if (w >> 30) & 1 else a
                w = int(words[base + (((w >> 9) & 0x1FF) if x <= ((w >> 18) & 0xFFF) else (w & 0x1FF))])
            depths += (float((w >> 9) & 0xF) + float(apl[min(w & 0x1FF, S.DET_APL_N - 1)])) - 1.0
        out[p] = depths < sstar
    return out


def fit_seed(seed: int, env_id: int, tick: int) -> int:
    """The 32-bit seed of the numpy stream IsolationForest.fit draws from at (env, tick)."""
    return int(R.draw(seed, env_id, tick, S.SITE_DET_FIT, 0, 0))


def fit_forest(X, seed32: int, n_fits: int = 1) -> np.ndarray:
    """Detector.train(logs) (CDSimulator.py:688-695): fit the reference's estimator on the [from, to] pairs and
    flatten it.  `n_fits` > 1: the tick asked several times (action 10 in several groups of one step_grouped
    call); like the reference, fit that often on the same rows from ONE continuing numpy stream and keep the last.
    Raises when scikit-learn is missing -- there is no substitute for the reference's estimator."""
    try:
        from sklearn.ensemble import IsolationForest
    except ImportError as e:   # pragma: no cover
        raise RuntimeError("trained-detector mode needs scikit-learn (the reference's own dependency, "
                           "CDSimulator.py:683); install it or do not use defender action 10") from e
    X = np.asarray(X, dtype=np.int64).reshape(-1, 2)
    if len(X) == 0:
        raise ValueError("Detector.train on an empty log is the random-detection mode (CDSimulator.py:688-690)")
    model = IsolationForest(n_estimators=2, max_samples=256, n_jobs=1, random_state=np.random.RandomState(int(seed32)))
    rows = X     # (the reference passes a list of [from, to] lists: validated into the same float32 matrix)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")     # "max_samples (256) is greater than the total number of samples"
        for _ in range(max(1, int(n_fits))):
            model.fit(rows)
    return flatten_forest(model)


_SSTAR_TABLE = None
NATIVE_SKLEARN = "1.7.2"     # the scikit-learn release whose fit csrc/cg_iforest.hpp restates (tests pin it against that one)


def sstar_table() -> np.ndarray:
    """S* by max_samples_ (0..256), f64 [257]: what the native fit writes into header words 0, 1."""
    global _SSTAR_TABLE
    if _SSTAR_TABLE is None:
        _SSTAR_TABLE = np.array([0.0] + [score_threshold(m) for m in range(1, S.DET_APL_N)], np.float64)
    return _SSTAR_TABLE


def fit_forests_native(rows_list, seeds, n_fits=None, threads: int | None = None):
    """The batch of Detector.train calls on the library's native estimator (cygym_fit_forests: csrc/cg_iforest.hpp,
    multi-threaded, no Python per fit).  Returns (words u32 [n, FOREST_WORDS], failed bool [n]): `failed` marks requests
    whose forest does not fit the flat layout -- fit those with `fit_forest` (scikit-learn)."""
    import os
    from . import _lib
    lib = _lib.load()
    n = len(rows_list)
    out = np.zeros((n, S.FOREST_WORDS), np.uint32)
    failed = np.zeros(n, np.uint8)
    if n == 0:
        return out, failed.astype(bool)
    ptr = np.zeros(n + 1, np.int64)
    ptr[1:] = np.cumsum([len(r) for r in rows_list])
    rows = np.ascontiguousarray(np.concatenate([np.asarray(r).reshape(-1, 2) for r in rows_list]).astype(np.uint16))
    sd = np.ascontiguousarray(np.asarray(seeds, np.uint32))
    nf = np.ascontiguousarray(np.asarray([1] * n if n_fits is None else n_fits, np.int32))
    tab = sstar_table()
    threads = threads or min(32, os.cpu_count() or 1)
    rc = lib.cygym_fit_forests(rows.ctypes.data, ptr.ctypes.data, sd.ctypes.data, nf.ctypes.data, tab.ctypes.data, n, int(threads),
                               out.ctypes.data, failed.ctypes.data)
    _lib.check(rc if rc < 0 else 0, None, "cygym_fit_forests")
    return out, failed.astype(bool)


def native_fit_available() -> bool:
    """The native estimator restates one scikit-learn release: use it when that release (or none) is installed -- with
    another one present scikit-learn itself fits, so that forests stay what the reference's own environment produces."""
    try:
        import sklearn
    except ImportError:
        return True
    return sklearn.__version__ == NATIVE_SKLEARN


def fit_forests(rows_list, seeds, n_fits=None, workers: int | None = None, engine: str = "auto") -> np.ndarray:
    """`fit_forest` for a batch of requests -> u32 [n, FOREST_WORDS].  engine "native": the library's own restatement of
    the estimator (cygym_fit_forests, csrc/cg_iforest.hpp: multi-threaded C++, tens of microseconds per forest);
    "sklearn": scikit-learn itself on a thread pool (about 4 ms of Python per forest, GIL-bound); "auto": native when
    the installed scikit-learn is the release it restates (or scikit-learn is absent), else scikit-learn."""
    n = len(rows_list)
    n_fits = [1] * n if n_fits is None else list(n_fits)
    if engine not in ("auto", "native", "sklearn"):
        raise ValueError("engine must be 'auto', 'native' or 'sklearn'")
    if n and (engine == "native" or (engine == "auto" and native_fit_available())):
        out, failed = fit_forests_native(rows_list, seeds, n_fits)
        for j in np.nonzero(failed)[0]:       # (a forest outside the flat layout: let scikit-learn raise or fit it)
            out[j] = fit_forest(rows_list[j], seeds[j], n_fits[j])
        return out
    out = np.zeros((n, S.FOREST_WORDS), np.uint32)
    if n == 0:
        return out
    import os
    workers = workers or min(8, os.cpu_count() or 1, n)

    def one(j):
        out[j] = fit_forest(rows_list[j], seeds[j], n_fits[j])

    if workers <= 1 or n < 4:
        for j in range(n):
            one(j)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(workers) as pool:
            list(pool.map(one, range(n)))
    return out


def training_window(hist_row: np.ndarray, log_total: int, turbo: bool = False, turbo_max_logs: int = 256,
                    turbo_stride: int = 2) -> np.ndarray:
    """The rows Detector.train sees: the last <= TRAIN_WINDOW entries of the env's history ring
    ([HIST_RING][2] u16, entry i of the log lives at i % HIST_RING), oldest first; in turbo mode clipped to the last
    `turbo_max_logs` of them and down-sampled by `turbo_stride` (volt_typhoon_env.py:165-169)."""
    n = min(int(log_total), S.TRAIN_WINDOW)
    idx = (np.arange(int(log_total) - n, int(log_total)) % S.HIST_RING).astype(np.int64)
    rows = np.asarray(hist_row).reshape(S.HIST_RING, 2)[idx].astype(np.int64)
    if turbo:
        rows = rows[-int(turbo_max_logs):]
        rows = rows[:: max(1, int(turbo_stride))]
    return rows


def decision_value(words: np.ndarray, point, apl: np.ndarray | None = None) -> float:
    """sklearn IsolationForest.decision_function for one (from, to) point over a flattened forest:
    0.5 - 2 ** (-s / (2 * apl[max_samples_])) with s the summed leaf values (the host-side twin of the device's
    anomaly score on the slow scan path)."""
    apl = apl_table() if apl is None else apl
    words = np.asarray(words, np.uint32)
    a, b = point
    depths = 0.0
    for t in range(S.FOREST_TREES):
        base = S.FOREST_HDR + t * S.FOREST_NODES
        w = int(words[base])
        for _ in range(16):
            if w >> 31:
                break
            x = b if (w >> 30) & 1 else a
            w = int(words[base + (((w >> 9) & 0x1FF) if x <= ((w >> 18) & 0xFFF) else (w & 0x1FF))])
        depths += (float((w >> 9) & 0xF) + float(apl[min(w & 0x1FF, S.DET_APL_N - 1)])) - 1.0
    den = S.FOREST_TREES * float(apl[min(int(words[7]), S.DET_APL_N - 1)])
    return 0.5 - 2.0 ** (-(depths / den if den != 0.0 else 1.0))     # (sklearn divides with where=den != 0, out=ones)
