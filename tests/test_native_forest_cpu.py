"""The library's native estimator (cygym_fit_forests, csrc/cg_iforest.hpp: scikit-learn's IsolationForest(n_estimators=2,
max_samples=256).fit restated in C++, numpy's legacy MT19937 stream included) against
  (1) the forests the REFERENCE fitted -- every Detector.train event of every golden fixture: rows + Philox-addressed
      seed in, the reference's forest out, word for word.  Needs no scikit-learn at test time;
  (2) scikit-learn itself on random training sets (sizes around every branch of its row sampler: <= 256 rows, 257-258
      rows -> reservoir sampling, more -> permutation; constant and heavily duplicated features; several fits on one
      stream), when the installed release is the one restated."""
import numpy as np
import pytest

import golden_io as gio
from cygym_amd import detector as D
from cygym_amd import spec as S


def _events():
    out = []
    for name in gio.fixture_names():
        z = np.load(gio.os.path.join(gio.GOLDEN, name + ".npz"))
        if z["det_env"].size:
            out.append(name)
    return out


@pytest.mark.parametrize("name", _events())
def test_native_fit_reproduces_the_references_forests(name):
    fx = gio.Fixture(name)
    evs = [(t, ev) for t, lst in sorted(fx.det_events.items()) for ev in lst]
    assert evs
    rows = [ev["rows"] for _, ev in evs]
    if fx.cfg.turbo:    # (fixture rows are already the clipped / strided window the reference trained on)
        pass
    seeds = [D.fit_seed(fx.cfg.seed, fx.cfg.env_id_base + ev["env"], ev["rng_tick"]) for _, ev in evs]
    words, failed = D.fit_forests_native(rows, seeds, [ev["n_fits"] for _, ev in evs], threads=4)
    assert not failed.any()
    for (t, ev), w in zip(evs, words):
        np.testing.assert_array_equal(w, ev["forest"], err_msg=f"{name} tick {t} env {ev['env']} ({len(ev['rows'])} rows, {ev['n_fits']} fits)")


def test_native_fit_equals_scikit_learn_on_random_training_sets():
    sklearn = pytest.importorskip("sklearn")
    if sklearn.__version__ != D.NATIVE_SKLEARN:
        pytest.skip(f"the native estimator restates scikit-learn {D.NATIVE_SKLEARN}, here {sklearn.__version__}")
    rs = np.random.RandomState(7)
    rows_list, seeds, nfs = [], [], []
    for trial in range(260):
        n = int(rs.choice([1, 2, 3, 5, 17, 64, 200, 255, 256, 257, 258, 259, 300, 1000, 2000]))
        M = int(rs.choice([2, 4, 16, 64, 256, 2048]))
        rows = rs.randint(0, M, size=(n, 2))
        u = rs.rand()
        if u < 0.15:
            rows[:, int(rs.randint(2))] = rows[0, 0]            # a constant feature
        elif u < 0.22:
            rows[:] = rows[0]                                   # every row the same
        elif u < 0.45:
            rows = rows[rs.randint(0, max(1, n // 8), size=n)]  # heavy duplication
        rows_list.append(rows); seeds.append(int(rs.randint(1 << 32, dtype=np.uint64))); nfs.append(int(rs.choice([1, 1, 1, 2, 3])))
    words, failed = D.fit_forests_native(rows_list, seeds, nfs, threads=3)
    assert not failed.any()
    for j, (rows, seed, nf) in enumerate(zip(rows_list, seeds, nfs)):
        np.testing.assert_array_equal(words[j], D.fit_forest(rows, seed, nf), err_msg=f"case {j}: {len(rows)} rows, seed {seed}, {nf} fits")
    # the dispatcher: auto == native here, and both engines agree through fit_forests
    np.testing.assert_array_equal(D.fit_forests(rows_list[:20], seeds[:20], nfs[:20], engine="auto"),
                                  D.fit_forests(rows_list[:20], seeds[:20], nfs[:20], engine="sklearn"))


def test_native_fit_rejects_what_the_flat_layout_cannot_hold():
    """Device ids >= 4096 do not fit the 12-bit threshold field: reported per request, never written wrong."""
    rows = np.array([[5000, 1], [1, 6000], [7000, 3], [2, 2]] * 10)
    words, failed = D.fit_forests_native([rows, np.array([[1, 2], [3, 4], [5, 6]])], [1, 2])
    assert failed[0] and not failed[1]
    from cygym_amd import _lib
    lib = _lib.load()
    assert lib.cygym_fit_forests(None, None, None, None, None, 1, 1, None, None) < 0
    ptr = np.zeros(2, np.int64)     # an empty request
    tab, out = D.sstar_table(), np.zeros((1, S.FOREST_WORDS), np.uint32)
    sd = np.zeros(1, np.uint32)
    r = np.zeros((1, 2), np.uint16)
    assert lib.cygym_fit_forests(r.ctypes.data, ptr.ctypes.data, sd.ctypes.data, None, tab.ctypes.data, 1, 1, out.ctypes.data, None) < 0
