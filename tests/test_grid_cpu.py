"""The grid consumer's host logic without a GPU: the same simulate_grid loop driven over the CPU oracle
(tests/grid_util.OracleGrid), against hand-computed schedules."""
import numpy as np
import pytest

from cygym_amd import abi
from cygym_amd import spec as S
from cygym_amd.rollout_grid import SequencePolicy, _action_at, baseline_schedule, can_train, simulate_grid
from cygym_amd.topology import make_topology
from grid_util import IntPolicy, OracleGrid

torch = pytest.importorskip("torch")


class Recorder(OracleGrid):
    """Keeps the action type every env played at every tick."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.played = []

    def step(self):
        self.played.append((self.act_np["mode"].copy(), self.act_np["atype"][:, 0].copy()))
        return super().step()


def _setup(M=16):
    topo, init, ck = make_topology(M, 2, seed=8, n_active=12)
    return topo, init, abi.EnvConfig(seed=8, **ck)


def test_fixed_sequences_follow_the_global_tick():
    """do_agent.py:237-238: `strat.actions[t % len(strat.actions)]` with the GLOBAL tick t.  A length-2 defender
    sequence therefore always plays actions[0] (its turns are the even ticks), a length-3 one plays 0, 2, 1, 0, ..."""
    topo, init, cfg = _setup()
    d2 = [(2, [0], [], 0), (11, [0], [1], 0)]
    d3 = [(2, [0], [], 0), (3, [0], [], 0), (11, [0], [1], 0)]
    a2 = [(1, [0], [], 0), (2, [0], [], 0)]
    a3 = [(1, [0], [], 0), (2, [0], [], 0), (3, [0], [], 0)]
    assert [_action_at(d3, t, "defender")[0] for t in (0, 2, 4, 6)] == [2, 11, 3, 2]
    og = Recorder(topo, cfg, 4, init, 1, 4)
    simulate_grid(og, [d2, d3], [a2, a3], 1, 12, randomize=False)
    at = np.stack([p[1] for p in og.played])             # [T, N]; cells: (d2,a2) (d2,a3) (d3,a2) (d3,a3)
    np.testing.assert_array_equal(at[0::2, 0], [2] * 6)                  # d2 on even ticks: always actions[0]
    np.testing.assert_array_equal(at[0::2, 2], [2, 11, 3, 2, 11, 3])     # d3: ticks 0, 2, 4, ... -> indices 0, 2, 1, ...
    np.testing.assert_array_equal(at[1::2, 0], [2] * 6)                  # a2 on odd ticks: always actions[1]
    np.testing.assert_array_equal(at[1::2, 1], [2, 1, 3, 2, 1, 3])       # a3: ticks 1, 3, 5, ... -> indices 1, 0, 2, ...
    modes = np.stack([p[0] for p in og.played])
    assert ((modes & 0xFF) == (np.arange(12) % 2)[:, None]).all()


def test_baseline_schedule_and_mode_words():
    D = [[(7, [0], [5, 6], 0)], "No Defense", "Preset"]
    A = ["No Attack", [(1, [0], [], 0)], "Preset"]
    sch = baseline_schedule(D, A, np.arange(9), 1, 4, 0)
    N, ND, PR, NA = (abi.BASELINES[k] for k in ("Nash", "No Defense", "Preset", "No Attack"))
    np.testing.assert_array_equal(sch[:, 0], [N, NA, NA, NA])       # scripted defender: frozen from the attacker's first turn on
    np.testing.assert_array_equal(sch[:, 1], [N, N, N, N])
    np.testing.assert_array_equal(sch[:, 2], [N, PR, PR, PR])
    np.testing.assert_array_equal(sch[:, 3], [ND, NA, ND, NA])
    np.testing.assert_array_equal(sch[:, 4], [ND, ND, ND, ND])
    np.testing.assert_array_equal(sch[:, 8], [PR, PR, PR, PR])
    topo, init, cfg = _setup()
    og = Recorder(topo, cfg, 9, init, 1, 4)
    simulate_grid(og, D, A, 1, 6, randomize=False)
    for t, (mode, _) in enumerate(og.played):
        want = baseline_schedule(D, A, np.arange(9), 1, 6, 0)[t]
        np.testing.assert_array_equal((mode >> S.MODE_BASELINE_SHIFT) & 7, want + 1, err_msg=f"tick {t}")
    # the oracle honours it: in cell 0 the defender's removal of devices 5, 6 happens at t = 0 only ...
    assert (og.ob.state["flags"][0, [5, 6]] & S.F_NYA).all()
    # ... and in cell 2 ("Preset" attacker probes at t = 1) every later defender action is a no-op as well
    assert og.ob.state["ienv"][2, S.I_LAST_ATYPE] in (2, 8)


def test_training_policies_need_a_detector_batch():
    topo, init, cfg = _setup()
    assert can_train(IntPolicy("defender", 16, [1, 10], 0)) and not can_train(IntPolicy("defender", 16, [1, 5], 0))
    assert can_train(lambda obs, t, M, L: None)                      # undeclared: assumed to
    assert can_train(SequencePolicy([(10, [0], [], 0)], "defender")) and not can_train(SequencePolicy("No Defense", "defender"))
    og = OracleGrid(topo, cfg, 2, init, 1, 4, detector=False)
    with pytest.raises(ValueError, match="detector=True"):
        simulate_grid(og, [[(10, [0], [], 0), (5, [0], [1], 0)]], [[(1, [0], [], 0)], "No Attack"], 1, 8)


def test_closed_loop_with_trainings_on_the_oracle():
    """Action 10 then scans through the loop's servicing path (status word -> service_detectors -> next tick), on the
    oracle: no env may end up unpinned, forests are installed, and skipping the servicing is caught."""
    pytest.importorskip("sklearn")
    topo, init, cfg = _setup()
    D = [[(10, [0], [], 0), (8, [0], [], 0), (5, [0], [1, 2, 3], 0)], IntPolicy("defender", 16, [10, 5, 5, 1], 3)]
    A = [[(1, [0], [], 0)]]
    og = OracleGrid(topo, cfg, 6, init, 1, 4, detector=True)
    simulate_grid(og, D, A, 3, 40, randomize=True)
    fl = og.ob.state["ienv"][:, S.I_FLAGS]
    assert og.fitted >= 6 and not (fl & (S.E_UNPINNED | S.E_DET_PENDING)).any() and (fl & S.E_DET_TRAIN)[:3].all()
    assert (og.ob.state["forest"][:3, 2] != 0).all()      # (cells 0-2 play the scripted defender)

    class Lazy(OracleGrid):
        def service_detectors(self):
            pass
    lazy = Lazy(topo, cfg, 6, init, 1, 4, detector=True)
    with pytest.raises(RuntimeError, match="without a current forest"):
        simulate_grid(lazy, D, A, 3, 40, randomize=True)
