"""Pin the CPU oracle (oracle/cygym_oracle.c) against outputs of the reference itself.

The fixtures under tests/golden/ were produced by oracle/harness/make_golden.py, which
runs the unmodified reference environment with its RNG call sites fed from the build's
Philox stream.  Integer masks / counters must match bit-exactly, rewards within 1e-9.
"""
import numpy as np
import pytest

import golden_io as gio
from cygym_amd import spec as S
from oracle import driver as od

NAMES = gio.fixture_names()


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name):
    fx = gio.Fixture(name)
    ob = od.OracleBatch(fx.topo, fx.cfg, fx.N)
    ob.load_state(fx.init)
    act = od.alloc_actions(fx.N, fx.G, fx.L)
    alive = np.ones(fx.N, bool)   # parity is defined while the topology is unchanged
    checked = 0
    for t in range(fx.T):
        fx.actions(t, act, flags=ob.state["flags"])
        obs, raw, shaped, done = ob.step(act)
        same = fx.exp["topo_same"][:, t].astype(bool)
        # where the reference ADDED edges (evolve star / PA), the build must have flagged it
        ovf = (ob.state["ienv"][:, S.I_FLAGS] & S.E_TOPO_OVF) != 0
        if fx.follows_topology():   # the added edges are part of the compared state: parity never ends
            assert not ovf.any(), f"{name} t={t}: extra-edge list overflowed"
        else:
            assert np.array_equal(ovf[alive], ~same[alive]), f"{name} t={t}: TOPO_OVF {ovf} vs topo_same {same}"
            alive &= same
        if not alive.any():
            break
        exp = fx.expected_state(t)
        sel = np.where(alive)[0]
        got = {k: v[sel] for k, v in ob.state.items()}
        bad = gio.compare_state(got, {k: v[sel] for k, v in exp.items()}, f"{name} t={t}")
        assert not bad, "\n".join(bad[:8])
        np.testing.assert_array_equal(obs[sel], fx.exp["obs"][sel, t], err_msg=f"{name} obs t={t}")
        np.testing.assert_allclose(raw[sel], fx.exp["raw"][sel, t], rtol=0, atol=1e-9, err_msg=f"{name} raw t={t}")
        np.testing.assert_allclose(shaped[sel], fx.exp["shaped"][sel, t], rtol=0, atol=1e-9, err_msg=f"{name} shaped t={t}")
        np.testing.assert_array_equal(done[sel], fx.exp["done"][sel, t], err_msg=f"{name} done t={t}")
        np.testing.assert_array_equal(ob.observe(1)[sel], fx.exp["obs_def"][sel, t], err_msg=f"{name} obs_def t={t}")
        np.testing.assert_array_equal(ob.observe(2)[sel], fx.exp["obs_att"][sel, t], err_msg=f"{name} obs_att t={t}")
        checked += 1
    assert checked > 0


def test_oracle_randomize_matches_reference():
    """randomize_compromise_and_ownership (volt_typhoon_env.py:330-383): pre -> post state."""
    fx = gio.Fixture("s16_randomize")
    ob = od.OracleBatch(fx.topo, fx.cfg, fx.N)
    ob.load_state(fx.pre)
    ob.randomize()
    exp = {k: (v if k != "blocked" else None) for k, v in fx.init.items()}
    for k in ("flags", "busy", "wl", "comp_by"):
        np.testing.assert_array_equal(ob.state[k], fx.init[k].astype(ob.state[k].dtype), err_msg=k)
    from cygym_amd import spec as S
    np.testing.assert_array_equal(ob.state["ienv"][:, S.I_RNG_TICK], fx.init["ienv"][:, S.I_RNG_TICK])
    # the reshuffle must actually have moved ownership somewhere
    assert (fx.pre["flags"] != fx.init["flags"]).any()
