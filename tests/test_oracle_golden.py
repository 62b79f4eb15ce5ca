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
    assert gio.check_oracle_against_fixture(gio.Fixture(name)) > 0


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
