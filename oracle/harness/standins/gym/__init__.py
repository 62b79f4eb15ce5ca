"""Minimal stand-in for the third-party `gym` package (absent from this image).

Only what the reference environment touches: gym.Env, gym.spaces.Discrete/Box,
gym.utils.seeding.np_random.  Written for the oracle harness; contains no
reference code.  Discrete.sample() draws from Python's global `random`, which the
harness never relies on for parity.
"""
from . import spaces  # noqa: F401
from . import utils   # noqa: F401


class Env:
    metadata = {}
    observation_space = None
    action_space = None

    def reset(self, *a, **k):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def seed(self, seed=None):
        return [seed]
