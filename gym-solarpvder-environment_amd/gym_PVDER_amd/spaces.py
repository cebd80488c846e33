"""Minimal Discrete / Box spaces used when neither ``gym`` nor ``gymnasium`` is installed.

The reference declares ``action_space = spaces.Discrete(5)`` and
``observation_space = spaces.Box(low=-10, high=10, shape=(11,), dtype=np.float32)``
(/root/reference/gym_PVDER/envs/PVDER_env.py:50-51).  When a gym package is importable its own
space classes are used instead (see :func:`make_spaces`), so wrappers that type-check keep working.
"""
from __future__ import annotations

import numpy as np


class Discrete:
    def __init__(self, n: int, seed=None):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64
        self._rng = np.random.default_rng(seed)

    def sample(self) -> int:
        return int(self._rng.integers(0, self.n))

    def contains(self, x) -> bool:
        if isinstance(x, (bool, np.bool_)):
            return False
        if isinstance(x, (int, np.integer)):
            return 0 <= int(x) < self.n
        if isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
            return 0 <= int(x) < self.n
        return False

    __contains__ = contains

    def seed(self, seed=None):
        self._rng = np.random.default_rng(seed)

    def __repr__(self):
        return "Discrete(%d)" % self.n


class Box:
    def __init__(self, low, high, shape, dtype=np.float32, seed=None):
        self.shape = tuple(shape)
        self.dtype = np.dtype(dtype)
        self.low = np.full(self.shape, low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype)
        self._rng = np.random.default_rng(seed)

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return bool(x.shape == self.shape and np.all(np.isfinite(x)) and np.all(x >= self.low)
                    and np.all(x <= self.high))

    __contains__ = contains

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)


def make_spaces():
    """(action_space, observation_space) with gym/gymnasium classes when available."""
    for mod in ("gym", "gymnasium"):
        try:
            sp = __import__(mod + ".spaces", fromlist=["spaces"])
            return sp.Discrete(5), sp.Box(low=-10, high=10, shape=(11,), dtype=np.float32)
        except Exception:  # package absent or broken: fall through to the built-in spaces
            continue
    return Discrete(5), Box(low=-10, high=10, shape=(11,), dtype=np.float32)
