"""Minimal `Box` space with the attributes gym 0.21's Box exposes to RL code (low, high,
shape, dtype, sample, contains).  gym itself is not a dependency of the accelerated path;
if `gym` is importable, `to_gym()` returns the real thing."""
import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if shape is None:
            shape = np.asarray(low).shape
        self.shape = tuple(shape)
        self.low = np.full(self.shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low, dtype=self.dtype)
        self.high = np.full(self.shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high, dtype=self.dtype)
        self._rng = np.random.RandomState()

    def seed(self, seed=None):
        self._rng = np.random.RandomState(seed)
        return [seed]

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def __repr__(self):
        return "Box(%s, %s, %s, %s)" % (self.low.min(), self.high.max(), self.shape, self.dtype)

    def to_gym(self):
        import gym.spaces  # optional
        return gym.spaces.Box(low=self.low, high=self.high, dtype=self.dtype.type)


class Dict:
    """gym.spaces.Dict stand-in: a mapping of named sub-spaces."""

    def __init__(self, spaces):
        self.spaces = dict(spaces)
        self.shape = None
        self.dtype = None

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()

    def contains(self, x):
        return set(x) == set(self.spaces) and all(self.spaces[k].contains(np.asarray(v, dtype=self.spaces[k].dtype)) for k, v in x.items())
