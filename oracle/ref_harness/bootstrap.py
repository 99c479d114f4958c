"""Oracle harness bootstrap (TEST TOOLING, build container only).

Makes the reference package at /root/reference importable in a container that lacks
gym / pygame / tkinter / shapely, so its *own* environment.py, vessel.py, sensor.py,
path.py, rewarder.py and obstacles.py run unmodified and emit golden vectors
(see make_golden.py).  Nothing from the reference is copied; nothing here ships in
the product path; nothing here is imported on the GPU box.

  gym      -> inert Env base + Box/Dict spaces + gym-0.21 style seeding (RandomState
              seeded from sha512 of the seed, as published in gym 0.21) + registry.
  pygame   -> inert module (renderer is imported by environment.py, never called).
  turtle   -> inert module (`from turtle import shape`, obstacles.py:4).
  shapely  -> ./shim/shapely (numpy restatement of the GEOS calls on the path).
"""
import hashlib
import importlib
import os
import struct
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))


class _Inert(types.ModuleType):
    """Module whose every attribute is a do-nothing callable/namespace."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _InertObj(name)


class _InertObj:
    def __init__(self, name="inert"):
        self._n = name

    def __call__(self, *a, **k):
        return _InertObj(self._n)

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _InertObj(name)

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


# ----------------------------------------------------------------------------- seeding
def _bigint_from_bytes(b):
    pad = 4 - len(b) % 4
    b = b + b"\0" * pad
    n = len(b) // 4
    vals = struct.unpack("{}I".format(n), b)
    return sum(v << (32 * i) for i, v in enumerate(vals))


def _hash_seed(seed, max_bytes=8):
    h = hashlib.sha512(str(seed).encode("utf8")).digest()
    return _bigint_from_bytes(h[:max_bytes])


def _int_list(big):
    if big == 0:
        return [0]
    out = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        out.append(mod)
    return out


def np_random(seed=None):
    """gym 0.21 `seeding.np_random`: RandomState seeded with ints derived from sha512(seed)."""
    if seed is None:
        seed = _bigint_from_bytes(os.urandom(8))
    seed = int(seed) % 2 ** 64
    rng = np.random.RandomState()
    rng.seed(_int_list(_hash_seed(seed)))
    return rng, seed


# ------------------------------------------------------------------------------ spaces
class Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)
        super().__init__(shape, dtype)
        self.low, self.high = low, high


class Dict(Space):
    def __init__(self, spaces):
        super().__init__(None, None)
        self.spaces = dict(spaces)

    def __getitem__(self, k):
        return self.spaces[k]


class Env:
    metadata = {}

    def seed(self, seed=None):
        return [seed]


_REGISTRY = {}


def register(id, entry_point=None, kwargs=None, **_):
    _REGISTRY[id] = (entry_point, dict(kwargs or {}))


def make(id, **kw):
    entry, kwargs = _REGISTRY[id]
    mod, cls = entry.split(":")
    kwargs = dict(kwargs)
    kwargs.update(kw)
    return getattr(importlib.import_module(mod), cls)(**kwargs)


def install():
    """Populate sys.modules / sys.path; idempotent. Returns the imported gym_auv module."""
    sys.dont_write_bytecode = True  # never write into /root/reference
    if "gym_auv" in sys.modules:
        return sys.modules["gym_auv"]

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    err = mod("gym.error", Error=type("Error", (Exception,), {}))
    spaces = mod("gym.spaces", Space=Space, Box=Box, Dict=Dict)
    seeding = mod("gym.utils.seeding", np_random=np_random)
    utils = mod("gym.utils", seeding=seeding)
    registration = mod("gym.envs.registration", register=register)
    envs = mod("gym.envs", registration=registration)
    gym = mod("gym", Env=Env, spaces=spaces, utils=utils, envs=envs, error=err, make=make)
    gym.__path__ = []  # mark as package
    for name in ("pygame", "pygame.freetype", "pygame.gfxdraw", "pygame.locals", "turtle"):
        sys.modules[name] = _Inert(name)
    sys.modules["pygame"].__path__ = []

    sys.path.insert(0, os.path.join(_HERE, "shim"))
    sys.path.insert(0, REFERENCE_ROOT)
    return importlib.import_module("gym_auv")


if __name__ == "__main__":
    g = install()
    print("imported", g.__file__, "scenarios:", sorted(g.SCENARIOS))
    print("effective dt:", g.DEFAULT_CONFIG.simulation.t_step_size,
          "min_goal_distance:", g.DEFAULT_CONFIG.episode.min_goal_distance)
