"""Env-local RNG with the stream the reference gets from gym 0.21's
`gym.utils.seeding.np_random` (environment.py:439-442; gym==0.21.0 is pinned in the
reference's requirements.txt:1 but absent from /root/reference): a NumPy RandomState seeded
with the 32-bit words of the first 8 bytes of sha512(str(seed))."""
import hashlib
import os
import struct
from typing import Optional, Tuple

import numpy as np


def _bigint_from_bytes(b: bytes) -> int:
    b = b + b"\0" * (4 - len(b) % 4)          # gym pads even when already aligned
    words = struct.unpack("{}I".format(len(b) // 4), b)
    return sum(v << (32 * i) for i, v in enumerate(words))


def hash_seed(seed: int, max_bytes: int = 8) -> int:
    return _bigint_from_bytes(hashlib.sha512(str(seed).encode("utf8")).digest()[:max_bytes])


def np_random(seed: Optional[int] = None) -> Tuple[np.random.RandomState, int]:
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError("Seed must be a non-negative integer or omitted, not {}".format(seed))
    if seed is None:
        seed = _bigint_from_bytes(os.urandom(8))
    seed = int(seed) % 2 ** 64
    big = hash_seed(seed)
    words = []
    while big > 0:
        big, mod = divmod(big, 2 ** 32)
        words.append(mod)
    rng = np.random.RandomState()
    rng.seed(words or [0])
    return rng, seed
