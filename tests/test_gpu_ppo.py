"""-m gpu: the PPO example drives the batched env with device tensors only."""
import math
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_ppo_runs_and_learns_something():
    import ppo
    hist = ppo.train(envs=1024, updates=6, rollout=16, log=lambda *_: None)
    assert len(hist) == 6
    assert all(math.isfinite(r) and math.isfinite(l) for r, l, _ in hist)
    assert hist[-1][0] > hist[0][0] - 1.0        # the mean step reward does not collapse
