"""-m gpu: the PPO example drives the batched env with device tensors only, with the
reference's hyper-parameters (scripts/run.py:332-357), and it LEARNS: on PathFollowNoObstacles-v0 the policy
picks up speed and turns onto the path within a few dozen updates (profiles/r02/ppo_pathfollow_*_seed*.log hold
120-update runs of three seeds: surge 0.2 -> 0.47 m/s, |heading error| 0.85 -> 0.3 rad)."""
import math
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_ppo_colav_runs():
    import ppo
    hist = ppo.train(envs=1024, updates=6, rollout=16, log=lambda *_: None)
    assert len(hist) == 6
    assert all(math.isfinite(h[0]) and math.isfinite(h[1]) for h in hist)
    assert hist[-1][0] > hist[0][0] - 1.0        # the mean step reward does not collapse


def test_ppo_learns_path_following():
    import ppo
    hist = ppo.train(envs=2048, updates=60, rollout=32, regen=0, task="pathfollow", log=lambda *_: None)
    k = 6
    first, last = hist[:k], hist[-k:]
    mean = lambda rows, i: sum(r[i] for r in rows) / len(rows)   # noqa: E731
    assert mean(last, 3) > mean(first, 3) + 0.12, (mean(first, 3), mean(last, 3))     # surge speed: it learnt to use the thruster
    assert mean(last, 4) < mean(first, 4) - 0.15, (mean(first, 4), mean(last, 4))     # |heading error|: ... and the rudder
    assert mean(last, 0) > hist[0][0] + 0.3, (hist[0][0], mean(last, 0))              # the step reward against the untrained policy's


def test_ppo_rollout_as_one_captured_graph_per_step():
    """--graph-rollout: policy forward, sampling, the environment's step (ONE kernel launch, enqueued through the C
    ABI on torch's capture stream) and the value net replayed as one device graph per rollout step.  It runs, stays
    finite, and the policy still picks up speed on path following."""
    import ppo
    hist = ppo.train(envs=1024, updates=4, rollout=16, regen=0, log=lambda *_: None, graph_rollout=True)
    assert len(hist) == 4 and all(math.isfinite(h[0]) and math.isfinite(h[1]) for h in hist)
    hist = ppo.train(envs=2048, updates=40, rollout=32, regen=0, task="pathfollow", log=lambda *_: None, graph_rollout=True,
                     graph_update=True)      # (... and forward + backward + clipping + Adam of a minibatch as one graph too)
    mean = lambda rows, i: sum(r[i] for r in rows) / len(rows)   # noqa: E731
    assert mean(hist[-5:], 3) > mean(hist[:5], 3) + 0.04, (mean(hist[:5], 3), mean(hist[-5:], 3))    # surge speed
    assert mean(hist[-5:], 0) > hist[0][0] + 0.3, (hist[0][0], mean(hist[-5:], 0))                    # step reward
