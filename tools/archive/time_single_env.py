"""Step latency of the single-environment gym.Env adapter (BASELINE configs[0]: 1 env, MovingObstacles, 180 sensors):
what SB-style code that steps ONE environment from Python sees, against the reference's 5.06 ms per step."""
import sys, os, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gym_auv_amd.config import effective_reference_config
from gym_auv_amd.env import make
env = make("MovingObstaclesNoRules-v0", env_config=effective_reference_config(use_lidar=True))
env.seed(0)
env.reset()
a = np.array([0.8, 0.05])
for _ in range(200):
    env.step(a)
n, t0, resets = 3000, time.perf_counter(), 0
for _ in range(n):
    _, _, done, _ = env.step(a)
    if done:
        env.reset(); resets += 1
dt = time.perf_counter() - t0
print("AuvEnv.step: %.1f us per step (%d steps, %d resets incl. world generation) = %.0f steps/s; reference: 5057 us" % (1e6 * dt / n, n, resets, n / dt))
