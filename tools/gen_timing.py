"""Time on-device world generation (auv_generate_worlds) against the host builder.
usage: python tools/gen_timing.py [n_worlds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gym_auv_amd import devgen
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config
from gym_auv_amd.world import build_world

W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = effective_reference_config(use_lidar=True)
spec = devgen.GeneratedWorlds(W, seed=1)
t0 = time.time()
env = BatchedAuvEnv(cfg, spec, W, device="cuda:0")
torch.cuda.synchronize()
t1 = time.time()
env.generate(devgen.GeneratedWorlds(W, seed=2))
torch.cuda.synchronize()
t2 = time.time()
draws = devgen.sample_draws(8, 17, 11, seed=1).numpy()
t3 = time.time()
for r in draws:
    build_world(devgen.world_from_draws(r))
t4 = time.time()
print("device: first generate (alloc + %d worlds + reset rows) %.3f s, regenerate in place %.3f s (%.1f us/world)"
      % (W, t1 - t0, t2 - t1, (t2 - t1) / W * 1e6))
print("host builder: %.3f s/world on one core" % ((t4 - t3) / 8))
obs, rew, done, _ = env.step(torch.zeros((W, 2), device="cuda:0"))
print("step ok", bool(torch.isfinite(obs).all()))
