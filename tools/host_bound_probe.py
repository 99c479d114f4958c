#!/usr/bin/env python3
"""Is the open-loop chain loop host-bound?  Time to ENQUEUE n steps (the Python loop returning) against the time until the GPU
has finished them, for 3 / 4 chains, the bank cycling or a fresh world per reset (whose tick costs the host a graph launch)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_auv_amd.batched_env import BatchedAuvEnv  # noqa: E402
from gym_auv_amd.config import effective_reference_config  # noqa: E402
from gym_auv_amd.devgen import FreshWorlds, GeneratedWorlds  # noqa: E402

dev = torch.device("cuda:0")
cfg = effective_reference_config(use_lidar=True)
n = 4096
pool = torch.rand((16, n, 2), device=dev) * torch.tensor([2.0, 0.3], device=dev) - torch.tensor([1.0, 0.15], device=dev)
for name, worlds, k in (("cycling_sub4", GeneratedWorlds(2 * n, seed=1), 4), ("cycling_sub3", GeneratedWorlds(2 * n, seed=1), 3),
                        ("fresh_sub3_p16", FreshWorlds(seed=1), 3), ("fresh_sub3_p64_b256", FreshWorlds(seed=1, period=64, batch_cap=256), 3)):
    env = BatchedAuvEnv(cfg, worlds, n, device=dev, auto_reset=True)
    env.reset()
    env.set_sub_batches(k, strict=True)
    for i in range(3000):
        env.step_pipelined(pool[i % 16])
    torch.cuda.synchronize()
    steps = 2000
    t0 = time.perf_counter()
    for i in range(steps):
        env.step_pipelined(pool[i % 16])
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    out = dict(case=name, host_us_per_step=round(1e6 * t_host / steps, 2), total_us_per_step=round(1e6 * t_all / steps, 2), rate_M=round(n * steps / t_all / 1e6, 1))
    if env._fresh is not None:
        out["fresh"] = env.fresh_stats()
    print(json.dumps(out), flush=True)
    env.close()
