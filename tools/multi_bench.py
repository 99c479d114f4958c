#!/usr/bin/env python3
"""Open-loop rate with SEVERAL steps per launch (auv_step_multi) against one step per launch, by chains and launch length:
python tools/multi_bench.py [workload polygons50|moving28] [envs]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gym_auv_amd.batched_env import BatchedAuvEnv  # noqa: E402
from gym_auv_amd.config import effective_reference_config  # noqa: E402
from gym_auv_amd.world import build_bank_parallel  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "polygons50"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
cfg = effective_reference_config(use_lidar=True)
gen, kw = ("polygon_world", dict(n_polygons=50)) if wl == "polygons50" else ("moving_obstacles_world", dict())
cache = "/tmp/multi_bench_%s_%d.npz" % (wl, n)
if os.path.exists(cache):
    z = np.load(cache)
    bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
else:
    bank = build_bank_parallel(gen, 1000 + np.arange(2 * n), procs=16, **kw)
    np.savez(cache, **bank)
slots = 64
ring = torch.rand((slots, n, 2), device=dev) * torch.tensor([2.0, 0.3], device=dev) - torch.tensor([1.0, 0.15], device=dev)
for k in (1, 2, 4):
    env = BatchedAuvEnv(cfg, bank, n, device=dev, auto_reset=True)
    env.reset()
    env.set_sub_batches(k, strict=True)
    steps = 1920
    for T, order in ((0, "-"), (8, "steps"), (64, "steps"), (8, "cohorts"), (16, "cohorts"), (64, "cohorts"), (64, "cohorts:6:16"), (64, "cohorts:20:40")):
        if T:
            o = order.split(":")
            env.set_multi_order(o[0], *([int(o[1]), int(o[2])] if len(o) > 1 else []))
        def run(m):
            if T == 0:
                for i in range(m):
                    env.step_pipelined(ring[i % slots])
            else:
                for i in range(0, m, T):
                    env.step_multi(ring, i % slots, T)
        run(steps // 2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps(dict(workload=wl, envs=n, chains=k, steps_per_launch=T or 1, order=order, rate_M=round(n * steps / dt / 1e6, 1), us_per_step=round(1e6 * dt / steps, 2),
                              health=env.health())), flush=True)
    env.close()
