#!/usr/bin/env python3
"""Does the cohort-pipelined workgroup order pay for a launch of ONE step (what a closed-loop caller issues)?  k_step_roles (role-major:
all dynamics, all sweeps, all searches, all finish workgroups) against k_step_multi with n_steps = 1 in the cohort order, by lead / lag,
back to back on one stream and with a host synchronisation after every launch.  python tools/single_step_order_bench.py [envs]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import glob  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

from gym_auv_amd.batched_env import BatchedAuvEnv  # noqa: E402
from gym_auv_amd.config import effective_reference_config  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda:0")
cfg = effective_reference_config(use_lidar=True)
files = sorted(glob.glob("/tmp/bank.polygons50.0.%d.*.npz" % n))
if not files:
    raise SystemExit("run `python bench.py --envs %d --bank-cache /tmp/bank --steps 20` first (it leaves the bank)" % n)
z = np.load(files[0])
bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
slots = 64
ring = (torch.rand((slots, n, 2), device=dev, dtype=torch.float64) * torch.tensor([2.0, 0.3], device=dev, dtype=torch.float64)
        - torch.tensor([1.0, 0.15], device=dev, dtype=torch.float64)).contiguous()
for k in (1, 4):
    env = BatchedAuvEnv(cfg, bank, n, device=dev, auto_reset=True)
    env.reset()
    env.set_sub_batches(k, strict=(k > 1))
    for order in ("roles", "cohorts:16:30", "cohorts:8:16", "cohorts:4:10", "cohorts:2:6", "cohorts:1:3"):
        if order != "roles":
            o = order.split(":")
            env.set_multi_order("cohorts", int(o[1]), int(o[2]))
        for sync in (False, True):
            def run(m):
                for i in range(m):
                    if order == "roles":
                        env.step_pipelined(ring[i % slots])
                    else:
                        env.step_multi(ring, i % slots, 1)
                    if sync:
                        torch.cuda.synchronize()
            steps = 600
            run(steps // 3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps(dict(envs=n, chains=k, order=order, host_sync_every_step=sync, rate_M=round(n * steps / dt / 1e6, 1),
                                  us_per_step=round(1e6 * dt / steps, 2))), flush=True)
    env.close()
