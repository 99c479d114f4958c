#!/usr/bin/env python3
"""Diagnostic (STAMPS build): which environments' sweeps end a chain's launch?  Sweep duration (state arrived -> rows
out) of the last step by group: environments that refreshed their nearby mask in that step (vessel step counter a
multiple of sensor_interval_load_obstacles) against the others, and against the number of nearby obstacles.
    AUV_HIP_LIB=gym_auv_amd/csrc_stamps/libauv_hip.so SUB=4 python tools/straggler_stats.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config

cfg = effective_reference_config(use_lidar=True)
n = 4096
z = np.load(os.environ.get("BANK") or sorted(__import__("glob").glob("/tmp/bank.polygons50.0.4096.4096.2*.npz"))[0])   # (bench.py --bank-cache /tmp/bank: name carries a source hash)
bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
env = BatchedAuvEnv(cfg, bank, n, auto_reset=True)
env.set_step_mode("one_launch")
sub = int(os.environ.get("SUB", "4"))
env.set_sub_batches(sub)
env.reset()
g = torch.Generator(device="cuda:0")
g.manual_seed(0)
pool = torch.rand((64, n, 2), generator=g, device="cuda:0") * torch.tensor([2.0, 0.3], device="cuda:0") - torch.tensor([1.0, 0.15], device="cuda:0")
for rep, steps in enumerate((1500, 1, 1, 1, 1, 1, 1)):
    for i in range(steps):
        env.step_pipelined(pool[i % 64]) if env.sub_batches > 1 else env.step(pool[i % 64])
    torch.cuda.synchronize()
    if rep == 0:
        continue
    st = env.read("STAMPS").cpu().numpy().astype(np.float64)
    cnt = env.read("COUNTERS").cpu().numpy().reshape(n, 4)
    near = (env.read("NEARBY").cpu().numpy().reshape(n, -1) != 0).sum(axis=1)
    dur = (st[:, 4] - st[:, 2]) / 100.0
    # the step just taken ran with vessel step counter y (already incremented); an auto-reset zeroes it
    y = cnt[:, 1]
    interval = int(cfg.vessel.sensor_interval_load_obstacles)
    refreshed = (y % interval == 0) & (y > 0)
    line = "step %d: sweep us all p50 %.1f p99 %.1f max %.1f" % (rep, np.percentile(dur, 50), np.percentile(dur, 99), dur.max())
    if refreshed.any():
        line += " | refreshed (%d envs) p50 %.1f max %.1f | others p50 %.1f p99 %.1f max %.1f" % (
            refreshed.sum(), np.percentile(dur[refreshed], 50), dur[refreshed].max(), np.percentile(dur[~refreshed], 50),
            np.percentile(dur[~refreshed], 99), dur[~refreshed].max())
    front, pairs, back = (st[:, 8] - st[:, 2]) / 100.0, (st[:, 11] - st[:, 8]) / 100.0, (st[:, 4] - st[:, 11]) / 100.0
    top = np.argsort(-dur)[:16]
    print("   phases (us) front / staging+pairs / returns: all envs median %.1f / %.1f / %.1f; slowest 16 median %.1f / %.1f / %.1f; nearby >= 10 (%d envs) median %.1f / %.1f / %.1f"
          % (np.median(front), np.median(pairs), np.median(back), np.median(front[top]), np.median(pairs[top]), np.median(back[top]),
             (near >= 10).sum(), np.median(front[near >= 10]), np.median(pairs[near >= 10]), np.median(back[near >= 10])))
    line += " | slowest 16: refreshed %d, nearby %s (all envs: mean %.1f)" % (refreshed[top].sum(), near[top].tolist(), near.mean())
    print(line)
    for lo, c in env._slices:
        d = dur[lo:lo + c]
        end = (st[lo:lo + c, 4] - st[lo:lo + c, 3].min()) / 100.0
        a = int(np.argmax(end))
        print("   chain [%d, %d): last sweep ends %.1f us after the launch's first wave; it took %.1f us, refreshed %s, nearby %d"
              % (lo, lo + c, end[a], d[a], bool(refreshed[lo + a]), near[lo + a]))
