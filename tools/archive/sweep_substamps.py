#!/usr/bin/env python3
"""Diagnostic (STAMPS build, three-launch shape): staging / item-prefix / pair-sweep cycles of every sweep, and what they scale with.
    AUV_HIP_LIB=gym_auv_amd/csrc_stamps/libauv_hip.so python tools/archive/sweep_substamps.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gym_auv_amd.batched_env import BatchedAuvEnv
from gym_auv_amd.config import effective_reference_config

cfg = effective_reference_config(use_lidar=True)
n = 4096
z = np.load(os.environ.get("BANK") or sorted(__import__("glob").glob("/tmp/bank.polygons50.0.4096.4096.2*.npz"))[0])
bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
env = BatchedAuvEnv(cfg, bank, n, auto_reset=True)
env.set_step_mode("side_by_side")
env.reset()
g = torch.Generator(device="cuda:0")
g.manual_seed(0)
pool = torch.rand((64, n, 2), generator=g, device="cuda:0") * torch.tensor([2.0, 0.3], device="cuda:0") - torch.tensor([1.0, 0.15], device="cuda:0")
for i in range(1500):
    env.step(pool[i % 64])
torch.cuda.synchronize()
st = env.read("STAMPS").cpu().numpy().astype(np.float64)
stage, prefix, items, n_it, t_raw = st[:, 7], st[:, 14], st[:, 15], st[:, 6], st[:, 5]
ok = n_it > 0
tot = stage + prefix + items
print("sweeping envs %d; cycles (shader clock) median: staging %.0f prefix %.0f items %.0f | work items median %.0f, raw segments median %.0f"
      % (ok.sum(), np.median(stage[ok]), np.median(prefix[ok]), np.median(items[ok]), np.median(n_it[ok]), np.median(t_raw[ok])))
top = np.argsort(-tot)[:64]
print("slowest 64: staging %.0f prefix %.0f items %.0f | work items %.0f (max %.0f), raw segments %.0f (max %.0f)"
      % (np.median(stage[top]), np.median(prefix[top]), np.median(items[top]), np.median(n_it[top]), n_it[top].max(), np.median(t_raw[top]), t_raw[top].max()))
for lo, hi in ((1, 64), (65, 128), (129, 192), (193, 256), (257, 100000)):
    m = ok & (n_it >= lo) & (n_it <= hi)
    if m.any():
        print("  work items %4d..%-6d %5d envs: items cycles median %.0f, staging %.0f (raw segments %.0f)" % (lo, hi, m.sum(), np.median(items[m]), np.median(stage[m]), np.median(t_raw[m])))
