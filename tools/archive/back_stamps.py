#!/usr/bin/env python3
"""Diagnostic (scratch STAMPS build csrc_stampsb, tools/archive/build_stampsb.sh): where the LiDAR wave's time goes after the pair sweep.
    AUV_HIP_LIB=gym_auv_amd/csrc_stampsb/libauv_hip.so SUB=4 python tools/archive/back_stamps.py"""
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
env.set_step_mode("one_launch")
sub = int(os.environ.get("SUB", "4"))
env.set_sub_batches(sub)
env.reset()
g = torch.Generator(device="cuda:0")
g.manual_seed(0)
pool = torch.rand((64, n, 2), generator=g, device="cuda:0") * torch.tensor([2.0, 0.3], device="cuda:0") - torch.tensor([1.0, 0.15], device="cuda:0")
for rep, steps in enumerate((1500, 1, 1, 1, 1)):
    for i in range(steps):
        env.step_pipelined(pool[i % 64]) if env.sub_batches > 1 else env.step(pool[i % 64])
    torch.cuda.synchronize()
    if rep == 0:
        continue
    st = env.read("STAMPS").cpu().numpy().astype(np.float64) / 100.0
    lid = env.read("LIDAR_D").cpu().numpy().reshape(n, -1)
    nh = (lid < cfg.vessel.sensor_range).sum(axis=1)
    ok = st[:, 12] > st[:, 11]          # environments that swept (n_act > 0)
    names = ["pre-work", "wait state", "front", "stage+pairs", "compaction", "wait weights", "free rows", "returns", "wave sum", "rest", "drain+word"]
    seq = [3, 1, 2, 8, 11, 10, 13, 12, 5, 6, 4, 14]
    dur = np.stack([st[:, seq[i + 1]] - st[:, seq[i]] for i in range(len(names))], axis=1)
    tot = st[:, 14] - st[:, 2]
    for label, sel in (("all sweeping (%d)" % ok.sum(), ok), ("slowest 64", np.isin(np.arange(n), np.argsort(-tot)[:64]) & ok)):
        print("step %d %s: " % (rep, label) + "  ".join("%s %.2f" % (nm, np.median(dur[sel, i])) for i, nm in enumerate(names))
              + "  | total from state %.2f, beams with a return %d" % (np.median(tot[sel]), np.median(nh[sel])))
    print("   not sweeping (%d envs): total from state %.2f" % ((~ok).sum(), np.median(tot[~ok])))
