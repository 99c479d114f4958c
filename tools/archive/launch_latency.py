#!/usr/bin/env python3
"""Host-visible latency of the first work after a device synchronize: a trivial torch kernel on the current stream /
on a side stream, and one step of the batch (four chains) -- wall time until stream.query() turns true."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
side = torch.cuda.Stream(dev)
cur = torch.cuda.current_stream(dev)


def wait(streams):
    while not all(s.query() for s in streams):
        pass


for name, st in (("current stream", cur), ("side stream", side)):
    ts = []
    for _ in range(20):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            x.add_(1.0)
        wait([st])
        ts.append(1e6 * (time.perf_counter() - t0))
    print("trivial kernel after synchronize, %s: median %.1f us, min %.1f, max %.1f" % (name, float(np.median(ts)), min(ts), max(ts)))

from gym_auv_amd.batched_env import BatchedAuvEnv  # noqa: E402
from gym_auv_amd.config import effective_reference_config  # noqa: E402
cfg = effective_reference_config(use_lidar=True)
n = 4096
z = np.load(os.environ.get("BANK") or sorted(__import__("glob").glob("/tmp/bank.polygons50.0.4096.4096.2*.npz"))[0])   # (bench.py --bank-cache /tmp/bank: name carries a source hash)
bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
env = BatchedAuvEnv(cfg, bank, n, device=dev, auto_reset=True)
env.set_sub_batches(4)
env.reset()
a = torch.zeros((n, 2), device=dev)
for nsteps in (1, 2, 5, 20):
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(nsteps):
            env.step_pipelined(a)
        t1 = time.perf_counter()
        wait(env._sub_streams)
        ts.append((1e6 * (t1 - t0), 1e6 * (time.perf_counter() - t0)))
    print("%2d step(s) of four chains after synchronize: enqueue median %.1f us, done median %.1f us (min %.1f)"
          % (nsteps, float(np.median([t[0] for t in ts])), float(np.median([t[1] for t in ts])), min(t[1] for t in ts)))
