#!/usr/bin/env python3
"""Where do the ~80-200 us go that a 20-step timed region carries beyond 20 x the sustained step time?
For warm-ups of 5 and 200 steps: host time to enqueue the 20 steps, time until all chains are done seen by polling
stream.query() from the host, and seen by torch.cuda.synchronize() (the contract's bracket).
    python tools/archive/short_timing.py [--steps 20]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--sub", type=int, default=4)
    args = ap.parse_args()
    from gym_auv_amd.batched_env import BatchedAuvEnv
    from gym_auv_amd.config import effective_reference_config
    cfg = effective_reference_config(use_lidar=True)
    n = 4096
    z = np.load(os.environ.get("BANK") or sorted(__import__("glob").glob("/tmp/bank.polygons50.0.4096.4096.2*.npz"))[0])   # (bench.py --bank-cache /tmp/bank: name carries a source hash)
    bank = {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    pool = torch.rand((64, n, 2), generator=g, device=dev) * torch.tensor([2.0, 0.3], device=dev) - torch.tensor([1.0, 0.15], device=dev)
    for warm in (5, 200, 5, 200):
        for how in ("sync", "poll"):
            env = BatchedAuvEnv(cfg, bank, n, device=dev, auto_reset=True)
            env.set_sub_batches(args.sub)
            env.reset()
            for i in range(warm):
                env.step_pipelined(pool[i % 64])
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(args.steps):
                env.step_pipelined(pool[(warm + i) % 64])
            t1 = time.perf_counter()
            if how == "poll":
                while not all(s.query() for s in env._sub_streams):
                    pass
            t2 = time.perf_counter()
            torch.cuda.synchronize(dev)
            t3 = time.perf_counter()
            print("warm-up %3d, %s: enqueue %6.1f us, done seen after %6.1f us, synchronize returned after %6.1f us  -> %.1f M env-steps/s"
                  % (warm, how, 1e6 * (t1 - t0), 1e6 * (t2 - t0), 1e6 * (t3 - t0), n * args.steps / (t3 - t0) / 1e6), flush=True)
            del env


if __name__ == "__main__":
    main()
