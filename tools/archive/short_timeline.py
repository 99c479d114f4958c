#!/usr/bin/env python3
"""Timeline of the driver's short command (--steps 20 --warmup 5): when does step i of chain c complete, relative to
the start of the timed region?  One event per chain and step, recorded on the chain's stream behind its launch.
    python tools/archive/short_timeline.py [--steps 20] [--warmup 5] [--sub 4]"""
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
    ap.add_argument("--warmup", type=int, default=5)
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
    for rep in range(3):
        env = BatchedAuvEnv(cfg, bank, n, device=dev, auto_reset=True)
        env.set_sub_batches(args.sub)
        env.reset()
        for i in range(args.warmup):
            env.step_pipelined(pool[i % 64])
        torch.cuda.synchronize(dev)
        evs = [[torch.cuda.Event(enable_timing=True) for _ in range(args.steps)] for _ in env._sub_streams]
        e0 = [torch.cuda.Event(enable_timing=True) for _ in env._sub_streams]
        t0 = time.perf_counter()
        for c, s in enumerate(env._sub_streams):
            e0[c].record(s)
        for i in range(args.steps):
            env.step_pipelined(pool[(args.warmup + i) % 64])
            for c, s in enumerate(env._sub_streams):
                evs[c][i].record(s)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        print("rep %d: wall %.1f us for %d steps" % (rep, 1e6 * (t1 - t0), args.steps))
        for c in range(len(evs)):
            done = [1e3 * e0[c].elapsed_time(ev) for ev in evs[c]]
            per = [done[0]] + [b - a for a, b in zip(done[:-1], done[1:])]
            print("  chain %d: step periods (us) %s | last done at %.1f" % (c, " ".join("%.0f" % p for p in per), done[-1]))
        del env


if __name__ == "__main__":
    main()
