#!/usr/bin/env python3
"""What does a side stream cost THREE sub-batch chains, by what it runs?  (a pass of the fresh-world mode is eight kernels of
~300 us in all on a stream that shares no hardware queue with the chains: are the chains paying for the work, for the slot, or
for the kernel boundaries -- each one a cache write-back / invalidate that every running kernel sees?)
usage: python tools/side_queue_probe2.py"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from gym_auv_amd.batched_env import BatchedAuvEnv, _LIB  # noqa: E402
from gym_auv_amd.config import effective_reference_config  # noqa: E402
from gym_auv_amd.devgen import GeneratedWorlds  # noqa: E402

dev = torch.device("cuda:0")
cfg = effective_reference_config(use_lidar=True)
n = 4096
env = BatchedAuvEnv(cfg, GeneratedWorlds(2 * n, seed=1), n, device=dev, auto_reset=True)
env.reset()
streams = env._concurrent_streams(4)
assert len(streams) == 4
side = streams.pop()
env.set_sub_batches(3, probe_streams=False)
env._sub_streams = streams
env._streams_c = (C.c_void_p * 3)(*[s.cuda_stream for s in streams])
pool = torch.rand((16, n, 2), device=dev) * torch.tensor([2.0, 0.3], device=dev) - torch.tensor([1.0, 0.15], device=dev)
buf = torch.zeros(1 << 20, device=dev)
GHZ = 2.1e9


def rate(work, steps=1600, every=16):
    for i in range(200):
        env.step_pipelined(pool[i % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step_pipelined(pool[i % 16])
        if work is not None and i % every == 0:
            with torch.cuda.stream(side):
                work()
    torch.cuda.synchronize()
    return round(n * steps / (time.perf_counter() - t0) / 1e6, 1)


def sleeps(k, us):
    def f():
        for _ in range(k):
            torch.cuda._sleep(int(us * 1e-6 * GHZ))
    return f


def fills(k):
    def f():
        for _ in range(k):
            buf[:64].fill_(1.0)
    return f


out = dict(no_side=rate(None), one_sleep_300us=rate(sleeps(1, 300)), eight_sleeps_37us=rate(sleeps(8, 37)), eight_tiny_fills=rate(fills(8)),
           thirtytwo_tiny_fills=rate(fills(32)), one_tiny_fill=rate(fills(1)))
print(json.dumps(out))
