#!/usr/bin/env python3
"""Does a side stream that runs a long kernel every few steps slow the four sub-batch chains down?  Only if it shares a hardware
queue with one of them (HIP multiplexes streams onto GPU_MAX_HW_QUEUES queues; a queue is FIFO).  For a handful of candidate
side streams: overlap with each chain stream (auv_streams_overlap: ~1 side by side, ~2 one after the other) and the open-loop
rate of the chains while the candidate runs a 300 us do-nothing kernel every 8 steps.
usage: [GPU_MAX_HW_QUEUES=8] python tools/side_queue_probe.py"""
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
env.set_sub_batches(4, strict=True)
pool = torch.rand((16, n, 2), device=dev) * torch.tensor([2.0, 0.3], device=dev) - torch.tensor([1.0, 0.15], device=dev)


def rate(side, steps=1500, every=8, spin_cycles=int(300e-6 * 2.1e9)):
    for i in range(200):
        env.step_pipelined(pool[i % 16])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        env.step_pipelined(pool[i % 16])
        if side is not None and i % every == 0:
            with torch.cuda.stream(side):
                torch.cuda._sleep(spin_cycles)
    torch.cuda.synchronize()
    return n * steps / (time.perf_counter() - t0)


out = dict(hw_queues=os.environ.get("GPU_MAX_HW_QUEUES", "default"), no_side=round(rate(None) / 1e6, 1), candidates=[])
ratio = C.c_float()
for k in range(6):
    s = torch.cuda.Stream(device=dev)
    ov = []
    for st in env._sub_streams:
        _LIB.auv_streams_overlap(env._h, C.c_void_p(s.cuda_stream), C.c_void_p(st.cuda_stream), C.byref(ratio))
        ov.append(round(ratio.value, 2))
    out["candidates"].append(dict(overlap_ratio_with_chains=ov, rate_with_300us_every_8_steps=round(rate(s) / 1e6, 1)))
print(json.dumps(out))
