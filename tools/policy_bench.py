#!/usr/bin/env python3
"""Time of the fused policy launch alone (csrc/k6_policy.hip) and of policy + environment step chains: tools/policy_bench.py [envs]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
import ppo  # noqa: E402
from gym_auv_amd.batched_env import BatchedAuvEnv  # noqa: E402
from gym_auv_amd.config import effective_reference_config  # noqa: E402
from gym_auv_amd.devgen import GeneratedWorlds  # noqa: E402
from gym_auv_amd.policy import FusedActorCritic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
cfg = effective_reference_config(use_lidar=True)
BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
for k in (1, 2, 4):
    env = BatchedAuvEnv(cfg, GeneratedWorlds(2 * n, 17, 11, seed=1), n, device="cuda:0")
    env.reset()
    env.set_sub_batches(k)
    net = ppo.ActorCritic(env.obs_dim).to("cuda:0")
    T = 256
    fused = FusedActorCritic(net, env, rollout=T, reward_scale=0.01, bf16=BF16)
    # the policy launch alone, back to back on each chain's stream
    fused.begin_rollout()
    for _ in range(3):
        for i in range(env.sub_batches):
            fused.act(i)
    torch.cuda.synchronize()
    fused.begin_rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        for i in range(env.sub_batches):
            fused.act(i)
    torch.cuda.synchronize()
    t_pol = (time.perf_counter() - t0) / 200
    # policy + step chains
    fused.begin_rollout()
    fused.rollout(16, flush=False)
    torch.cuda.synchronize()
    fused.begin_rollout()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fused.rollout(T)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_roll = (time.perf_counter() - t0) / T
    print("envs %d chains %d: policy launches alone %.1f us per full step; policy + env step %.1f us per step = %.1f M env-steps/s "
          "(host enqueue %.1f us per step)" % (n, env.sub_batches, 1e6 * t_pol, 1e6 * t_roll, n / t_roll / 1e6, 1e6 * t_host / T), flush=True)
    env.close()
