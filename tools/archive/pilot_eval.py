#!/usr/bin/env python3
"""Env-side diagnosis for F2 (is the reference's task -- MovingObstaclesNoRules-v0 with the Colav reward -- achievable at
all in this environment?): hand-written controllers, no learning, on device-generated worlds, scored by the library's
episode log: goal rate, collision rate, what else ends episodes, return, length.

  blind     full thrust, rudder = look-ahead pilot on the heading error (observation 4), ignores the LiDAR
  avoid     the same plus a reactive term: the closeness-weighted bearing of what the forward beams see pushes the rudder
            away from it

    python tools/archive/pilot_eval.py --envs 4096 --steps 12000
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=12000)
    ap.add_argument("--gain", type=float, default=1.5)
    ap.add_argument("--avoid-gain", type=float, default=2.0)
    ap.add_argument("--task", default="colav")
    args = ap.parse_args()
    from gym_auv_amd.batched_env import BatchedAuvEnv
    from gym_auv_amd.config import effective_reference_config
    from gym_auv_amd.devgen import GeneratedWorlds
    dev = torch.device("cuda:0")
    n = args.envs
    cfg = effective_reference_config(use_lidar=True)
    S = cfg.vessel.n_sensors
    ang = -np.pi + (np.arange(S) + 1) * 2 * np.pi / S                        # body-frame beam angles (vessel.py:66-68)
    ang_t = torch.as_tensor(ang, device=dev, dtype=torch.float32)
    front = (ang_t.abs() < np.pi / 2).float()
    w_side = -torch.sign(ang_t) * torch.cos(ang_t).clamp(min=0.0) * front      # something ahead-left pushes right and vice versa
    out = {}
    for name in ("blind", "avoid"):
        env = BatchedAuvEnv(cfg, GeneratedWorlds(2 * n, 17, 11, seed=11), n, device=dev, auto_reset=True)
        obs = env.reset()
        act = torch.ones((n, 2), dtype=torch.float32, device=dev)
        rows = []
        for t in range(args.steps):
            steer = args.gain * obs[:, 4]
            if name == "avoid":
                clos = obs[:, 6:6 + S]
                steer = steer + args.avoid_gain * (clos * w_side).sum(1) / 8.0
                # slow down when something is close dead ahead
                ahead = (clos * (ang_t.abs() < 0.35).float()).max(1).values
                act[:, 0] = torch.where(ahead > 0.5, torch.full_like(ahead, 0.3), torch.ones_like(ahead))
            act[:, 1] = steer.clamp(-0.15, 0.15)
            obs, rew, done, _ = env.step(act)
            if t % 500 == 499:
                rows.append(env.episode_log().cpu().numpy())
        rows.append(env.episode_log().cpu().numpy())
        log = np.concatenate(rows) if rows else np.zeros((0, 8))
        k = len(log)
        col, goal = log[:, 3] > 0, log[:, 4] > 0
        other = ~col & ~goal
        out[name] = dict(episodes=int(k), goal_rate=round(float(goal.mean()), 4), collision_rate=round(float(col.mean()), 4),
                         other_rate=round(float(other.mean()), 4), mean_return=round(float(log[:, 1].mean()), 1),
                         mean_timesteps=round(float(log[:, 2].mean()), 1), mean_progress=round(float(log[:, 5].mean()), 3),
                         mean_cross_track_error=round(float(log[:, 6].mean()), 2),
                         return_goal=round(float(log[goal, 1].mean()), 1) if goal.any() else None,
                         timesteps_goal=round(float(log[goal, 2].mean()), 1) if goal.any() else None,
                         return_other=round(float(log[other, 1].mean()), 1) if other.any() else None,
                         timesteps_other=round(float(log[other, 2].mean()), 1) if other.any() else None)
        print(name, json.dumps(out[name]), flush=True)
        env.close()


if __name__ == "__main__":
    main()
