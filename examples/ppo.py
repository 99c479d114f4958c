#!/usr/bin/env python3
"""PPO on the batched MI355X environment, tensors never leave the GPU.

Hyper-parameters follow the reference's training script (/root/reference/scripts/run.py:332-357:
MlpPolicy [256, 128, 64], gamma 0.999, lambda 0.98, 4 epochs, 32 minibatches, ent_coef 0.01,
lr 2e-4, clip 0.2); the rollout length is shortened because one batched step already yields
thousands of transitions (the reference collected 8 envs x 1024 steps per update).

Scenarios are generated ON THE DEVICE: every environment gets its own
MovingObstacles world and the whole bank is regenerated from fresh random draws every `--regen`
updates (a few milliseconds), so training never waits for host-side world generation; finished
episodes in between restart on the next world of the bank.  `--worlds host` uses the host
generator (bit-compatible RNG streams with the reference) instead.

    python examples/ppo.py --envs 4096 --updates 20 --rollout 32
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/ppo.py   # data parallel
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class ActorCritic(nn.Module):
    def __init__(self, obs_dim, act_dim=2, hidden=(256, 128, 64)):
        super().__init__()
        def mlp(out):
            layers, d = [], obs_dim
            for h in hidden:
                layers += [nn.Linear(d, h), nn.Tanh()]
                d = h
            return nn.Sequential(*layers, nn.Linear(d, out))
        self.pi, self.v = mlp(act_dim), mlp(1)
        self.log_std = nn.Parameter(torch.full((act_dim,), -0.5))

    def dist(self, obs):
        return torch.distributions.Normal(self.pi(obs), self.log_std.exp(), validate_args=False)   # (no host sync: graph-capturable)


def train(envs=4096, updates=10, rollout=32, device="cuda:0", seed=0, log=print, worlds="generated", regen=5, log_every=1,
          task="colav", step_mode=None, graph_rollout=False, graph_update=False):
    from gym_auv_amd import distributed as D
    from gym_auv_amd.batched_env import BatchedAuvEnv
    from gym_auv_amd.config import effective_reference_config
    from gym_auv_amd.devgen import GeneratedWorlds
    from gym_auv_amd.world import build_bank_parallel
    rank, world = D.rank(), D.world_size()
    torch.manual_seed(seed + rank)
    # task "colav": MovingObstaclesNoRules-v0 (17 moving + 11 static obstacles, LiDAR on, ColavRewarder);
    # task "pathfollow": PathFollowNoObstacles-v0 (no obstacles, LiDAR off, PathFollowRewarder) -- gym_auv/__init__.py:105-121
    colav = task == "colav"
    cfg = effective_reference_config(use_lidar=colav)
    nm, ns = (17, 11) if colav else (0, 0)
    if worlds == "generated":
        bank = GeneratedWorlds(envs, nm, ns, seed=1000 * seed + rank)
    else:
        bank = build_bank_parallel("moving_obstacles_world", range(5000 + 512 * rank, 5000 + 512 * rank + min(envs, 512)),
                                   procs=min(8, os.cpu_count() or 1), **({} if colav else dict(n_moving=0, n_static=0)))
    env = BatchedAuvEnv(cfg, bank, envs, device=device, auto_reset=True, rewarder="colav" if colav else "pathfollow")
    if step_mode:
        env.set_step_mode(step_mode)
    low = torch.as_tensor(env.action_space.low, device=device)
    high = torch.as_tensor(env.action_space.high, device=device)
    net = ActorCritic(env.obs_dim).to(device)
    if world > 1:   # data parallel over GPUs: same initial weights, gradients averaged over RCCL
        for prm in net.parameters():
            torch.distributed.broadcast(prm.data, 0)
    graph_update = graph_update and world == 1            # (the gradient all-reduce of data parallelism stays eager)
    opt = torch.optim.Adam(net.parameters(), lr=2e-4, capturable=graph_update)
    upd_graph, upd_in, upd_loss = None, None, None
    gamma, lam, clip, ent_coef, epochs, n_mb = 0.999, 0.98, 0.2, 0.01, 4, 32
    obs = env.reset().clone()
    history = []
    # --graph-rollout: policy forward, sampling, the environment's step (one kernel launch, enqueued on torch's
    # capture stream through the C ABI) and the value net as ONE captured device graph, replayed once per step
    roll = None
    if graph_rollout:
        s_obs = obs.clone()
        s_a = torch.zeros((envs, 2), device=device)
        s_lp, s_v = torch.zeros(envs, device=device), torch.zeros(envs, device=device)
        s_rew, s_done = torch.zeros(envs, device=device), torch.zeros(envs, device=device)

        def one_step():
            with torch.no_grad():
                mu, log_std = net.pi(s_obs), net.log_std
                a = mu + log_std.exp() * torch.randn_like(mu)        # (torch.normal with tensor arguments does not capture)
                nobs, rew, done, _ = env.step(torch.max(torch.min(a, high), low))
                lp = (-0.5 * ((a - mu) / log_std.exp()) ** 2 - log_std - 0.9189385332046727).sum(-1)
                s_a.copy_(a), s_lp.copy_(lp), s_v.copy_(net.v(s_obs).squeeze(-1))
                s_rew.copy_(rew), s_done.copy_(done.float())
                return nobs
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):                      # (warm-up off the capture, as torch asks)
            for _ in range(3):
                one_step()
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize()
        env.reset()
        roll = torch.cuda.CUDAGraph()
        with torch.cuda.graph(roll):
            s_next = one_step()
        obs = env.reset().clone()
    for upd in range(updates):
        if worlds == "generated" and regen > 0 and upd and upd % regen == 0:
            # fresh scenarios for every environment, built on the device; all envs restart
            env.generate(GeneratedWorlds(envs, nm, ns, seed=1000 * seed + rank + 7919 * upd))
            obs = env.reset().clone()
        t0 = time.time()
        O, A, LP, R, Dn, V = [], [], [], [], [], []
        with torch.no_grad():
            for _ in range(rollout if roll is None else 0):
                dist = net.dist(obs)
                a = dist.sample()
                nobs, rew, done, _ = env.step(torch.max(torch.min(a, high), low))
                O.append(obs), A.append(a), LP.append(dist.log_prob(a).sum(-1)), V.append(net.v(obs).squeeze(-1))
                R.append(rew.clone() * 0.01), Dn.append(done.float())         # reward scale for the value net
                obs = nobs.clone()
            for _ in range(rollout if roll is not None else 0):
                s_obs.copy_(obs)
                roll.replay()
                O.append(obs), A.append(s_a.clone()), LP.append(s_lp.clone()), V.append(s_v.clone())
                R.append(s_rew * 0.01), Dn.append(s_done.clone())
                obs = s_next.clone()
            last_v = net.v(obs).squeeze(-1)
            adv, gae = [None] * rollout, torch.zeros(envs, device=device)
            for t in reversed(range(rollout)):
                nv = last_v if t == rollout - 1 else V[t + 1]
                delta = R[t] + gamma * nv * (1 - Dn[t]) - V[t]
                gae = delta + gamma * lam * (1 - Dn[t]) * gae
                adv[t] = gae
        torch.cuda.synchronize()
        t_roll = time.time() - t0
        O, A, LP, V = torch.cat(O), torch.cat(A), torch.cat(LP), torch.cat(V)
        ADV = torch.cat(adv)
        RET = ADV + V
        ADV = (ADV - ADV.mean()) / (ADV.std() + 1e-8)
        n = O.shape[0]

        def minibatch_step(o, a, lp, advn, ret):
            dist = net.dist(o)
            ratio = (dist.log_prob(a).sum(-1) - lp).exp()
            pg = -torch.min(ratio * advn, ratio.clamp(1 - clip, 1 + clip) * advn).mean()
            vf = 0.5 * (net.v(o).squeeze(-1) - ret).pow(2).mean()
            loss = pg + 0.5 * vf - ent_coef * dist.entropy().sum(-1).mean()
            opt.zero_grad(set_to_none=not graph_update)
            loss.backward()
            if world > 1:
                for prm in net.parameters():
                    torch.distributed.all_reduce(prm.grad)
                    prm.grad /= world
            nn.utils.clip_grad_norm_(net.parameters(), 0.5)
            opt.step()
            return loss

        for _ in range(epochs):
            perm = torch.randperm(n, device=device)
            for mb in perm.chunk(n_mb):
                if not graph_update:
                    loss = minibatch_step(O[mb], A[mb], LP[mb], ADV[mb], RET[mb])
                    continue
                # --graph-update: forward, backward, gradient clipping and the Adam step of one minibatch as ONE captured
                # device graph (static input buffers, refilled by gathers before every replay)
                if upd_in is None:
                    upd_in = [O[mb].clone(), A[mb].clone(), LP[mb].clone(), ADV[mb].clone(), RET[mb].clone()]
                    side = torch.cuda.Stream(device=device)
                    side.wait_stream(torch.cuda.current_stream(device))
                    with torch.cuda.stream(side):              # (warm-up off the capture: three real steps on this minibatch)
                        for _w in range(3):
                            minibatch_step(*upd_in)
                    torch.cuda.current_stream(device).wait_stream(side)
                    upd_graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(upd_graph):
                        upd_loss = minibatch_step(*upd_in)
                    continue
                for buf, src in zip(upd_in, (O, A, LP, ADV, RET)):
                    torch.index_select(src, 0, mb, out=buf)
                upd_graph.replay()
                loss = upd_loss
        torch.cuda.synchronize()
        mean_r = float(torch.stack(R).mean().item()) / 0.01
        sps = world * envs * rollout / (time.time() - t0)
        # what the policy does, read off the observations of the rollout (vessel.py:24-35: surge, sway, yaw
        # rate, look-ahead heading error, heading error, cross-track error / 100)
        surge, he, cte = float(O[:, 0].mean()), float(O[:, 4].abs().mean()), float(O[:, 5].abs().mean()) * 100
        history.append((mean_r, float(loss.item()), sps, surge, he, cte))
        if rank == 0 and (upd % log_every == 0 or upd == updates - 1):
            log("update %3d  mean step reward %8.3f  surge %.3f m/s  |heading error| %.2f rad  |cross-track| %6.1f m  loss %8.4f  "
                "rollout %.2e env-steps/s (policy in the loop), %.2e incl. learning"
                % (upd, mean_r, surge, he, cte, loss.item(), world * envs * rollout / t_roll, sps))
    # the one collective of the environment side: finished-episode statistics of all ranks
    stats = D.gather_episode_stats(env.episode_stats())
    if rank == 0:
        fin = stats["episodes"] > 0
        log("episodes finished %d; last-episode return mean %.1f, collision rate %.2f, goal rate %.2f" % (
            int(stats["episodes"].sum()), float(stats["episode_return"][fin].mean()) if fin.any() else float("nan"),
            float(stats["collision"][fin].mean()) if fin.any() else float("nan"),
            float(stats["reached_goal"][fin].mean()) if fin.any() else float("nan")))
        k = max(1, min(10, len(history) // 4))
        first, last = history[:k], history[-k:]
        log("learning: mean step reward %.3f -> %.3f, surge %.3f -> %.3f m/s, |heading error| %.2f -> %.2f rad (first / last %d updates)"
            % (sum(h[0] for h in first) / k, sum(h[0] for h in last) / k, sum(h[3] for h in first) / k, sum(h[3] for h in last) / k,
               sum(h[4] for h in first) / k, sum(h[4] for h in last) / k, k))
    env.close()
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--updates", type=int, default=10)
    ap.add_argument("--rollout", type=int, default=32)
    ap.add_argument("--worlds", default="generated", choices=["generated", "host"])
    ap.add_argument("--regen", type=int, default=5, help="regenerate the world bank on the device every this many updates")
    ap.add_argument("--log-every", type=int, default=1)
    ap.add_argument("--task", default="colav", choices=["colav", "pathfollow"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--graph-rollout", type=int, default=0, help="1: one captured device graph per rollout step")
    ap.add_argument("--graph-update", type=int, default=0, help="1: one captured device graph per minibatch step of the update")
    ap.add_argument("--step-mode", default=None, help="launch shape of a step (BatchedAuvEnv.STEP_MODES); default: the library's")
    a = ap.parse_args()
    from gym_auv_amd import distributed as D
    _rank, _world, local = D.init_from_env()     # one process per GPU under torch.distributed.run; cuda:0 alone
    train(a.envs, a.updates, a.rollout, device="cuda:%d" % local, seed=a.seed, worlds=a.worlds, regen=a.regen, log_every=a.log_every,
          task=a.task, step_mode=a.step_mode, graph_rollout=bool(a.graph_rollout), graph_update=bool(a.graph_update))
