#!/usr/bin/env python3
"""PPO on the batched MI355X environment, tensors never leave the GPU.

Hyper-parameters follow the reference's training script (/root/reference/scripts/run.py:332-357: MlpPolicy
[256, 128, 64] for policy and value, gamma 0.999, lambda 0.98, 4 epochs, 32 minibatches, ent_coef 0.01, lr 2e-4,
clip 0.2).  The rollout length is a flag (the reference collected 8 envs x 1024 steps per update; one batched step
already yields thousands of transitions).

The rollout runs as K independent chains (BatchedAuvEnv.set_sub_batches): the policy of sub-batch A is evaluated while
sub-batch B steps -- the shape of stable-baselines' step_async / step_wait over SubprocVecEnv workers (run.py:293-296),
without a per-step rendezvous.  With --graph-rollout every chain's step (policy forward, sampling, the environment's
one-launch step through the C ABI, value net, storing the transition) is ONE captured device graph replayed on the
chain's stream.

Trainer-side options (none of them changes the environment): --act-space normalized samples in the box [-1, 1]^2 and
maps it onto the action space (default: raw units clipped to the space, as stable-baselines does with a Box -- with the
rudder's range of +-0.15 a Gaussian of std 0.6 is then clipped most of the time, i.e. bang-bang steering, which is what
learns fastest here); --ret-norm 1 normalises the returns by a running mean / std for the value head (a collision is
-5000 against step rewards of order 1); --orthogonal 1 starts the last policy layer small.  Progress is read from the
library's episode log (auv_episode_log): goal / collision / give-up rates of the episodes that ended during each update.

Scenarios are generated ON THE DEVICE, and by default EVERY FINISHED EPISODE LANDS ON A WORLD NOBODY HAS SEEN
(`--worlds fresh`, BatchedAuvEnv(worlds=FreshWorlds(...))): what the reference's reset() -> _generate() gives its learner
(environment.py:176-218), with the generator running beside the rollouts on a side stream.  `--worlds generated` keeps a bank
of two worlds per environment that auto-reset cycles through (optionally rebuilt every `--regen` updates, which restarts every
episode); `--worlds host` uses the host generator (bit-compatible RNG streams with the reference).

    python examples/ppo.py --envs 4096 --updates 200 --rollout 256 --graph-rollout 1
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/ppo.py   # data parallel
"""
import argparse
import os
import sys
import time

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

LOG_SQRT_2PI = 0.9189385332046727


class ActorCritic(nn.Module):
    def __init__(self, obs_dim, act_dim=2, hidden=(256, 128, 64), log_std=-0.5, orthogonal=False):
        super().__init__()

        def mlp(out, last_gain):
            layers, d = [], obs_dim
            for h in hidden:
                lin = nn.Linear(d, h)
                if orthogonal:
                    nn.init.orthogonal_(lin.weight, 2 ** 0.5)
                    nn.init.zeros_(lin.bias)
                layers += [lin, nn.Tanh()]
                d = h
            lin = nn.Linear(d, out)
            if orthogonal:
                nn.init.orthogonal_(lin.weight, last_gain)
                nn.init.zeros_(lin.bias)
            return nn.Sequential(*layers, lin)
        self.pi, self.v = mlp(act_dim, 0.01), mlp(1, 1.0)
        self.log_std = nn.Parameter(torch.full((act_dim,), float(log_std)))

    def log_prob(self, mu, a):
        return (-0.5 * ((a - mu) / self.log_std.exp()) ** 2 - self.log_std - LOG_SQRT_2PI).sum(-1)

    def entropy(self):
        return (0.5 + LOG_SQRT_2PI + self.log_std).sum()


def clip_grad_norm(params, max_norm):
    """Global-norm gradient clipping in plain tensor operations (no host synchronisation), safe inside a captured graph.
    The norm is accumulated with 0-d additions, NOT torch.stack / torch.cat: on ROCm those stage the table of their inputs'
    addresses through pinned host memory with an asynchronous copy, a stream capture records that copy with the HOST
    address, the caching host allocator recycles the buffer, and a replay then uploads somebody else's bytes as the table --
    the "gradient norm" comes out as inf although every gradient element is finite, the clip scales the gradients to
    zero and the policy freezes silently (what round 2 / 3's --graph-update did; docs/HISTORY.md)."""
    grads = [p.grad for p in params if p.grad is not None]
    sq = None
    for g in grads:
        s = (g * g).sum()
        sq = s if sq is None else sq + s
    total = torch.sqrt(sq)
    coef = (max_norm / (total + 1e-6)).clamp(max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


PROBE = {} if os.environ.get("AUV_PPO_PROBE") else None


def probe_report(probe, diag_row, log):
    """A replayed update step has written a corrupt record: where do the record's inputs live, what do they
    hold NOW, and which other live tensor shares their bytes?"""
    import gc
    torch.cuda.synchronize()
    spans = {k: (v.data_ptr(), v.data_ptr() + max(1, v.numel()) * v.element_size()) for k, v in probe.items()}
    for k, v in probe.items():
        raw = v.view(torch.uint8) if v.dim() else v.reshape(1).view(torch.uint8)
        log("probe %-10s ptr 0x%x dtype %s value now %s bytes %s" % (k, v.data_ptr(), v.dtype, v.item(), bytes(raw.cpu().tolist()).hex()))
    log("probe diag_row ptr 0x%x now %s" % (diag_row.data_ptr(), diag_row.tolist()))
    seen = 0
    for obj in gc.get_objects():
        try:
            if not (torch.is_tensor(obj) and obj.is_cuda and obj.numel()):
                continue
            lo = obj.untyped_storage().data_ptr()
            hi = lo + obj.untyped_storage().nbytes()
        except Exception:
            continue
        for k, (a, b) in spans.items():
            if lo < b and a < hi and obj is not probe[k]:
                seen += 1
                log("probe OVERLAP %s [0x%x, 0x%x) with live tensor storage [0x%x, 0x%x) shape %s dtype %s" % (k, a, b, lo, hi, tuple(obj.shape), obj.dtype))
    log("probe overlaps with live tensors: %d" % seen)
    try:
        for seg in torch.cuda.memory_snapshot():
            for k, (a, b) in spans.items():
                if seg["address"] <= a < seg["address"] + seg["total_size"]:
                    blk = [bl for bl in _blocks(seg) if bl[0] <= a < bl[0] + bl[1]]
                    log("probe %-10s in segment 0x%x size %d pool %s stream %s; block %s" % (k, seg["address"], seg["total_size"], seg.get("segment_pool_id"), seg.get("stream"), blk))
    except Exception as exc:
        log("probe: no allocator snapshot (%s)" % exc)


def _blocks(seg):
    out, addr = [], seg["address"]
    for bl in seg["blocks"]:
        out.append((addr, bl["size"], bl["state"]))
        addr += bl["size"]
    return out


def average_gradients(params, world):
    """Data parallelism: gradients averaged over the ranks (RCCL all-reduce over xGMI; gloo on CPU).  One flat
    buffer per call -- a single collective instead of one per parameter tensor."""
    grads = [p.grad for p in params if p.grad is not None]
    if world <= 1 or not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    torch.distributed.all_reduce(flat)
    flat /= world
    off = 0
    for g in grads:
        g.copy_(flat[off:off + g.numel()].view_as(g))
        off += g.numel()


def train(envs=4096, updates=10, rollout=32, device="cuda:0", seed=0, log=print, worlds="fresh", regen=0, log_every=1,
          task="colav", step_mode=None, graph_rollout=False, sub_batches=4, minibatches=32,
          reward_scale=0.01, reward_clip=0.0, min_cumulative_reward=None, act_space="raw", ret_norm=False, orthogonal=False, ent_coef=0.01, log_std=-0.5, lr=2e-4,
          fused_policy=True, graph_update=False, policy_bf16=False):
    from gym_auv_amd import distributed as D
    from gym_auv_amd.batched_env import BatchedAuvEnv
    from gym_auv_amd.config import effective_reference_config
    from gym_auv_amd.devgen import FreshWorlds, GeneratedWorlds
    from gym_auv_amd.world import build_bank_parallel
    rank, world = D.rank(), D.world_size()
    torch.manual_seed(seed + rank)
    # task "colav": MovingObstaclesNoRules-v0 (17 moving + 11 static obstacles, LiDAR on, ColavRewarder);
    # task "pathfollow": PathFollowNoObstacles-v0 (no obstacles, LiDAR off, PathFollowRewarder) -- gym_auv/__init__.py:105-121
    colav = task == "colav"
    cfg = effective_reference_config(use_lidar=colav)
    if min_cumulative_reward is not None:       # (diagnosis only: the reference ends an episode below -2000, config.py:16)
        cfg.episode.min_cumulative_reward = float(min_cumulative_reward)
    nm, ns = (17, 11) if colav else (0, 0)
    if worlds == "fresh":
        # a new scenario on every reset; the world of an episode = f(seed, GLOBAL environment index, serial): data-parallel ranks
        # own disjoint index ranges
        bank = FreshWorlds(n_moving=nm, n_static=ns, seed=1000 * seed, env_index_base=rank * envs)
    elif worlds == "generated":
        bank = GeneratedWorlds(2 * envs, nm, ns, seed=1000 * seed + rank)
    else:
        bank = build_bank_parallel("moving_obstacles_world", range(5000 + 512 * rank, 5000 + 512 * rank + min(envs, 512)),
                                   procs=min(8, os.cpu_count() or 1), **({} if colav else dict(n_moving=0, n_static=0)))
    env = BatchedAuvEnv(cfg, bank, envs, device=device, auto_reset=True, rewarder="colav" if colav else "pathfollow")
    if step_mode:
        env.set_step_mode(step_mode)
    K = max(1, int(sub_batches))
    if not colav:
        K = 1                                   # (no LiDAR: the three-launch shape; one chain)
    slices = env.set_sub_batches(K)
    K = env.sub_batches
    streams = env._sub_streams
    low = torch.as_tensor(env.action_space.low, device=device)
    high = torch.as_tensor(env.action_space.high, device=device)
    if act_space == "normalized":     # the policy's box is [-1, 1]^2, mapped onto the action space
        a_mid, a_half = 0.5 * (high + low), 0.5 * (high - low)
        c_lo, c_hi = -torch.ones_like(low), torch.ones_like(high)
    else:                             # raw units, clipped to the action space (what stable-baselines does with a Box)
        a_mid, a_half = torch.zeros_like(low), torch.ones_like(high)
        c_lo, c_hi = low, high
    net = ActorCritic(env.obs_dim, log_std=log_std, orthogonal=orthogonal).to(device)
    params = list(net.parameters())
    pi_params, v_params = list(net.pi.parameters()) + [net.log_std], list(net.v.parameters())
    if world > 1:   # data parallel over GPUs: same initial weights, gradients averaged over RCCL
        for prm in params:
            torch.distributed.broadcast(prm.data, 0)
    graph_update = int(bool(graph_update)) if world == 1 else 0      # (the gradient all-reduce of data parallelism stays eager)
    opt = torch.optim.Adam(params, lr=lr, capturable=bool(graph_update))
    # --graph-update: a device-side record of every minibatch step (the two gradient norms before clipping, the loss, the
    # largest |advantage| and probability ratio, counts of non-finite inputs and gradient elements), written by the step
    # itself -- also inside a captured graph, where nothing can be printed -- and read once per update
    DIAG = 8192
    diag = torch.zeros((DIAG, 8), device=device)
    diag_pos = torch.zeros(1, dtype=torch.int64, device=device)
    diag_row = torch.zeros((1, 8), device=device)
    upd_graph, upd_in, upd_loss = None, None, None
    gamma, lam, clip, epochs, n_mb = 0.999, 0.98, 0.2, 4, int(minibatches)
    ret_mean, ret_std = torch.zeros((), device=device), torch.ones((), device=device)
    T, Dobs = int(rollout), env.obs_dim
    env.reset()
    act_buf = torch.zeros((envs, 2), device=device)
    # per-chain rollout storage, written inside the chain's step at a device-side position
    buf = []
    for lo, cnt in slices:
        buf.append(dict(O=torch.zeros((T, cnt, Dobs), device=device), A=torch.zeros((T, cnt, 2), device=device),
                        LP=torch.zeros((T, cnt), device=device), V=torch.zeros((T, cnt), device=device),
                        R=torch.zeros((T, cnt), device=device), Dn=torch.zeros((T, cnt), device=device),
                        t=torch.zeros(1, dtype=torch.int64, device=device)))

    def chain_step(i):
        """One transition of sub-batch i, everything enqueued on the current stream: store the observation, evaluate
        policy and value net, sample, step the sub-batch's environments, store the transition."""
        lo, cnt = slices[i]
        b = buf[i]
        with torch.no_grad():
            o = env.obs[lo:lo + cnt]
            b["O"].index_copy_(0, b["t"], o.unsqueeze(0))
            mu = net.pi(o)
            a = mu + net.log_std.exp() * torch.randn_like(mu)              # in the normalised box
            b["A"].index_copy_(0, b["t"], a.unsqueeze(0))
            b["LP"].index_copy_(0, b["t"], net.log_prob(mu, a).unsqueeze(0))
            b["V"].index_copy_(0, b["t"], net.v(o).squeeze(-1).unsqueeze(0))
            act_buf[lo:lo + cnt] = a_mid + a_half * torch.max(torch.min(a, c_hi), c_lo)
            env.step_slice(i, act_buf, stream=torch.cuda.current_stream(device))
            r = env.reward[lo:lo + cnt]
            if reward_clip > 0.0:       # the LEARNING signal only: a collision's -5000 against step rewards of order 1 leaves
                r = r.clamp(-reward_clip, reward_clip)     # normalised advantages with nothing but collision noise in them
            b["R"].index_copy_(0, b["t"], (r * reward_scale).unsqueeze(0))
            b["Dn"].index_copy_(0, b["t"], env.done[lo:lo + cnt].float().unsqueeze(0))
            b["t"] += 1

    fused = None
    if fused_policy:
        # the policy in the loop as ONE HIP launch per chain and step (gym_auv_amd/policy.py, csrc/k6_policy.hip): actor and
        # critic on the matrix cores in f32, sampling / log-probability / transition stores in its epilogue; a whole rollout
        # of every chain is one C call.  The torch modules stay the owners of the weights (refresh() after each update).
        from gym_auv_amd.policy import FusedActorCritic
        fused = FusedActorCritic(net, env, rollout=T, reward_scale=reward_scale, reward_clip=reward_clip, act_mid=a_mid.tolist(),
                                 act_half=a_half.tolist(), clip_lo=c_lo.tolist(), clip_hi=c_hi.tolist(), seed=1000 * seed + rank,
                                 bf16=policy_bf16)
        graph_rollout = False
    graphs = None
    if graph_rollout:
        # every chain's step as ONE captured device graph (replayed on the chain's stream)
        graphs = []
        for i in range(K):
            with torch.cuda.stream(streams[i]):
                for _ in range(3):                        # (warm-up off the capture, as torch asks)
                    chain_step(i)
                buf[i]["t"].zero_()
        torch.cuda.synchronize()
        for i in range(K):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=streams[i]):
                chain_step(i)
            buf[i]["t"].zero_()
            graphs.append(g)
        torch.cuda.synchronize()
        env.reset()
        env.episode_log()                                 # (drop what the warm-up steps logged)
    history = []
    n_total = envs * T
    for upd in range(updates):
        if worlds == "generated" and regen > 0 and upd and upd % regen == 0:
            # fresh scenarios for every environment, built on the device; all envs restart
            env.generate(GeneratedWorlds(2 * envs, nm, ns, seed=1000 * seed + rank + 7919 * upd))
            env.reset()
        t0 = time.time()
        cur = torch.cuda.current_stream(device)
        if fused is not None:
            fused.refresh()                               # (the update of the previous round moved the weights)
            for i in range(K):
                streams[i].wait_stream(cur)
            fused.begin_rollout()
            fused.rollout(T)                              # T x K x (policy launch + environment step), one C call
            for i in range(K):
                cur.wait_stream(streams[i])
        else:
            for i in range(K):
                buf[i]["t"].zero_()                       # (on the caller's stream, BEFORE the chain is told to wait for it)
                streams[i].wait_stream(cur)               # (the update of the previous round wrote the weights there)
            for t_roll_step in range(T):
                if worlds == "fresh" and t_roll_step % 16 == 0:
                    env.refill()                              # (this loop steps slice by slice: the library cannot tick by itself)
                for i in range(K):
                    with torch.cuda.stream(streams[i]):
                        if graphs is not None:
                            graphs[i].replay()
                        else:
                            chain_step(i)
            for i in range(K):
                cur.wait_stream(streams[i])
        torch.cuda.synchronize()
        t_roll_only = time.time() - t0                   # the rollout proper: policy in the loop, environment steps, transition stores
        if env.health()["pending"]:                      # (a replayed step is not checked by the library: ADVICE r2)
            raise RuntimeError("in-launch hand-over timed out during the rollout; env.step() will recover and report")
        with torch.no_grad():
            if fused is not None:
                O, A, LP, V, R, Dn = fused.buffers()
                V = V * ret_std + ret_mean
            else:
                O = torch.cat([b["O"] for b in buf], 1)
                A = torch.cat([b["A"] for b in buf], 1)
                LP = torch.cat([b["LP"] for b in buf], 1)
                V = torch.cat([b["V"] for b in buf], 1) * ret_std + ret_mean        # value head predicts normalised returns
                R = torch.cat([b["R"] for b in buf], 1)
                Dn = torch.cat([b["Dn"] for b in buf], 1)
            last_v = net.v(env.obs).squeeze(-1) * ret_std + ret_mean
            if fused is not None:
                adv, RET = fused.gae(V, last_v, gamma, lam)            # one launch (auv_gae) instead of T x 6 small ones
            else:
                adv = torch.zeros_like(R)
                gae = torch.zeros(envs, device=device)
                for t in reversed(range(T)):
                    nv = last_v if t == T - 1 else V[t + 1]
                    delta = R[t] + gamma * nv * (1 - Dn[t]) - V[t]
                    gae = delta + gamma * lam * (1 - Dn[t]) * gae
                    adv[t] = gae
            if fused is None:
                RET = adv + V
            # running statistics of the returns (one pass of exponential averaging per update)
            m, s = RET.mean(), RET.std()
            if world > 1:
                ms = torch.stack([m, s])
                torch.distributed.all_reduce(ms)
                m, s = ms[0] / world, ms[1] / world
            if not ret_norm:
                pass                                      # (value head in scaled-reward units)
            elif upd == 0:
                ret_mean.copy_(m), ret_std.copy_(s.clamp(min=1e-3))
            else:
                ret_mean.mul_(0.9).add_(0.1 * m), ret_std.mul_(0.9).add_(0.1 * s.clamp(min=1e-3))
            RETn = ((RET - ret_mean) / ret_std).reshape(n_total)
            ADV = adv.reshape(n_total)
            ADV = (ADV - ADV.mean()) / (ADV.std() + 1e-8)
            O, A, LP = O.reshape(n_total, Dobs), A.reshape(n_total, 2), LP.reshape(n_total)
        torch.cuda.synchronize()
        t_roll = time.time() - t0

        def minibatch_step(o, a, lp, advn, retn):
            mu = net.pi(o)
            ratio = (net.log_prob(mu, a) - lp).exp()
            pg = -torch.min(ratio * advn, ratio.clamp(1 - clip, 1 + clip) * advn).mean()
            vf = 0.5 * (net.v(o).squeeze(-1) - retn).pow(2).mean()
            loss = pg + 0.5 * vf - ent_coef * net.entropy()
            opt.zero_grad(set_to_none=not graph_update)
            loss.backward()
            average_gradients(params, world)
            # (policy and value net are separate networks: clipped separately, so that a value loss swollen by a rare
            # -5000 collision return cannot scale the policy's gradient away under a shared norm)
            # counts of non-finite inputs / gradient elements (float accumulation)
            bad_in = torch.zeros((), device=device)
            for x in (o, a, lp, advn, retn):
                bad_in = bad_in + (~torch.isfinite(x)).sum(dtype=torch.float32)
            bad_g = torch.zeros((), device=device)
            for q in params:
                if q.grad is not None:
                    bad_g = bad_g + (~torch.isfinite(q.grad)).sum(dtype=torch.float32)
            n_pi = clip_grad_norm(pi_params, 0.5)
            n_v = clip_grad_norm(v_params, 0.5)
            # (the record is filled element by element: no stack / cat inside a region that may be captured, see clip_grad_norm)
            for j, x in enumerate((n_pi, n_v, loss.detach(), advn.abs().max(), ratio.detach().max(), bad_in, bad_g)):
                diag_row[0, j].copy_(x)
            diag.index_copy_(0, diag_pos % DIAG, diag_row)
            diag_pos.add_(1)
            opt.step()
            if PROBE is not None and torch.cuda.is_current_stream_capturing():
                # (AUV_PPO_PROBE=1, tools/graph_update_probe.sh: keep the captured step's record inputs alive so that their
                # addresses can be compared with every other live tensor once a replay shows a corrupt record)
                PROBE.update(bad_in=bad_in, bad_g=bad_g, n_pi=n_pi, n_v=n_v, loss=loss, max_adv=advn.abs().max(), ratio_max=ratio.detach().max())
            return loss

        # --graph-update: a minibatch's forward, backward, clipping and Adam step as ONE captured device graph with static input
        # buffers, refilled by gathers before every replay.  Round 3 removed this path because a replayed graph "read its own
        # gradient norm as inf" and froze the weights without the cause having been found; it is back with the record
        # above, so that a freeze names its cause: see DESIGN.md section 8 for what the record showed.
        d0 = int(diag_pos.item())
        mb_size = n_total // n_mb
        for _ in range(epochs):
            perm = torch.randperm(n_total, device=device)
            for mb in perm.chunk(n_mb):
                if not graph_update or mb.numel() != mb_size:      # (a ragged last chunk would RESIZE the static buffers)
                    loss = minibatch_step(O[mb], A[mb], LP[mb], ADV[mb], RETn[mb])
                    continue
                if upd_in is None:
                    upd_in = [O[mb].clone(), A[mb].clone(), LP[mb].clone(), ADV[mb].clone(), RETn[mb].clone()]
                    side = torch.cuda.Stream(device=device)
                    side.wait_stream(torch.cuda.current_stream(device))
                    with torch.cuda.stream(side):                  # (warm-up off the capture, as torch asks: three real steps)
                        for _w in range(3):
                            minibatch_step(*upd_in)
                    torch.cuda.current_stream(device).wait_stream(side)
                    upd_graph = torch.cuda.CUDAGraph()
                    if PROBE is not None:
                        upd_graph.enable_debug_mode()
                    with torch.cuda.graph(upd_graph):
                        upd_loss = minibatch_step(*upd_in)
                    if PROBE is not None:               # (the captured graph's nodes and edges: is it ONE chain?)
                        upd_graph.debug_dump(os.environ.get("AUV_PPO_PROBE_DOT", "/tmp/ppo_update_graph.dot"))
                    continue
                for dst, src in zip(upd_in, (O, A, LP, ADV, RETn)):
                    torch.index_select(src, 0, mb, out=dst)
                upd_graph.replay()
                loss = upd_loss
        torch.cuda.synchronize()
        dt_all = time.time() - t0
        # ---- what happened: step rewards of the rollout, and the episodes that ended during it (library's episode log)
        mean_r = float(R.mean().item()) / reward_scale
        ep = env.episode_log()
        if ep.shape[0]:
            col, goal = float((ep[:, 3] > 0).double().mean()), float((ep[:, 4] > 0).double().mean())
            ep_ret, ep_len, ep_prog = float(ep[:, 1].mean()), float(ep[:, 2].mean()), float(ep[:, 5].mean())
        else:
            col = goal = ep_ret = ep_len = ep_prog = float("nan")
        surge, he, cte = float(O[:, 0].mean()), float(O[:, 4].abs().mean()), float(O[:, 5].abs().mean()) * 100
        d1 = int(diag_pos.item())
        rows = diag[torch.arange(d0, d1, device=device) % DIAG] if d1 - d0 <= DIAG else diag
        steps_bad = int((~torch.isfinite(rows[:, :3])).any(1).sum()) + int((rows[:, 5:7] > 0).any(1).sum())
        history.append(dict(update=upd, minibatch_steps=d1 - d0, minibatch_steps_nonfinite=steps_bad,
                            grad_norm_pi=float(rows[:, 0].mean()), grad_norm_v=float(rows[:, 1].mean()), max_ratio=float(rows[:, 4].max()),
                            mean_step_reward=mean_r, loss=float(loss.item()), surge=surge, heading_error=he,
                            weight_l1=float(sum(p_.detach().abs().sum() for p_ in pi_params)),
                            cross_track=cte, episodes=int(ep.shape[0]), goal_rate=goal, collision_rate=col, ep_return=ep_ret,
                            ep_len=ep_len, ep_progress=ep_prog, rollout_sps=world * n_total / t_roll_only, rollout_gae_sps=world * n_total / t_roll, sps=world * n_total / dt_all))
        if PROBE and steps_bad:
            probe_report(PROBE, diag_row, log)
            PROBE.clear()
        if graph_update and steps_bad:
            # The captured update's own record shows values that cannot come from its inputs (a gradient norm of inf over
            # finite gradient elements, or a count of non-finite inputs far beyond the number of inputs): the replayed
            # graph's reductions returned stale memory (DESIGN.md section 8 -- reproduced in this loop, not root-caused
            # below the framework).  Such a step scales the gradients by 0 or garbage; from here on the update runs eagerly.
            log("update %4d: the captured update step returned corrupt reductions in %d of %d replays -- falling back to the eager update"
                % (upd, steps_bad, d1 - d0))
            graph_update, upd_graph = 0, None
        if rank == 0 and (upd % log_every == 0 or upd == updates - 1):
            if steps_bad:
                bad = rows[(~torch.isfinite(rows[:, :3])).any(1) | (rows[:, 5:7] > 0).any(1)][:4]
                log("update %4d: %d of %d minibatch steps non-finite; first rows [|g_pi|, |g_v|, loss, max|adv|, max ratio, bad inputs, bad grads]: %s"
                    % (upd, steps_bad, d1 - d0, bad[:, :7].tolist()))
            log("update %4d  loss %9.4f  |w| %.4f  step reward %7.3f  surge %.3f  |he| %.2f  |cte| %6.1f m  std %s | episodes %5d: goal %.3f collision %.3f "
                "other %.3f  return %8.1f  length %6.1f  progress %.3f | rollout %.2e env-steps/s (policy in the loop), %.2e incl. learning"
                % (upd, float(loss.item()), float(sum(p_.detach().abs().sum() for p_ in pi_params)), mean_r, surge, he, cte,
                   ["%.3f" % x for x in net.log_std.exp().tolist()], ep.shape[0], goal, col,
                   1.0 - goal - col if ep.shape[0] else float("nan"), ep_ret, ep_len, ep_prog, world * n_total / t_roll_only, world * n_total / dt_all))
    # the one collective of the environment side: finished-episode statistics of all ranks
    stats = D.gather_episode_stats(env.episode_stats())
    if rank == 0 and history:
        fin = stats["episodes"] > 0
        log("episodes finished %d; last-episode return mean %.1f, collision rate %.2f, goal rate %.2f" % (
            int(stats["episodes"].sum()), float(stats["episode_return"][fin].mean()) if fin.any() else float("nan"),
            float(stats["collision"][fin].mean()) if fin.any() else float("nan"),
            float(stats["reached_goal"][fin].mean()) if fin.any() else float("nan")))
        k = max(1, min(10, len(history) // 4))
        first, last = history[:k], history[-k:]
        avg = lambda rows, key: sum(r[key] for r in rows) / len(rows)   # noqa: E731
        log("learning: mean step reward %.3f -> %.3f, surge %.3f -> %.3f m/s, |heading error| %.2f -> %.2f rad, goal rate %.3f -> %.3f, "
            "collision rate %.3f -> %.3f (first / last %d updates)"
            % (avg(first, "mean_step_reward"), avg(last, "mean_step_reward"), avg(first, "surge"), avg(last, "surge"),
               avg(first, "heading_error"), avg(last, "heading_error"), avg(first, "goal_rate"), avg(last, "goal_rate"),
               avg(first, "collision_rate"), avg(last, "collision_rate"), k))
    if worlds == "fresh" and rank == 0:
        st = env.fresh_stats()
        log("fresh worlds: %d rebuilt beside the rollouts, %d episode(s) had to re-use their world (0 = every reset met an unseen one)"
            % (st["regenerated"], st["reused"]))
    env.close()
    return history


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--updates", type=int, default=10)
    ap.add_argument("--rollout", type=int, default=32)
    ap.add_argument("--worlds", default="fresh", choices=["fresh", "generated", "host"])
    ap.add_argument("--regen", type=int, default=0, help="regenerate the world bank on the device every this many updates (0: never)")
    ap.add_argument("--log-every", type=int, default=1)
    ap.add_argument("--task", default="colav", choices=["colav", "pathfollow"])
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--sub-batches", type=int, default=4, help="rollout chains (BatchedAuvEnv.set_sub_batches)")
    ap.add_argument("--minibatches", type=int, default=32)
    ap.add_argument("--graph-rollout", type=int, default=0, help="1: one captured device graph per chain and rollout step (torch-module policy only)")
    ap.add_argument("--graph-update", type=int, default=0,
                    help="1 (EXPERIMENTAL): a minibatch's forward, backward, clipping and Adam step as one captured device graph (single "
                         "rank).  Every replay writes a record; the update falls back to eager the moment a record is corrupt")
    ap.add_argument("--policy-bf16", type=int, default=0,
                    help="1: the fused policy launch multiplies with bf16 weights on the bf16 matrix cores (faster rollouts, ~1e-2 on the "
                         "action means: the PPO ratio then compares log-probabilities of slightly different policies); default 0 = exact f32")
    ap.add_argument("--fused-policy", type=int, default=1,
                    help="1 (default): the policy in the loop is ONE HIP launch per chain and step (gym_auv_amd/policy.py); 0: the torch modules")
    ap.add_argument("--step-mode", default=None, help="launch shape of a step (BatchedAuvEnv.STEP_MODES); default: the library's")
    ap.add_argument("--act-space", default="raw", choices=["raw", "normalized"])
    ap.add_argument("--ret-norm", type=int, default=0)
    ap.add_argument("--orthogonal", type=int, default=0)
    ap.add_argument("--ent-coef", type=float, default=0.01)
    ap.add_argument("--log-std", type=float, default=-0.5)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--reward-clip", type=float, default=0.0, help="> 0: step rewards are clipped to +- this for the learning signal")
    ap.add_argument("--min-cumulative-reward", type=float, default=None,
                    help="diagnosis: EpisodeConfig.min_cumulative_reward (reference: -2000; the episode ends below it)")
    a = ap.parse_args()
    from gym_auv_amd import distributed as D
    _rank, _world, local = D.init_from_env()     # one process per GPU under torch.distributed.run; cuda:0 alone
    train(a.envs, a.updates, a.rollout, device="cuda:%d" % local, seed=a.seed, worlds=a.worlds, regen=a.regen, log_every=a.log_every,
          task=a.task, step_mode=a.step_mode, graph_rollout=bool(a.graph_rollout), fused_policy=bool(a.fused_policy), graph_update=a.graph_update, policy_bf16=bool(a.policy_bf16),
          sub_batches=a.sub_batches, minibatches=a.minibatches, act_space=a.act_space, ret_norm=bool(a.ret_norm),
          orthogonal=bool(a.orthogonal), ent_coef=a.ent_coef, log_std=a.log_std, lr=a.lr, reward_clip=a.reward_clip,
          min_cumulative_reward=a.min_cumulative_reward)
