"""-m gpu: the fused policy launch (csrc/k6_policy.hip, auv_policy_act / auv_policy_rollout) against the fp32 torch modules
it replaces in rollouts (examples/ppo.py ActorCritic == the reference's MlpPolicy [256, 128, 64], scripts/run.py:332-357).
Tolerance: 1e-5 on means and values (f32 MFMA chains against hipBLASLt's f32 GEMMs: same precision, another summation
order), written at each comparison."""
import os
import sys

import numpy as np
import pytest
import torch

from gym_auv_amd.config import effective_reference_config
from gym_auv_amd.scenarios import moving_obstacles_world
from gym_auv_amd.world import build_world, pack_bank

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))


def _setup(n, k, T, use_lidar=True, seed=0, **kw):
    import ppo
    from gym_auv_amd.batched_env import BatchedAuvEnv
    from gym_auv_amd.policy import FusedActorCritic
    cfg = effective_reference_config(use_lidar=use_lidar)
    bank = pack_bank([build_world(moving_obstacles_world(2000 + i) if use_lidar else moving_obstacles_world(2000 + i, 0, 0)) for i in range(16)])
    env = BatchedAuvEnv(cfg, bank, n, device="cuda:0", rewarder="colav" if use_lidar else "pathfollow")
    env.reset()
    env.set_sub_batches(k, probe_streams=False)
    torch.manual_seed(seed)
    net = ppo.ActorCritic(env.obs_dim).to("cuda:0")
    with torch.no_grad():
        net.log_std.copy_(torch.tensor([-0.5, -1.1]))
        for m in net.modules():                      # weights of ordinary size, biases that matter
            if isinstance(m, torch.nn.Linear):
                m.bias.uniform_(-0.3, 0.3)
    fused = FusedActorCritic(net, env, rollout=T, debug=True, seed=seed, **kw)
    return env, net, fused


@pytest.mark.parametrize("n,k,use_lidar", [(1000, 3, True), (64, 1, True), (300, 2, False)])
def test_policy_launch_matches_the_torch_modules(n, k, use_lidar):
    T = 6
    env, net, fused = _setup(n, k, T, use_lidar, reward_scale=0.01, reward_clip=50.0)
    fused.begin_rollout()
    lo_a = torch.as_tensor(env.action_space.low, device="cuda:0")
    hi_a = torch.as_tensor(env.action_space.high, device="cuda:0")
    prev = None
    for t in range(T + 1):
        obs = env.obs.clone()
        rew, done = env.reward.clone(), env.done.clone()
        for i in range(env.sub_batches):
            fused.act(i)
        torch.cuda.synchronize()
        if t == T:
            break                                              # the flush call: only R / Dn of the last step
        with torch.no_grad():
            mu_ref, v_ref = net.pi(obs), net.v(obs).squeeze(-1)
        mu, eps = fused.mu, fused.eps
        O, A, LP, V, R, Dn = fused.buffers()
        assert float((mu - mu_ref).abs().max()) <= 1e-5        # f32 on the matrix cores against torch's f32 GEMMs
        assert float((V[t] - v_ref).abs().max()) <= 1e-5 * max(1.0, float(v_ref.abs().max()))
        assert torch.equal(O[t], obs)
        a_ref = mu + net.log_std.exp() * eps
        assert float((A[t] - a_ref).abs().max()) <= 1e-6
        with torch.no_grad():
            lp_ref = net.log_prob(mu_ref, A[t])
        assert float((LP[t] - lp_ref).abs().max()) <= 2e-4     # ((a - mu) / sigma)^2 amplifies the 1e-5 of mu by |z| / sigma
        act_ref = torch.max(torch.min(A[t], hi_a), lo_a)
        assert torch.equal(fused.actions, act_ref)
        if t >= 1:
            assert torch.equal(R[t - 1], rew.clamp(-50.0, 50.0) * 0.01) and torch.equal(Dn[t - 1], done.float())
        for i in range(env.sub_batches):
            assert int(fused.buf[i]["ctr"][0]) == t + 1 and int(fused.buf[i]["ctr"][2]) == 0
            env.step_slice(i, fused.actions)
        torch.cuda.synchronize()
    O, A, LP, V, R, Dn = fused.buffers()
    assert torch.equal(R[T - 1], env.reward.clamp(-50.0, 50.0) * 0.01) and torch.equal(Dn[T - 1], env.done.float())
    for i in range(env.sub_batches):
        assert int(fused.buf[i]["ctr"][0]) == T + 1
    env.close()


def test_policy_noise_is_standard_normal_and_uncorrelated():
    env, net, fused = _setup(1024, 1, 1)
    eps = []
    for s in range(200):
        fused.begin_rollout()
        fused.act(0)
        torch.cuda.synchronize()                                # (the launch runs on the sub-batch's stream, the clone on the caller's)
        eps.append(fused.eps.clone())
    e = torch.stack(eps).double()                              # [steps, rows, 2]
    assert abs(float(e.mean())) < 0.01 and abs(float(e.var()) - 1.0) < 0.02
    assert abs(float((e ** 4).mean()) - 3.0) < 0.1             # kurtosis of a normal
    assert abs(float((e[..., 0] * e[..., 1]).mean())) < 0.01   # the two action components
    assert abs(float((e[1:] * e[:-1]).mean())) < 0.01          # consecutive steps of one environment
    assert abs(float((e[:, 1:] * e[:, :-1]).mean())) < 0.01    # neighbouring environments
    assert float(e.abs().max()) < 6.5 and torch.unique(e).numel() > 0.95 * e.numel()
    env.close()


def test_policy_rollout_call_equals_the_step_by_step_loop():
    """auv_policy_rollout (T steps of every chain in one C call) == policy launch + env.step_slice from Python: the same
    generator counters, so the same samples, the same environment states bit for bit; refresh() picks new weights up."""
    T = 12
    env_a, net_a, fa = _setup(512, 2, T, seed=3, reward_scale=0.01)
    env_b, net_b, fb = _setup(512, 2, T, seed=3, reward_scale=0.01)
    for rnd in range(2):
        fa.begin_rollout(), fb.begin_rollout()
        fa.rollout(T)
        for t in range(T):
            for i in range(env_b.sub_batches):
                fb.act(i)
                env_b.step_slice(i, fb.actions)
        for i in range(env_b.sub_batches):
            fb.act(i)                                          # flush
        torch.cuda.synchronize()
        for x, y in zip(fa.buffers(), fb.buffers()):
            assert torch.equal(x, y)
        for f in ("STATE", "OBS64", "COUNTERS", "REWARD64"):
            assert torch.equal(env_a.read(f), env_b.read(f)), f
        with torch.no_grad():                                  # an "optimiser step": both nets move the same way
            for p, q in zip(net_a.parameters(), net_b.parameters()):
                d = 0.05 * torch.randn_like(p)
                p.add_(d), q.add_(d)
        fa.refresh(), fb.refresh()
    env_a.close(), env_b.close()


def test_gae_kernel_matches_the_reference_recursion():
    """auv_gae against the plain backward recursion (what examples/ppo.py used to run as T small tensor operations per
    rollout; stable-baselines' PPO2 runner computes the same on the host), gamma 0.999, lam 0.98 (scripts/run.py:341-346)."""
    env, net, fused = _setup(1000, 3, 20)
    T, N = 20, 1000
    g = torch.Generator(device="cuda:0").manual_seed(1)
    fused.R.copy_(torch.randn((T, N), device="cuda:0", generator=g))
    V = torch.randn((T, N), device="cuda:0", generator=g)
    fused.Dn.copy_((torch.rand((T, N), device="cuda:0", generator=g) < 0.1).float())
    last_v = torch.randn(N, device="cuda:0", generator=g)
    adv, ret = fused.gae(V, last_v, 0.999, 0.98)
    ref = torch.zeros_like(adv)
    gae = torch.zeros(N, device="cuda:0")
    for t in reversed(range(T)):
        nv = last_v if t == T - 1 else V[t + 1]
        delta = fused.R[t] + 0.999 * nv * (1 - fused.Dn[t]) - V[t]
        gae = delta + 0.999 * 0.98 * (1 - fused.Dn[t]) * gae
        ref[t] = gae
    assert float((adv - ref).abs().max()) <= 1e-5 * float(ref.abs().max())    # (f32, another association of the products)
    assert float((ret - (ref + V)).abs().max()) <= 1e-5 * float((ref + V).abs().max())
    env.close()


def test_policy_bf16_option_matches_a_bf16_emulation():
    """FusedActorCritic(bf16=True) -- optional, NOT the reference's arithmetic: weights stored as bf16, activations rounded to
    bf16 on the way into v_mfma_f32_16x16x32_bf16, f32 accumulation / bias / tanh.  Against the same computation spelt out in
    torch (2e-3: the summation order differs), and within 5e-2 of the exact f32 modules."""
    import ppo
    from gym_auv_amd.policy import FusedActorCritic
    env, net, _ = _setup(512, 2, 4)
    fused = FusedActorCritic(net, env, rollout=4, debug=True, bf16=True)
    fused.begin_rollout()
    obs = env.obs.clone()
    for i in range(env.sub_batches):
        fused.act(i)
    torch.cuda.synchronize()

    def emulate(seq, x):
        lin = [m for m in seq if isinstance(m, torch.nn.Linear)]
        for j, l in enumerate(lin):
            x = x.bfloat16().float() @ l.weight.bfloat16().float().t() + l.bias
            if j < len(lin) - 1:
                x = torch.tanh(x)
        return x
    with torch.no_grad():
        mu_emu, v_emu = emulate(net.pi, obs), emulate(net.v, obs).squeeze(-1)
        mu_f32, v_f32 = net.pi(obs), net.v(obs).squeeze(-1)
    assert float((fused.mu - mu_emu).abs().max()) <= 2e-3
    assert float((fused.V[0] - v_emu).abs().max()) <= 2e-3
    assert float((fused.mu - mu_f32).abs().max()) <= 5e-2 and float((fused.V[0] - v_f32).abs().max()) <= 5e-2
    env.close()
