"""FusedActorCritic -- the PPO policy in the loop as ONE HIP launch per sub-batch and step (csrc/k6_policy.hip).

The reference trains PPO2 with `MlpPolicy`, net_arch [256, 128, 64] for policy and value function
(/root/reference/scripts/run.py:332-357), stepping its environments through a VecEnv (run.py:293-296).  With thousands of
environments per GPU stepping in ~30 us, evaluating the two small MLPs with stock tensor operations (eight GEMMs and a
dozen element-wise launches per step) is what bounds a rollout.  This class keeps the torch modules as the owner of the
weights -- the optimiser updates them as usual -- and evaluates them during rollouts with `auv_policy_act`: observation
rows -> actor and critic on the matrix cores in f32 -> sampled action into the environment's action buffer, log-probability
and value into the rollout buffers; reward / done of the previous step are stored by the same launch.

    fused = FusedActorCritic(net, env, rollout=T, reward_scale=0.01)     # net.pi / net.v: Linear-Tanh x 3 + Linear, net.log_std
    fused.refresh()                                # after every optimiser step: repack the weights (device-side copies)
    fused.begin_rollout()                          # t = 0 on every chain
    fused.rollout(T)                               # T steps of every sub-batch chain, policy + env.step, one C call
    O, A, LP, V, R, Dn = fused.buffers()           # [T, N, ...] views for the update
"""
import ctypes as C
from typing import List, Optional

import torch
import torch.nn as nn

from . import _capi
from .batched_env import BatchedAuvEnv, _check, _LIB

HIDDEN = (256, 128, 64)


def _pad16(x: int) -> int:
    return (x + 31) & ~31          # (rows are padded to the kernel's k-step of 32 columns)


def pack_linear(weight: torch.Tensor, out_pad: int, in_pad: int) -> torch.Tensor:
    """A Linear.weight [out, in] zero-padded to [out_pad, in_pad] (multiples of 16 / 32) and laid out in MFMA fragment order
    (include/auv_hip.h, auv_policy_io): [n-tile][k-step J][half h][group g][row n][4 floats] -- lane 16 g + n of a wave holds
    k = 32 J + 8 g + 4 h + (0..3) of row n of the tile, and each load instruction of the kernel reads 1 KB contiguously."""
    w = torch.zeros((out_pad, in_pad), dtype=torch.float32, device=weight.device)
    w[:weight.shape[0], :weight.shape[1]].copy_(weight)
    return w.view(out_pad // 16, 16, in_pad // 32, 4, 2, 4).permute(0, 2, 4, 3, 1, 5).contiguous().view(-1)


def pack_linear_bf16(weight: torch.Tensor, out_pad: int, in_pad: int) -> torch.Tensor:
    """The same matrix as bf16 fragments for v_mfma_f32_16x16x32_bf16 (auv_policy_io::params_bf16): [n-tile][k-step J][group g]
    [row n][8 bf16] -- lane 16 g + n holds k = 32 J + 8 g + (0..7) of row n, 16 bytes."""
    w = torch.zeros((out_pad, in_pad), dtype=torch.float32, device=weight.device)
    w[:weight.shape[0], :weight.shape[1]].copy_(weight)
    return w.view(out_pad // 16, 16, in_pad // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1).to(torch.bfloat16)


class FusedActorCritic:
    def __init__(self, net: nn.Module, env: BatchedAuvEnv, rollout: int, reward_scale: float = 1.0, reward_clip: float = 0.0,
                 act_mid=None, act_half=None, clip_lo=None, clip_hi=None, seed: int = 0, store_obs: bool = True, debug: bool = False,
                 bf16: bool = False):
        self.net, self.env, self.T = net, env, int(rollout)
        self.device = env.device
        self._lin = []
        for seq, out in ((net.pi, 2), (net.v, 1)):
            lin = [m for m in seq if isinstance(m, nn.Linear)]
            act = [m for m in seq if not isinstance(m, nn.Linear)]
            dims = tuple(l.out_features for l in lin[:-1])
            if dims != HIDDEN or lin[-1].out_features != out or lin[0].in_features != env.obs_dim or not all(isinstance(a, nn.Tanh) for a in act):
                raise ValueError("FusedActorCritic evaluates the reference's architecture: obs -> %s tanh -> %d (scripts/run.py:332-357); "
                                 "got %s -> %d" % (list(HIDDEN), out, list(dims), lin[-1].out_features))
            self._lin.append(lin)
        self.k0p = _pad16(env.obs_dim)
        n_float = int(_LIB.auv_policy_param_floats(env.obs_dim))
        self.params = torch.zeros(n_float, dtype=torch.float32, device=self.device)
        assert self.params.data_ptr() % 16 == 0
        # bf16 = True: the weights also as bf16 fragments, and the launch multiplies on v_mfma_f32_16x16x32_bf16 (one MFMA where
        # the exact path issues eight; ~1e-2 on the means -- NOT the reference's arithmetic; default off)
        self.bf16 = bool(bf16)
        n_w = (n_float - 4) // 2 - (sum(HIDDEN) + 16)                                  # weight elements of one net
        self.params_bf16 = torch.zeros(2 * n_w, dtype=torch.bfloat16, device=self.device) if self.bf16 else None
        if env._slices is None:
            env.set_sub_batches(1)
        self.slices = list(env._slices)
        D = env.obs_dim
        lo_t = torch.as_tensor(env.action_space.low, dtype=torch.float32)
        hi_t = torch.as_tensor(env.action_space.high, dtype=torch.float32)
        # default: raw units clipped to the action space (what stable-baselines does with a Box)
        mid = [0.0, 0.0] if act_mid is None else [float(x) for x in act_mid]
        half = [1.0, 1.0] if act_half is None else [float(x) for x in act_half]
        clo = lo_t.tolist() if clip_lo is None else [float(x) for x in clip_lo]
        chi = hi_t.tolist() if clip_hi is None else [float(x) for x in clip_hi]
        self.actions = torch.zeros((env.n_envs, 2), dtype=torch.float32, device=self.device)
        N = env.n_envs
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=self.device)   # noqa: E731
        # ONE set of rollout buffers [T, N, ...] for the whole batch: every chain writes its own columns, nothing is
        # concatenated afterwards
        self.O = z(self.T, N, D) if store_obs else None
        self.A, self.LP, self.V, self.R, self.Dn = z(self.T, N, 2), z(self.T, N), z(self.T, N), z(self.T, N), z(self.T, N)
        self.mu = z(N, 2) if debug else None
        self.eps = z(N, 2) if debug else None
        self.buf: List[dict] = []
        self._ios = (_capi.AuvPolicyIO * len(self.slices))()
        for i, (lo, cnt) in enumerate(self.slices):
            b = dict(ctr=torch.zeros(4, dtype=torch.int64, device=self.device))
            self.buf.append(b)
            io = self._ios[i]
            io.obs, io.params, io.ctr = env.obs.data_ptr(), self.params.data_ptr(), b["ctr"].data_ptr()
            io.reward_in, io.done_in, io.actions_out = env.reward.data_ptr(), env.done.data_ptr(), self.actions.data_ptr()
            io.O = self.O.data_ptr() if store_obs else None
            io.A, io.LP, io.V, io.R, io.Dn = (x.data_ptr() for x in (self.A, self.LP, self.V, self.R, self.Dn))
            io.mu_out = self.mu[lo:].data_ptr() if debug else None       # ([ne][2] views of the slice's rows)
            io.eps_out = self.eps[lo:].data_ptr() if debug else None
            io.seed = (int(seed) * 0x9E3779B97F4A7C15 + i) & 0xFFFFFFFFFFFFFFFF
            io.obs_dim, io.T, io.ld, io.env_base = D, self.T, N, 0
            io.params_bf16 = self.params_bf16.data_ptr() if self.bf16 else None
            for k in range(2):
                io.act_mid[k], io.act_half[k], io.clip_lo[k], io.clip_hi[k] = mid[k], half[k], clo[k], chi[k]
            io.reward_scale, io.reward_clip = float(reward_scale), float(reward_clip)
        # host mirror of every chain's rollout position and generator step (the device copies in `ctr` stay in step with them):
        # rollout() names them to the launches, which then need no count-off
        self._t = [0] * len(self.slices)
        self._g = [0] * len(self.slices)
        self.refresh()

    # ------------------------------------------------------------------------------ weights
    @torch.no_grad()
    def refresh(self):
        """Repack the modules' weights into the kernel's layout (rows padded to a multiple of 16 columns, the last layer to 16
        rows): device-side copies on the current stream, no host synchronisation.  Call after every optimiser step that
        precedes a rollout."""
        off = off16 = 0
        p = self.params
        for lin in self._lin:
            for j, l in enumerate(lin):
                out_p = l.out_features if j < 3 else 16
                in_p = self.k0p if j == 0 else l.in_features
                p[off:off + out_p * in_p].copy_(pack_linear(l.weight, out_p, in_p))
                if self.bf16:
                    self.params_bf16[off16:off16 + out_p * in_p].copy_(pack_linear_bf16(l.weight, out_p, in_p))
                    off16 += out_p * in_p
                off += out_p * in_p
                p[off:off + l.out_features].copy_(l.bias)
                off += out_p
        p[off:off + 2].copy_(self.net.log_std)
        assert off + 4 == p.numel()

    # ------------------------------------------------------------------------------ rollouts
    def begin_rollout(self):
        """t = 0 on every chain (enqueued on each chain's stream, behind whatever the chain did last)."""
        for i, b in enumerate(self.buf):
            with torch.cuda.stream(self.env._sub_streams[i]):
                b["ctr"][0] = 0
            self._t[i] = 0

    def act(self, i: int, stream: Optional[torch.cuda.Stream] = None):
        """The policy launch of sub-batch i alone (auv_policy_act) on `stream` (default: the sub-batch's): writes the
        slice's rows of `self.actions` and the transition at the chain's position t, then moves t on."""
        lo, cnt = self.slices[i]
        st = self.env._sub_streams[i] if stream is None else stream
        _check(_LIB.auv_policy_act(self.env._h, lo, cnt, C.byref(self._ios[i]), C.c_void_p(st.cuda_stream)), "auv_policy_act")
        self._t[i] = min(self._t[i] + 1, self.T + 1)
        self._g[i] += 1

    def rollout(self, n_steps: int, flush: bool = True):
        """`n_steps` transitions of every sub-batch: per step and chain the policy launch and the environment's step of that
        slice, back to back on the chain's stream -- one C call (auv_policy_rollout), nothing returns to Python in between.
        `flush`: a final policy call stores reward / done of the last step.  The chains are not ordered against the caller's
        stream: order them yourself (wait_stream) around the call."""
        env = self.env
        k = len(self.slices)
        t0, g0 = (C.c_int64 * k)(*self._t), (C.c_int64 * k)(*self._g)
        _check(_LIB.auv_policy_rollout(env._h, env.sub_batches, env._bounds_c, env._streams_c, self._ios,
                                       C.c_void_p(env.obs.data_ptr()), C.c_void_p(env.reward.data_ptr()),
                                       C.c_void_p(env.done.data_ptr()), int(n_steps), int(bool(flush)), t0, g0), "auv_policy_rollout")
        n_launch = int(n_steps) + int(bool(flush))
        for i in range(k):
            self._t[i] = min(self._t[i] + n_launch, self.T + 1)
            self._g[i] += n_launch

    def buffers(self):
        """(O, A, LP, V, R, Dn) of the whole batch, [T, N, ...] (written in place by the chains: no copy)."""
        return self.O, self.A, self.LP, self.V, self.R, self.Dn

    def gae(self, V: torch.Tensor, last_v: torch.Tensor, gamma: float, lam: float):
        """Generalised advantage estimation of the stored rollout (auv_gae, one launch on the current stream): returns
        (adv, ret), [T, N].  `V` [T, N] and `last_v` [N] in the units of the stored rewards (pass `self.V` or a rescaled copy)."""
        adv, ret = torch.empty_like(self.R), torch.empty_like(self.R)
        V, last_v = V.contiguous(), last_v.contiguous()
        st = torch.cuda.current_stream(self.device)
        _check(_LIB.auv_gae(self.env._h, C.c_void_p(self.R.data_ptr()), C.c_void_p(V.data_ptr()), C.c_void_p(self.Dn.data_ptr()),
                            C.c_void_p(last_v.data_ptr()), float(gamma), float(lam), C.c_void_p(adv.data_ptr()),
                            C.c_void_p(ret.data_ptr()), self.T, self.env.n_envs, C.c_void_p(st.cuda_stream)), "auv_gae")
        return adv, ret
